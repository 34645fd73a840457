"""Worker of tests/test_gpu_shard.py: one rank of a G-rank run of BASELINE configs[4].  Sequence s -> rank s mod G
(stvo_amd.shard); every rank drives a REAL stvo_seq (per-sequence cameras) on the GPU for its sequences, the ranks
exchange nothing but the final pose blocks (all-gather over the process group: gloo here, RCCL in bench.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))


def run_rank(seq_ids, n_frames, n_pts, n_lines, device_id=0):
    """Poses [len(seq_ids), n_frames - 1, 16] + stereo counts of the given config-5 sequences in one stvo_seq."""
    from stvo_amd import capi, synth
    from stvo_amd.ctypes_types import match_params, opt_params
    B = len(seq_ids)
    seqs = [synth.make_config5_sequence(s, n_frames=n_frames, n_pts=n_pts, n_lines=n_lines) for s in seq_ids]
    cams = [synth.config5_cam(s) for s in seq_ids]
    ctx = capi.Context(device_id=device_id, max_rows=2048, max_batch=B)
    dev = capi.Sequences(ctx, B, 2048, 128, cams, match_params("kitti"), opt_params("kitti"))
    poses = np.zeros((B, n_frames - 1, 16)); status = np.zeros((B, n_frames - 1), np.int32)
    try:
        for k in range(n_frames):
            res, _ = dev.push([s[k] for s in seqs])
            if k:
                poses[:, k - 1] = res["T"]
                status[:, k - 1] = res["status"]
    finally:
        dev.close()
        ctx.close()
    return poses, status


def main():
    import torch
    import torch.distributed as dist
    from stvo_amd import shard
    n_frames, n_pts, n_lines, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    world, rank, _ = shard.env_world()
    dist.init_process_group("gloo")
    seq_ids = shard.sequences_for_rank(8, world, rank)
    poses, status = run_rank(seq_ids, n_frames, n_pts, n_lines, device_id=0)  # one GPU on the test box: both ranks share it
    total, _ = shard.aggregate(dist, len(seq_ids) * (n_frames - 1), 1.0)
    blocks = shard.gather_poses(dist, poses.reshape(-1, 16))
    stat = shard.gather_poses(dist, status.reshape(-1, 1).astype(np.float64))
    if rank == 0:
        np.savez(out_path, total=total, world=world, **{f"poses_{r}": blocks[r] for r in range(world)},
                 **{f"status_{r}": stat[r] for r in range(world)})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
