"""Multi-GPU sharding of independent stereo sequences: one process per GPU, sequence s
runs on rank s mod G, no data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only for the timing barrier and the throughput aggregation."""
import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def sequences_for_rank(n_sequences: int, world: int, rank: int):
    """Sequence s -> rank s mod world (BASELINE config 5)."""
    return [s for s in range(n_sequences) if s % world == rank]


def aggregate(dist, frames_local: int, seconds_local: float, device="cpu"):
    """(total frame pairs over all ranks, max seconds over ranks): value = total / max_time, computed by
    every rank so that rank 0 can print it.  With dist None (single process) this is the identity."""
    if dist is None:
        return frames_local, seconds_local
    import torch
    t = torch.tensor([seconds_local], dtype=torch.float64, device=device)
    n = torch.tensor([float(frames_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return int(round(n.item())), float(t.item())


def gather_scalars(dist, x_local: float, device="cpu"):
    """[x of rank 0, x of rank 1, ...] on every rank (reporting: per-rank frame pairs/s next to the whole-job value)."""
    if dist is None:
        return [float(x_local)]
    import torch
    t = torch.tensor([float(x_local)], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_poses(dist, poses_local, device="cpu"):
    """All-gather per-rank [n,16] pose blocks to every rank (optional reporting path)."""
    import torch
    t = torch.as_tensor(poses_local, dtype=torch.float64, device=device).contiguous()
    if dist is None:
        return [t.cpu().numpy()]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy() for o in out]
