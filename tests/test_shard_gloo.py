"""N > 1 path on CPU: two processes over gloo exercise the sharding + aggregation that bench.py uses
with RCCL on the GPUs (the hot path itself has no collective: sequences are independent)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stvo_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = shard.sequences_for_rank(8, world, rank)
    frames_local = 1000 * len(seqs)
    seconds_local = 1.0 + 0.5 * rank  # the slowest rank sets the job time
    dist.barrier()
    total, tmax = shard.aggregate(dist, frames_local, seconds_local)
    poses = np.tile(np.eye(4).reshape(1, 16) * (rank + 1), (3, 1))
    gathered = shard.gather_poses(dist, poses)
    rates = shard.gather_scalars(dist, 1000.0 * (rank + 1))   # bench.py: per_rank_frame_pairs_per_s
    q.put((rank, seqs, total, tmax, [float(g[0, 0]) for g in gathered], rates))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, tot0, t0, g0, p0), (r1, s1, tot1, t1, g1, p1) = res
    assert p0 == p1 == [1000.0, 2000.0]                       # one rate per rank, in rank order, on every rank
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]          # sequence s -> rank s mod G
    assert tot0 == tot1 == 8000                               # whole-job frame pairs
    assert t0 == t1 == 1.5                                    # max over ranks
    assert g0 == g1 == [1.0, 2.0]


def test_single_process_identity():
    assert shard.aggregate(None, 123, 4.5) == (123, 4.5)
    assert shard.gather_scalars(None, 7.5) == [7.5]
    assert shard.sequences_for_rank(8, 1, 0) == list(range(8))
    assert sum(len(shard.sequences_for_rank(8, 4, r)) for r in range(4)) == 8
    assert shard.env_world()[0] >= 1
