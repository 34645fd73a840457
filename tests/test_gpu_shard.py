"""BASELINE configs[4] sharded over two ranks (sequence s -> rank s mod 2, one process per rank, each with a real
stvo_seq on the GPU; the only traffic between ranks is the reporting all-gather, as in bench.py) must reproduce the
1-rank run of all eight sequences pose for pose, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

import shard_worker
from stvo_amd import shard

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_config5_equals_one_rank(tmp_path):
    n_frames, n_pts, n_lines = 4, 500, 40
    one_p, one_s = shard_worker.run_rank(list(range(8)), n_frames, n_pts, n_lines)   # all eight sequences in one stvo_seq
    assert (one_s == 0).mean() > 0.9
    out = tmp_path / "two_rank.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "tests", "shard_worker.py"), str(n_frames), str(n_pts), str(n_lines), str(out)]
    subprocess.run(cmd, check=True, timeout=600, env=env, cwd=ROOT)
    z = np.load(out)
    assert int(z["world"]) == 2 and int(z["total"]) == 8 * (n_frames - 1)
    for r in range(2):
        ids = shard.sequences_for_rank(8, 2, r)
        got = z[f"poses_{r}"].reshape(len(ids), n_frames - 1, 16)
        assert got.tobytes() == one_p[ids].tobytes(), f"rank {r}: pose blocks differ from the 1-rank run"
        assert np.array_equal(z[f"status_{r}"].reshape(len(ids), -1).astype(np.int32), one_s[ids])


def test_bench_two_ranks_runs_its_multi_gpu_branch(tmp_path):
    """`python bench.py --gpus 2` end to end: the self-spawn under torch.distributed.run, one process per rank, sequence s on rank
    s mod 2, barrier + synchronize around the timed region, max-over-ranks time and the summed frame count — the code an 8-GPU node
    executes, here with STVO_BENCH_BACKEND=gloo because both ranks share this box's one GPU (RCCL refuses two ranks on one device).
    One JSON line from rank 0 with n_gpus == rccl_ranks == 2 and value = all ranks' frame pairs / the slowest rank's time."""
    import json
    env = dict(os.environ, STVO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "2", "--warmup", "1", "--repeats", "1",
           "--no-extras", "--no-cpu-baseline", "--no-clocks"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo"
    assert out["steps"] == 2 and out["scaling"] == "weak"
    # whole-job aggregate: 2 ranks x 64 streams x 2 steps over the max-over-ranks time
    frames = 2 * 64 * 2
    assert abs(out["value"] - frames / (out["ms_per_step"] * 1e-3 * 2)) / out["value"] < 1e-6
    assert out["parity_sampled"] is None or out["parity_sampled"].get("ok", True)
