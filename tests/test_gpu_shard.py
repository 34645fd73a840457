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
    assert abs(out["value"] - frames / (out["ms_per_step"] * 1e-3 * 2)) / out["value"] < 1e-4
    assert out["parity_sampled"] is None or out["parity_sampled"].get("ok", True)


def _run_bench(extra_args, env_extra=None, timeout=900):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.strip().splitlines()
    # stdout carries the one short line and nothing else of ours (gloo announces "[Gloo] Rank r is connected to .." there on its own,
    # the ranks' announcements interleaved)
    assert len([ln for ln in lines if ln.startswith("{")]) == 1, p.stdout[:2000]
    assert lines[-1].startswith("{") and len(lines[-1]) < 6000, len(lines[-1])  # (round 5: a 20 KB line was recorded by the driver as "parsed": null)
    return json.loads(lines[-1])


def test_bench_line_is_short_and_parses():
    """`python bench.py` (small batch) on the GPU: ONE stdout line, < 6 KB, with the contract's keys, a roofline measured by the light
    pass (event pairs on the three big kernels only) and the cpu_baseline object; the full record lands in bench_extras.json."""
    import json
    out = _run_bench(["--batch", "64", "--steps", "2", "--warmup", "1", "--repeats", "2", "--no-extras", "--no-clocks"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "roofline_pose", "roofline_grid_scan", "cpu_baseline", "parity_sampled"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["parity_sampled"]["ok"]
    for r in (out["roofline"], out["roofline_pose"], out["roofline_grid_scan"]):
        assert r["avg_launch_ms"] > 0 and r["achieved"] > 0 and 0 < r["frac"] < 1 and r["launches_timed"] == 2
    assert out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["value"] > 0
    full = json.load(open(os.path.join(ROOT, out["extras_file"])))
    assert full["value"] == pytest.approx(out["value"], rel=1e-5) and "stage_ms" in full and "note" in full["roofline"]


def test_bench_eight_ranks_gloo_on_one_gpu():
    """`python bench.py --gpus 8` as an 8-GPU node would run it (8 ranks, sequence s on rank s, 16 streams per rank), over gloo on this
    box's one GPU: n_gpus == 8, whole-job value, one rate per rank."""
    out = _run_bench(["--gpus", "8", "--batch", "16", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-extras", "--no-cpu-baseline", "--no-clocks"],
                     env_extra={"STVO_BENCH_BACKEND": "gloo", "STVO_N1_VALUE": "100000"})
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "weak"
    assert len(out["per_rank_frame_pairs_per_s"]) == 8 and all(v > 0 for v in out["per_rank_frame_pairs_per_s"])
    frames = 8 * 16 * 2
    assert abs(out["value"] - frames / (out["ms_per_step"] * 1e-3 * 2)) / out["value"] < 1e-4
    assert out["scaling_efficiency_vs_n1"] == pytest.approx(out["value"] / 8e5, rel=1e-4)
    assert out["parity_sampled"]["ok"]
