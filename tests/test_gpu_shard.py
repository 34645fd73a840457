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
