"""Pins the oracle's grid bucketing + Bresenham restatement against the REFERENCE's own code:
(a) live, against oracle/_ref/libstvo_ref.so (reference gridStructure.cpp + lineIterator.cpp compiled
unmodified by oracle/Makefile) when it is present, and (b) always, against the committed golden
vectors in tests/golden/ref_grid_goldens.npz that tests/golden/gen_ref_goldens.py produced from it."""
import os

import numpy as np
import pytest

import oracle_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_grid_goldens.npz")


def test_line_coords_vs_committed_goldens(oracle):
    g = np.load(GOLD)
    ends, off, cells = g["line_ends"], g["line_off"], g["line_cells"]
    for i in range(len(ends)):
        got = oracle.line_coords(*ends[i])
        assert np.array_equal(got, cells[off[i]:off[i + 1]]), (i, ends[i])


def test_grid_get_vs_committed_goldens(oracle):
    g = np.load(GOLD)
    for c in range(int(g["n_cases"])):
        ent, owner, q, w = g[f"ent_{c}"], g[f"owner_{c}"], g[f"q_{c}"], tuple(int(v) for v in g[f"w_{c}"])
        off, out = g[f"off_{c}"], g[f"out_{c}"]
        start, items = oracle.grid_build(ent, owner)
        n2 = int(owner.max()) + 1
        for k in range(len(q)):
            got = np.sort(oracle.window_gather(start, items, int(q[k, 0]), int(q[k, 1]), w, n2))
            assert np.array_equal(got, out[off[k]:off[k + 1]]), (c, k)


def test_live_reference_build_matches_oracle(oracle):
    ref = oracle_lib.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libstvo_ref.so not built (reference not mounted)")
    rng = np.random.default_rng(11)
    buf = np.empty((512, 2), np.int32)
    for _ in range(3000):
        x1, y1, x2, y2 = rng.uniform(-3, 67), rng.uniform(-3, 51), rng.uniform(-3, 67), rng.uniform(-3, 51)
        if rng.random() < 0.2:
            x2 = x1 + rng.uniform(-0.5, 0.5)
        if rng.random() < 0.2:
            y2 = y1
        n = ref.ref_line_coords(x1, y1, x2, y2, buf.reshape(-1), 512)
        assert np.array_equal(oracle.line_coords(x1, y1, x2, y2), buf[:n])


DEV_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_device_grid_goldens.npz")


def test_oracle_vs_pixel_space_goldens_three_kitti_calibrations(oracle):
    """The oracle's cell computation, bucket grid, Bresenham and window gather against reference outputs generated from
    PIXEL coordinates at the three KITTI image sizes of BASELINE configs[4] (1241x376, 1242x375, 1226x370)."""
    g = np.load(DEV_GOLD)
    ws = int(g["ws"])
    for c, (cols, rows) in enumerate(g["sizes"]):
        inv = np.array([64.0 / cols, 48.0 / rows])
        kp_l, kp_r = g[f"kp_l_{c}"], g[f"kp_r_{c}"]
        ent = (kp_r.astype(np.float64) * inv).astype(np.int32)
        start, items = oracle.grid_build(ent)
        off, out = g[f"pcell_off_{c}"], g[f"pcell_out_{c}"]
        for cell in range(64 * 48):
            assert np.array_equal(np.sort(items[start[cell]:start[cell + 1]]), out[off[cell]:off[cell + 1]]), (c, cell)
        q = g[f"pcells_l_{c}"]
        assert np.array_equal((kp_l.astype(np.float64) * inv).astype(np.int32), q)
        off, out = g[f"pcand_off_{c}"], g[f"pcand_out_{c}"]
        for i in range(len(q)):
            got = np.sort(oracle.window_gather(start, items, int(q[i, 0]), int(q[i, 1]), (ws, 0, 0, 0), len(kp_r)))
            assert np.array_equal(got, out[off[i]:off[i + 1]]), (c, i)
        kl_r = g[f"kl_r_{c}"].astype(np.float64)
        loff, lcells = g[f"lline_off_{c}"], g[f"lline_cells_{c}"]
        for j in range(len(kl_r)):
            got = oracle.line_coords(kl_r[j, 0] * inv[0], kl_r[j, 1] * inv[1], kl_r[j, 2] * inv[0], kl_r[j, 3] * inv[1])
            assert np.array_equal(got, lcells[loff[j]:loff[j + 1]]), (c, j)
