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
