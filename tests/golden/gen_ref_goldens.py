"""Generates tests/golden/ref_grid_goldens.npz from the REFERENCE's own code.

Runs only in the development container: needs oracle/_ref/libstvo_ref.so, which oracle/Makefile
builds from /root/reference/src/{gridStructure,lineIterator}.cpp (compiled unmodified, by path).
The fixture holds data only: inputs (seeded) and the reference's outputs.
    python tests/golden/gen_ref_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "stvo-pl_amd", "python"))
import oracle_lib  # noqa: E402


def main():
    ref = oracle_lib.load_ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20250227)
    out = {}
    # --- Bresenham (LineIterator / getLineCoords) ---
    ends = []
    for i in range(400):
        x1, y1, x2, y2 = rng.uniform(-2, 66), rng.uniform(-2, 50), rng.uniform(-2, 66), rng.uniform(-2, 50)
        kind = i % 8
        if kind == 1: x2 = x1                       # vertical
        if kind == 2: y2 = y1                       # horizontal
        if kind == 3: x2, y2 = x1 + 0.3, y1 - 0.2   # sub-cell
        if kind == 4: x1, y1, x2, y2 = np.floor([x1, y1, x2, y2])  # integer end points
        if kind == 5: x2, y2 = x1 + (y2 - y1), y2   # exact diagonal
        ends.append((x1, y1, x2, y2))
    # KITTI-shaped: pixel end points scaled by 64/1241, 48/376 exactly as stereoFrame.cpp:335 does
    for i in range(200):
        s = np.float32([rng.uniform(0, 1241), rng.uniform(0, 376)]); e = np.float32([rng.uniform(0, 1241), rng.uniform(0, 376)])
        iw, ih = 64 / 1241.0, 48 / 376.0
        ends.append((float(s[0]) * iw, float(s[1]) * ih, float(e[0]) * iw, float(e[1]) * ih))
    ends = np.array(ends, np.float64)
    buf = np.empty((1024, 2), np.int32)
    off = [0]; cells = []
    for x1, y1, x2, y2 in ends:
        n = ref.ref_line_coords(x1, y1, x2, y2, buf.reshape(-1), 1024)
        cells.append(buf[:n].copy()); off.append(off[-1] + n)
    out.update(line_ends=ends, line_off=np.array(off, np.int64), line_cells=np.concatenate(cells))
    # --- GridStructure::at / get ---
    cases = [(300, 250, (10, 0, 0, 0), 1.0), (500, 200, (10, 0, 0, 0), 0.2), (100, 400, (3, 2, 1, 1), 0.1),
             (50, 60, (0, 0, 0, 0), 1.0), (200, 200, (70, 70, 50, 50), 1.0)]
    for c, (n2, nq, w, spread) in enumerate(cases):
        mult = 1 + (c % 2) * 3  # some owners appear in several cells (rasterised lines)
        ent = np.stack([rng.integers(-2, int(66 * spread) + 1, n2 * mult), rng.integers(-2, int(50 * spread) + 1, n2 * mult)], 1).astype(np.int32)
        owner = np.repeat(np.arange(n2, dtype=np.int32), mult)
        q = np.stack([rng.integers(-3, 68, nq), rng.integers(-3, 52, nq)], 1).astype(np.int32)
        o = np.empty(nq + 1, np.int32); res = np.empty(nq * n2 + 16, np.int32)
        tot = ref.ref_grid_get(np.ascontiguousarray(ent).reshape(-1), owner.ctypes.data, len(ent), 48, 64, np.ascontiguousarray(q).reshape(-1),
                               nq, w[0], w[1], w[2], w[3], o, res, len(res))
        assert tot <= len(res)
        out.update({f"ent_{c}": ent, f"owner_{c}": owner, f"q_{c}": q, f"w_{c}": np.array(w, np.int32), f"off_{c}": o.copy(),
                    f"out_{c}": res[:tot].copy()})
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "ref_grid_goldens.npz"), **out)
    print("wrote ref_grid_goldens.npz:", len(ends), "lines,", len(cases), "grid cases")


if __name__ == "__main__":
    main()
