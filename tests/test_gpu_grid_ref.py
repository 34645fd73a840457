"""HIP vs REFERENCE goldens, not through the oracle: the device-side GridStructure (point_cells_kernel), LineIterator
rasteriser (line_cells_kernel) and GridStructure::get window gather (grid_cover) of the stvo_seq_* pipeline are read back
through the stvo_seq_debug_grid test hook and compared cell for cell / candidate for candidate with
tests/golden/ref_device_grid_goldens.npz, which tests/golden/gen_ref_device_grid_goldens.py produced from the reference's
own src/gridStructure.cpp + src/lineIterator.cpp (compiled unmodified).  Three sequences side by side, one per KITTI
calibration / image size of BASELINE configs[4] (per-sequence grid scale through stvo_seq_create_multi)."""
import os

import numpy as np
import pytest

from stvo_amd import synth
from stvo_amd.ctypes_types import match_params, opt_params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_device_grid_goldens.npz")
CAMS = [synth.KITTI_CAM, synth.KITTI03_CAM, synth.KITTI04_CAM]


def test_device_grid_rasteriser_and_window_gather_vs_reference_goldens():
    from stvo_amd import capi
    g = np.load(GOLD)
    assert [tuple(s) for s in g["sizes"]] == [(c["width"], c["height"]) for c in CAMS]
    rng = np.random.default_rng(5)
    frames = []
    for c in range(3):
        kp_l, kp_r, kl_l, kl_r = g[f"kp_l_{c}"], g[f"kp_r_{c}"], g[f"kl_l_{c}"], g[f"kl_r_{c}"]
        frames.append(dict(kp_l=kp_l, oct_l=np.zeros(len(kp_l), np.int32), desc_l=synth.random_desc(rng, len(kp_l)),
                           kp_r=kp_r, desc_r=synth.random_desc(rng, len(kp_r)),
                           kl_l=kl_l, oct_ll=np.zeros(len(kl_l), np.int32), ldesc_l=synth.random_desc(rng, len(kl_l)),
                           kl_r=kl_r, ldesc_r=synth.random_desc(rng, len(kl_r))))
    mp = match_params("kitti"); op = opt_params("kitti")
    assert mp.matching_s_ws == int(g["ws"])
    ctx = capi.Context(device_id=0, max_rows=1024, max_batch=3)
    dev = capi.Sequences(ctx, 3, 1024, 128, CAMS, mp, op)
    try:
        dev.push(frames)
        for c in range(3):
            # ---- key-points
            start, items, cells_l, coff, cand = dev.debug_grid(c, lines=False)
            off, out = g[f"pcell_off_{c}"], g[f"pcell_out_{c}"]
            assert np.array_equal(start, off), f"sequence {c}: point cell_start differs from the reference grid"
            for cell in range(64 * 48):
                assert np.array_equal(np.sort(items[start[cell]:start[cell + 1]]), out[off[cell]:off[cell + 1]]), (c, cell)
            assert np.array_equal(cells_l, g[f"pcells_l_{c}"])
            assert np.array_equal(coff, g[f"pcand_off_{c}"]) and np.array_equal(cand, g[f"pcand_out_{c}"]), f"sequence {c}: point window gather"
            # ---- key-lines (every Bresenham cell of every right line)
            start, items, cells_l, coff, cand = dev.debug_grid(c, lines=True)
            off, out = g[f"lcell_off_{c}"], g[f"lcell_out_{c}"]
            assert np.array_equal(start, off), f"sequence {c}: line cell_start differs from the reference grid"
            for cell in range(64 * 48):
                assert np.array_equal(np.sort(items[start[cell]:start[cell + 1]]), out[off[cell]:off[cell + 1]]), (c, cell)
            assert np.array_equal(cells_l, g[f"lcells_l_{c}"])
            assert np.array_equal(coff, g[f"lcand_off_{c}"]) and np.array_equal(cand, g[f"lcand_out_{c}"]), f"sequence {c}: line window gather"
    finally:
        dev.close()
        ctx.close()
