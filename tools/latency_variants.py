#!/usr/bin/env python3
"""Single-stream stvo_seq_push latency under the developer switches of the step (which launches a stage uses):
    python tools/latency_variants.py        -> one line per variant, KITTI-shaped (102 key-lines) and EuRoC-shaped (300 key-lines)"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(seq, cam, preset, n=61):
    import ctypes as C
    from stvo_amd import capi
    from stvo_amd.ctypes_types import POSE_RESULT_DTYPE, match_params, opt_params
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=1)
    dev = capi.Sequences(ctx, 1, 2048, 512, cam, match_params(preset), opt_params(preset))
    packed = [dev._pack([fr]) for fr in seq]
    res = np.zeros(1, dtype=POSE_RESULT_DTYPE); counts = np.zeros(4, np.int32)
    ts = []
    for ff, keep in packed:
        t0 = time.perf_counter()
        ctx._chk(ctx.lib.stvo_seq_push(dev.h, C.byref(ff), res.ctypes.data_as(C.c_void_p), counts))
        ts.append(time.perf_counter() - t0)
    dev.close(); ctx.close()
    return float(np.median(np.array(ts[10:]) * 1e3))


def main():
    import torch  # noqa: F401
    from stvo_amd import synth
    kitti = synth.make_stereo_sequence(synth.frame_seed(77, 0), n_frames=61, n_pts=1650, n_lines=85, cam=synth.KITTI_CAM)
    euroc = synth.make_stereo_sequence(synth.frame_seed(78, 0), n_frames=61, n_pts=660, n_lines=250, cam=synth.EUROC_CAM,
                                       depth=(0.5, 8.0), octave_probs=[.5, .25, .15, .1], outlier_frac=0.4)
    keys = ["STVO_LINE_FORK", "STVO_LINE_FUSED", "STVO_MATCH_SMALL", "STVO_GRID_TAIL"]
    variants = [dict(), dict(STVO_LINE_FORK="early"), dict(STVO_LINE_FORK="early", STVO_LINE_FUSED="0"),
                dict(STVO_LINE_FORK="early", STVO_MATCH_SMALL="0"), dict(STVO_LINE_FORK="early", STVO_LINE_FUSED="0", STVO_MATCH_SMALL="0"),
                dict(STVO_LINE_FORK="early", STVO_GRID_TAIL="0"),
                dict(STVO_LINE_FORK="early", STVO_LINE_FUSED="0", STVO_MATCH_SMALL="0", STVO_GRID_TAIL="0")]
    for v in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(v)
        a = run(kitti, synth.KITTI_CAM, "kitti"); b = run(euroc, synth.EUROC_CAM, "euroc")
        print(f"{str(v):100s} kitti {a:.4f} ms   euroc {b:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
