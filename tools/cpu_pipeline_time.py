"""The oracle's time per frame for the whole per-frame pipeline (stereo association + f2f + optimizePose), one host core.
    python tools/cpu_pipeline_time.py            KITTI-shaped, points only and points + lines (BASELINE configs[0] / [2])
    python tools/cpu_pipeline_time.py --config4  EuRoC-shaped, line-heavy, optimiser modes 0 / 1 / 2 (BASELINE configs[3])"""
import sys, time
sys.path.insert(0, "stvo-pl_amd/python"); sys.path.insert(0, "tests")
import numpy as np
from stvo_amd import synth
from stvo_amd.ctypes_types import match_params, opt_params
import oracle_lib, pipeline_ref
orc = oracle_lib.load()
if "--config4" in sys.argv:
    frames = synth.make_stereo_sequence(synth.SEED0, n_frames=21, n_pts=660, n_lines=250, cam=synth.EUROC_CAM, depth=(0.5, 8.0),
                                        octave_probs=[.5, .25, .15, .1], outlier_frac=0.4)
    for mode in (0, 1, 2):
        t = time.perf_counter()
        pipeline_ref.run_sequence(orc, frames, synth.EUROC_CAM, match_params("euroc"), opt_params("euroc", mode=mode))
        dt = (time.perf_counter() - t) / 21
        print(f"oracle (1 core) full per-frame pipeline, EuRoC-shaped config 4, optimiser mode {mode}: {dt*1e3:.2f} ms/frame")
else:
    for nl, has_l in ((0, 0), (85, 1)):
        frames = synth.make_stereo_sequence(synth.SEED0, n_frames=21, n_pts=1650, n_lines=nl)
        t = time.perf_counter()
        ref = pipeline_ref.run_sequence(orc, frames, synth.KITTI_CAM, match_params("kitti"), opt_params("kitti", has_lines=has_l))
        dt = (time.perf_counter() - t) / 21  # 21 stereo associations, 20 f2f + pose
        print(f"oracle (1 core) full per-frame pipeline, lines={nl}: {dt*1e3:.2f} ms/frame")
