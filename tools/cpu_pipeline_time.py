import sys, time
sys.path.insert(0, "stvo-pl_amd/python"); sys.path.insert(0, "tests")
import numpy as np
from stvo_amd import synth
from stvo_amd.ctypes_types import match_params, opt_params
import oracle_lib, pipeline_ref
orc = oracle_lib.load()
for nl, has_l in ((0, 0), (85, 1)):
    frames = synth.make_stereo_sequence(synth.SEED0, n_frames=21, n_pts=1650, n_lines=nl)
    t = time.perf_counter()
    ref = pipeline_ref.run_sequence(orc, frames, synth.KITTI_CAM, match_params("kitti"), opt_params("kitti", has_lines=has_l))
    dt = (time.perf_counter() - t) / 21  # 21 stereo associations, 20 f2f + pose
    print(f"oracle (1 core) full per-frame pipeline, lines={nl}: {dt*1e3:.2f} ms/frame")
