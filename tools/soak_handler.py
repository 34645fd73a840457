import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "stvo-pl_amd/python")
import numpy as np, pathlib, tempfile
import oracle_lib, pipeline_ref, test_gpu_handler as th
from stvo_amd import synth
from stvo_amd.ctypes_types import match_params, opt_params
orc = oracle_lib.Oracle(oracle_lib.load().lib) if hasattr(oracle_lib, "Oracle") else oracle_lib.load()
cam = synth.KITTI_CAM
frames = synth.make_stereo_sequence(4242, n_frames=40, n_pts=1200, n_lines=80, cam=cam)
tmp = pathlib.Path(tempfile.mkdtemp())
for pipe in (True, False):
    res, _ = th.run_app(tmp, frames, cam, "kitti", pipeline=pipe)
    ref = pipeline_ref.run_sequence(orc, frames, cam, match_params("kitti"), opt_params("kitti"))
    th.compare(res, ref)
    print("soak ok, pipeline =", pipe, len(res), "frames; final Tfw translation", np.round(res[-1]["Tfw"].reshape(4,4)[:3,3], 3))
