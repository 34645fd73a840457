import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from stvo_amd import capi, synth
from stvo_amd.ctypes_types import match_params, opt_params
B, S = 320, 3
ids = np.arange(B) % 8
streams = [synth.make_config5_sequence(int(s), n_frames=S, n_pts=600, n_lines=40, replica=400 + b // 8) for b, s in enumerate(ids)]
cams = [synth.config5_cam(int(s)) for s in ids]
ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
dev = capi.Sequences(ctx, B, 2048, 128, cams, match_params("kitti"), opt_params("kitti"))
dev.set_slots(S)
for k in range(S):
    dev.upload(k, [st[k] for st in streams])
for i, cur in enumerate([0, 1, 2, 1, 0]):
    print("step", i, flush=True)
    dev.step_dev(cur)
    res, counts = dev.read()
    print("  ok", float((res["status"] == 0).mean()), flush=True)
dev.close(); ctx.close()
