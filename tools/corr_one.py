"""One clustered-descriptor batch through stvo_track_batched_dev (f2f match + pose), for a kernel trace: tools/corr_prof.sh."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stvo-pl_amd", "python"))
import torch
from stvo_amd import capi, synth
from stvo_amd.ctypes_types import opt_params
from stvo_amd.devbatch import TrackBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kw = dict(cluster_frac=0.6, cluster_size=8, spread_p=0.06)
frames = [synth.make_f2f_points(synth.frame_seed(3, k % 64), n=2000, desc_model="clustered", cluster_kw=kw) for k in range(B)]
batch = TrackBatch(frames, max_pts=2048)
prm = opt_params("kitti", has_lines=0)
ctx = capi.Context(0, 2048, B); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3): ctx.track_batched(batch, synth.KITTI_CAM, prm, 0.75, 0.75, 1)
ctx.synchronize()
plan = ctx.last_reverse_plan(B).astype(np.int64)
print("claimed %.0f light %.0f heavy %.0f |S| %.0f tau %.1f" % tuple(plan[k].mean() for k in range(5)))
ctx.set_kernel_timing(True)
for _ in range(5): ctx.track_batched(batch, synth.KITTI_CAM, prm, 0.75, 0.75, 1)
print("fwd ms %.4f  reverse ms %.4f" % ctx.get_kernel_timing()[:2])
ctx.close()
