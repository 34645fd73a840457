import sys, os, numpy as np
sys.path.insert(0,'/root/repo/stvo-pl_amd/python'); sys.path.insert(0,'/root/repo/tests')
import torch
from stvo_amd import capi, synth
from stvo_amd.ctypes_types import opt_params
from stvo_amd.devbatch import TrackBatch
import oracle_lib
orc=oracle_lib.load()
B=64
for kw in [dict(cluster_frac=0.6, cluster_size=8, spread_p=0.06), dict(cluster_frac=0.5, cluster_size=6, spread_p=0.08), dict(cluster_frac=0.8, cluster_size=12, spread_p=0.05), dict(cluster_frac=0.4, cluster_size=4, spread_p=0.1)]:
    frames=[synth.make_f2f_points(synth.frame_seed(3,k), n=2000, desc_model="clustered", cluster_kw=kw) for k in range(B)]
    batch=TrackBatch(frames, max_pts=2048)
    prm=opt_params("kitti", has_lines=0)
    ctx=capi.Context(0, 2048, B); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.track_batched(batch, synth.KITTI_CAM, prm, 0.75, 0.75, 1); ctx.synchronize()
    plan=ctx.last_reverse_plan(B).astype(np.int64)
    m=batch.m12_pts()
    exp,_=orc.match(frames[0]["prev_desc"], frames[0]["curr_desc"], 0.75)
    print(kw, "claimed %.0f light %.0f heavy %.0f |S| %.0f tau %.1f matches %.0f ok %.2f parity %s" % (plan[0].mean(), plan[1].mean(), plan[2].mean(), plan[3].mean(), plan[4].mean(), (m>=0).sum(1).mean(), (batch.results()["status"]==0).mean(), np.array_equal(m[0,:2000], exp)))
    ctx.set_kernel_timing(True)
    for _ in range(5): ctx.track_batched(batch, synth.KITTI_CAM, prm, 0.75, 0.75, 1)
    print("   fwd ms %.4f  reverse ms %.4f" % ctx.get_kernel_timing()[:2])
    ctx.close()
