import sys, os
sys.path.insert(0, "stvo-pl_amd/python")
import torch
from stvo_amd import capi, synth
from stvo_amd.ctypes_types import opt_params
from stvo_amd.devbatch import TrackBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = [synth.make_f2f_points(synth.frame_seed(0, k), n=2000) for k in range(B)]
batch = TrackBatch(frames, max_pts=2048)
prm = opt_params("kitti", has_lines=0)
for ov in (0, 1):
    ctx = capi.Context(0, 2048, B)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_overlap(ov)
    ctx.track_batched(batch, synth.KITTI_CAM, prm, 0.75, 0.75, 1); ctx.synchronize()
    r = {st: ctx.time_stage(batch, synth.KITTI_CAM, prm, 0.75, st, 10) for st in (2, 3, 4)}
    print(f"B={B} overlap_pad={ov}: forward {r[2]:.3f} ms  lazy-reverse {r[3]:.3f} ms  both-full {r[4]:.3f} ms  nsel mean {ctx.last_reverse_counts(B).mean():.0f}")
    ctx.close()
