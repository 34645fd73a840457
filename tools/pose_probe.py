#!/usr/bin/env python3
"""Developer probe: the pose kernel alone on B frame pairs of BASELINE configs[1] shape (already matched by one
stvo_track_batched_dev call), average launch time and — with STVO_POSE_PROF=1 — the per-phase tick breakdown.
    STVO_POSE_PROF=1 python tools/pose_probe.py [--batch 512] [--points 2000] [--lines 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--points", type=int, default=2000)
ap.add_argument("--lines", type=int, default=0)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--f32-obs", action="store_true", help="round the observations to float (what the per-frame pipeline feeds)")
a = ap.parse_args()
import torch  # noqa: E402
from stvo_amd import capi, synth  # noqa: E402
from stvo_amd.ctypes_types import opt_params  # noqa: E402
from stvo_amd.devbatch import TrackBatch  # noqa: E402

B = a.batch
if a.lines:
    frames = [synth.make_f2f_points_lines(synth.frame_seed(0, k), n=a.points, n_lines=a.lines) for k in range(B)]
else:
    frames = [synth.make_f2f_points(synth.frame_seed(0, k), n=a.points) for k in range(B)]
if a.f32_obs:
    import numpy as np
    for fr in frames:
        fr["curr_pl"] = fr["curr_pl"].astype(np.float32).astype(np.float64)
batch = TrackBatch(frames, max_pts=2048, max_lines=128 if a.lines else 0)
prm = opt_params("kitti", has_lines=1 if a.lines else 0)
ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.track_batched(batch, synth.KITTI_CAM, prm, 0.75, 0.75, 1)
ctx.synchronize()
res = batch.results()
print(f"B={B}: matched {res['n_matched_pt'].mean():.0f} pts, iters stage1 {res['iters'][:, 0].mean():.2f} stage2 {res['iters'][:, 1].mean():.2f}, "
      f"ok {float((res['status'] == 0).mean()):.3f}")
ms = ctx.time_stage(batch, synth.KITTI_CAM, prm, 0.75, 1, a.iters)
print(f"pose kernel: {ms * 1e3:.1f} us per launch of {B} frame pairs")
ctx.close()
