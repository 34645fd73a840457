#!/usr/bin/env python3
"""Throughput of the ORB point front-end (stvo_orb_detect_dev) on B KITTI-sized images resident in HBM, next to the oracle
(oracle/stvo_orb_oracle.c, one host core).  python tools/bench_orb.py [--batch 256] [--iters 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
import torch  # noqa: E402
from stvo_amd import capi, synth  # noqa: E402
import oracle_lib  # noqa: E402

B, K = a.batch, 2048
base = [synth.make_image(500 + k) for k in range(8)]
imgs = np.stack([np.roll(base[b % 8], 7 * (b // 8), axis=1) for b in range(B)])
ctx = capi.Context(0, 2048, 4)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
orb = capi.Orb(ctx, B, 1241, 376, max_keypoints=K)
d = dict(img=torch.from_numpy(imgs).cuda(), kp=torch.zeros(B, K, 2, device="cuda"), resp=torch.zeros(B, K, device="cuda"),
         ang=torch.zeros(B, K, device="cuda"), desc=torch.zeros(B, K, 32, dtype=torch.uint8, device="cuda"),
         n=torch.zeros(B, dtype=torch.int32, device="cuda"))


def run():
    ctx._chk(ctx.lib.stvo_orb_detect_dev(orb.h, d["img"].data_ptr(), d["kp"].data_ptr(), d["resp"].data_ptr(), d["ang"].data_ptr(),
                                         d["desc"].data_ptr(), d["n"].data_ptr()))


for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
orc = oracle_lib.load()
t1 = time.perf_counter()
for b in range(8):
    orc.orb_detect(imgs[b], cap=K)
dtc = (time.perf_counter() - t1) / 8
px = B * 1241 * 376
print(json.dumps({"images_per_launch": B, "ms_per_launch": dt * 1e3, "images_per_s": B / dt, "mean_keypoints": float(d["n"].float().mean()),
                  "image_bytes_GBps": px / dt / 1e9, "oracle_ms_per_image_1_core": dtc * 1e3, "speedup_vs_1_core": dtc / (dt / B)}))
orb.close(); ctx.close()
