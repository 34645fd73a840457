#!/usr/bin/env python3
"""LSD detector: HIP vs oracle on a few images — differences, timings.  python tools/lsd_probe.py [--batch 64] [--iters 3]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=2); ap.add_argument("--iters", type=int, default=2)
a = ap.parse_args()
from stvo_amd import capi, synth
import oracle_lib
o = oracle_lib.load()
cols, rows, B = 1241, 376, a.batch
imgs = np.stack([synth.make_image(500 + b) for b in range(min(B, 4))])
imgs = np.stack([imgs[b % len(imgs)] for b in range(B)])
ctx = capi.Context(0, 2048, 4)
lsd = capi.Lsd(ctx, B, cols, rows, capi.lsd_params(min_length=0.025 * rows, nfeatures=300), max_keylines=512)
import ctypes as C
ctx.lib.stvo_lsd_debug.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
ctx.lib.stvo_lsd_debug(lsd.h, 1, None)
segs, n = lsd.segments(imgs)
gd = np.zeros((8192, 8)); ctx.lib.stvo_lsd_debug(lsd.h, 1, gd.ctypes.data_as(C.c_void_p))
od = np.zeros((65536, 8)); o.lib.orc_lsd_set_debug.argtypes = [C.c_void_p]; o.lib.orc_lsd_set_debug(od.ctypes.data_as(C.c_void_p))
for b in range(min(B, 2)):
    ref = o.lsd_segments(imgs[b], o.lsd_opts())
    m = min(len(ref), len(segs[b]))
    bad = np.nonzero(np.any(segs[b][:m] != ref[:m], axis=1))[0]
    print(f"image {b}: {n[b]} vs {len(ref)} segments, {len(bad)} rows differ; first {bad[:5]}")
    for k in bad[:5]:
        print("   hip", segs[b][k], "ref", ref[k], "diff", segs[b][k] - ref[k])
        if b == 0:
            print("      hip", gd[k]); print("      ref", od[k]); print("      rel", (gd[k] - od[k]) / np.maximum(np.abs(od[k]), 1e-300))
    if b == 0:
        m2 = min(m, 8192)
        for c, name in enumerate(("cx", "cy", "Ixx", "Iyy", "Ixy", "theta", "l_min", "l_max")):
            print(f"   {name}: {int((gd[:m2, c] != od[:m2, c]).sum())} of {m2} differ")
        o.lib.orc_lsd_set_debug(None)
        q = gd[8190]
        q2 = gd[8191]
        if q2[0] == -1.0:
            print(f"   many-waves kernel, committer cycles: total {q[0]:.0f}  growing itself {q[1]:.0f}  waiting for an in-flight seed {q[2]:.0f}  validating + taking pending {q[3]:.0f} | "
                  f"regions taken {q[4]:.0f}  grown by the committer {q[5]:.0f}  failed validation {q[6]:.0f}  waited for {q[7]:.0f}")
            sp = gd[8189].view(np.uint64); dp = gd[8188]
            if sp[3] > 0:
                print(f"      speculating waves {sp[3]}: busy {sp[0]} of {sp[1]} cycles ({100.0 * sp[0] / max(sp[1], 1):.0f} %), regions {sp[2]} ({sp[0] / max(sp[2], 1):.0f} cycles each)")
                print(f"      dispatcher: passes {dp[0]:.0f} (no idle wave {dp[3]:.0f}), chunks scanned {dp[1]:.0f}, seeds handed out {dp[2]:.0f}, candidates too close to a growth in flight {dp[4]:.0f}")
            t5 = gd[8187]
            if t5[3] > 0:
                print(f"      committer: several-records passes incl. their preamble {t5[0]:.0f} ({t5[3]:.0f} tries)  queue peek of the one-by-one path {t5[1]:.0f}  refresh {t5[2]:.0f}")
            if q2[3] > 0:
                print(f"      regions through the feeder's records {q2[1]:.0f}  records passed over {q2[2]:.0f}  batches {q2[3]:.0f}  passes over several records {q2[4]:.0f} (undone {q2[5]:.0f})")
            continue
        print(f"   rounds: publish {q2[0]:.0f}  list read + address + issue {q2[1]:.0f}  wait for the loads {q2[2]:.0f}  | seed set-up {q2[3]:.0f}")
        print(f"   image 0, cycles: total {q[0]:.0f}  grow {q[1]:.0f} (of which resolving {q[2]:.0f})  rect {q[3]:.0f} | rounds {q[4]:.0f}  pixels added {q[5]:.0f}  regions {q[6]:.0f}  batches {q[7]:.0f}")
ts = []
for _ in range(a.iters):
    t0 = time.perf_counter(); lsd.detect(imgs); ts.append(time.perf_counter() - t0)
dt = float(np.median(ts))
print("   per call, ms:", " ".join(f"{t * 1e3:.1f}" for t in ts))
t0 = time.perf_counter(); o.lsd_detect(imgs[0], o.lsd_opts(min_length=0.025 * rows, nfeatures=300)); dc = time.perf_counter() - t0
print(f"{B} images: {dt * 1e3:.1f} ms per call incl. copies = {B / dt:.0f} images/s; oracle {dc * 1e3:.1f} ms per image on one core")
lsd.close(); ctx.close()
