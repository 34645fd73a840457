#!/usr/bin/env python3
"""ORB point front-end: ms per launch of stvo_orb_detect_levels_dev on B KITTI-size images resident in HBM, without PyTorch (device
buffers through the HIP runtime via ctypes: starts in a second on a fresh box, tools/bench_orb.py imports torch).  Checks the first
images against the oracle.   python tools/orb_probe.py [--batch 256] [--iters 20]"""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
from stvo_amd import capi, synth
import oracle_lib
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]


def dev(nbytes):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), nbytes) == 0
    return p


B, K, cols, rows = a.batch, 2048, 1241, 376
base = [synth.make_image(500 + k) for k in range(8)]
imgs = np.ascontiguousarray(np.stack([np.roll(base[b % 8], 7 * (b // 8), axis=1) for b in range(B)]))
ctx = capi.Context(0, 2048, 4)
orb = capi.Orb(ctx, B, cols, rows, max_keypoints=K)
d_img, d_kp, d_resp, d_ang, d_desc, d_n = dev(imgs.nbytes), dev(B * K * 8), dev(B * K * 4), dev(B * K * 4), dev(B * K * 32), dev(B * 4)
assert hip.hipMemcpy(d_img, imgs.ctypes.data_as(C.c_void_p), imgs.nbytes, 1) == 0
run = lambda: orb.detect_dev(d_img, d_kp, d_resp, d_ang, d_desc, d_n)
for _ in range(3):
    run()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    run()
ctx.synchronize()
dt = (time.perf_counter() - t0) / a.iters
n = np.zeros(B, np.int32); desc = np.zeros((B, K, 32), np.uint8); kp = np.zeros((B, K, 2), np.float32)
hip.hipMemcpy(n.ctypes.data_as(C.c_void_p), d_n, n.nbytes, 2); hip.hipMemcpy(desc.ctypes.data_as(C.c_void_p), d_desc, desc.nbytes, 2)
hip.hipMemcpy(kp.ctypes.data_as(C.c_void_p), d_kp, kp.nbytes, 2)
o = oracle_lib.load()
ok = True
for b in range(min(B, 2)):
    ref = o.orb_detect(imgs[b], nfeatures=2000, fast_th=20)
    ok = ok and n[b] == len(ref["kp"]) and np.array_equal(kp[b, :n[b]], ref["kp"]) and np.array_equal(desc[b, :n[b]], ref["desc"])
print(f"{B} images: {dt * 1e3:.3f} ms per launch = {B / dt:.0f} images/s; mean key-points {n.mean():.0f}; first images equal to the oracle: {ok}")
for p in (d_img, d_kp, d_resp, d_ang, d_desc, d_n):
    hip.hipFree(p)
orb.close(); ctx.close()
