"""Does splitting the 1024 streams of the headline into independent halves on two streams help?  (round 5 experiment)
   python tools/two_pipes.py [n_pipes] [total_streams]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))
import bench
from stvo_amd import capi, synth
from stvo_amd.ctypes_types import match_params, opt_params
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
TOTAL = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
S, steps = 4, 24
B = TOTAL // P
seq_ids, replicas = bench.stream_ids(0, 1, TOTAL)
streams = bench.generate_streams(seq_ids, replicas, S, 1650, 85)
mp, op = match_params("kitti"), opt_params("kitti")
ctxs, pipes = [], []
for p in range(P):
    sl = slice(p * B, (p + 1) * B)
    cams = [synth.config5_cam(int(s)) for s in seq_ids[sl]]
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
    pipe = capi.Sequences(ctx, B, 2048, 128, cams, mp, op)
    pipe.set_slots(S)
    for k in range(S):
        pipe.upload(k, [st[k] for st in streams[sl]])
    ctx.synchronize()
    ctxs.append(ctx); pipes.append(pipe)
orders = [bench.ping_pong(S) for _ in range(P)]
def run(n):
    for _ in range(n):
        for p in range(P):
            pipes[p].step_dev(next(orders[p]))
    for c in ctxs: c.synchronize()
run(6)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); run(steps); ts.append(time.perf_counter() - t0)
dt = float(np.median(ts))
print(f"{P} pipe(s) x {B} streams: {TOTAL * steps / dt:.0f} frame pairs/s, {dt / steps * 1e3:.4f} ms per {TOTAL}-stream step")
res, _ = pipes[0].read(); print("committed", float((res["status"] == 0).mean()))
for p in pipes: p.close()
for c in ctxs: c.close()
