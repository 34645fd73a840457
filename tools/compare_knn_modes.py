"""Developer check: the matrix-core matcher (K1m) against the VALU matcher (K1 + K1v) on the bench workload.
    STVO_KNN_MFMA=0 python tools/compare_knn_modes.py dump /tmp/a.npz ; STVO_KNN_MFMA=2 python tools/compare_knn_modes.py dump /tmp/b.npz
    python tools/compare_knn_modes.py cmp /tmp/a.npz /tmp/b.npz"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stvo-pl_amd", "python"))
import numpy as np

if sys.argv[1] == "dump":
    import torch
    from stvo_amd import capi, synth
    from stvo_amd.ctypes_types import opt_params
    from stvo_amd.devbatch import TrackBatch
    B = int(os.environ.get("CMP_B", "64")); n = 2000
    frames = [synth.make_f2f_points(synth.frame_seed(0, k), n=n) for k in range(B)]
    batch = TrackBatch(frames, max_pts=2048, max_lines=0, device="cuda:0")
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.track_batched(batch, synth.KITTI_CAM, opt_params("kitti", has_lines=0), 0.75, 0.75, 1)
    ctx.synchronize(); torch.cuda.synchronize()
    res = batch.results()
    np.savez(sys.argv[2], m12=batch.m12_pts(), T=res["T"], status=res["status"], iters=res["iters"], n_inl=res["n_inliers_pt"])
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = (a["m12"] == b["m12"])
    print("m12 identical:", bool(same.all()), "differing entries:", int((~same).sum()), "of", same.size)
    if not same.all():
        bad = np.argwhere(~same)[:10]
        for f, i in bad:
            print("  frame", f, "row", i, a["m12"][f, i], b["m12"][f, i])
    print("matches per frame (a, b):", (a["m12"] >= 0).sum(1)[:8], (b["m12"] >= 0).sum(1)[:8])
    print("iters a:", a["iters"][:8], "b:", b["iters"][:8], " max |dT|:", float(np.abs(a["T"] - b["T"]).max()))
