#!/usr/bin/env python3
"""bench.py — stereo frame-pairs/s through the hot path {f2f brute-force match + optimizePose} on MI355X.

One "step" = one pass of the hot path over one batch of B synthetic frame pairs that are already
resident in HBM (BASELINE.json configs[1]: synthetic 1241x376 stereo, ~2000 ORB key-points per frame,
brute-force point match + optimizePose, KITTI parameters).  Prints ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: independent sequences are sharded one batch per rank (weak scaling); the only collective
is the timing barrier / max-reduction over RCCL — the path itself has no exchange step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# K1 issues, per (query, train) pair, 8 full-rate VALU ops (v_xor_b32: 32 lanes/clk/SIMD) and 11 half-rate
# ones (v_bcnt_u32_b32, v_lshl_or_b32, v_med3_u32, v_min_u32: 16 lanes/clk/SIMD, measured with
# tools/valu_rates.hip).  Roof of that mix at 256 CU x 4 SIMD x 2.4 GHz: 19 / (8/78.6e12 + 11/39.3e12).
K1_LANE_OPS_PER_PAIR = 19
VALU_PEAK_LANE_OPS = K1_LANE_OPS_PER_PAIR / (8 / 78.6e12 + 11 / 39.3e12)  # = 49.8e12


# K1m (the default matcher) takes the 256-bit distances from the matrix cores: 2 x 256 int8 multiply-accumulate ops per
# (query, train) pair.  Dense int8 peak = 2x the bf16 dense peak (MI355X_MICROARCH.md: bf16 ~2.5 PF dense, "i8 ~2x bf16
# rate (2xK)"; the guide's own micro-benchmark floor for v_mfma_i32_32x32x32_i8 is 4404 TOP/s).
I8_MFMA_PEAK_TOPS = 5000.0
I8_MFMA_MEASURED_FLOOR_TOPS = 4404.0
K1M_OPS_PER_PAIR = 2 * 256


def committed_traffic(kernel="hamming_knn2_kernel", path=os.path.join(ROOT, "profiles", "r01_f_hbm_counters.txt")):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate
    `--pmc` runs summarised by tools/rocprof_summary.py; values there are KB per dispatch).  None if unavailable."""
    try:
        tot = {}
        for line in open(path):
            f = line.split()
            if len(f) >= 6 and f[4] in ("FETCH_SIZE", "WRITE_SIZE") and kernel in line:
                tot[f[4]] = float(f[1]) * 1024.0
        return tot["FETCH_SIZE"] + tot["WRITE_SIZE"] if len(tot) == 2 else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(frames, prm, budget_s=12.0):
    """The oracle (scalar C port of the reference path) timed on this box's host cores, 1 thread,
    on a bounded sample of the same workload.  Checker code: used here ONLY as the CPU baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from stvo_amd import synth
    orc = oracle_lib.load()
    z3, z2 = np.zeros((0, 3)), np.zeros((0, 2))
    done, t0 = 0, time.perf_counter()
    # ~10-15 s of CPU work: the batch of this step, repeated until the time budget is used up
    for fr in (frames[i % len(frames)] for i in range(4 * len(frames))):
        m12, _ = orc.match(fr["prev_desc"], fr["curr_desc"], 0.75, 1)
        sel = np.nonzero(m12 >= 0)[0]
        rec = dict(P=fr["prev_P"][sel], pl_obs=fr["curr_pl"][m12[sel]], sigma2p=fr["prev_sigma2"][sel],
                   inlier_p=np.ones(len(sel), np.int32), sP=z3, eP=z3, le_obs=z3, spl=z2, epl=z2, sigma2l=np.zeros(0),
                   inlier_l=np.zeros(0, np.int32))
        orc.optimize_pose(np.eye(4), synth.KITTI_CAM, prm, rec)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{done} frame pairs of the same workload (the step's batch, cycled), oracle/stvo_oracle.c (-O3), {dt:.1f} s, "
                      f"host has {os.cpu_count()} cores"}


def cpu_baseline_variants(frames, prm, budget_s=5.0):
    """Two more CPU figures asked for by SURVEY.md §8(d), same oracle, same workload: (a) the reference's own thread
    fan-out for this config — matches_12 and matches_21 on two threads (lrInParallel, src/matching.cpp:68-78), the
    optimisation single-threaded; (b) independent frame pairs on many host threads (the oracle's C functions run
    outside the GIL).  Reported next to `cpu_baseline`, never instead of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    import oracle_lib
    from stvo_amd import synth
    orc = oracle_lib.load()
    z3, z2 = np.zeros((0, 3)), np.zeros((0, 2))

    def pose(fr, m12):
        sel = np.nonzero(m12 >= 0)[0]
        rec = dict(P=fr["prev_P"][sel], pl_obs=fr["curr_pl"][m12[sel]], sigma2p=fr["prev_sigma2"][sel],
                   inlier_p=np.ones(len(sel), np.int32), sP=z3, eP=z3, le_obs=z3, spl=z2, epl=z2, sigma2l=np.zeros(0),
                   inlier_l=np.zeros(0, np.int32))
        orc.optimize_pose(np.eye(4), synth.KITTI_CAM, prm, rec)

    out = {}
    with ThreadPoolExecutor(2) as ex:  # (a)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            fr = frames[done % len(frames)]
            f21 = ex.submit(orc.match_nnr, fr["curr_desc"], fr["prev_desc"], 0.75)
            m12, _ = orc.match_nnr(fr["prev_desc"], fr["curr_desc"], 0.75)
            m21, _ = f21.result()
            ok = (m12 >= 0) & (m21[np.maximum(m12, 0)] == np.arange(len(m12)))  # src/matching.cpp:80-86
            pose(fr, np.where(ok, m12, -1))
            done += 1
        dt = time.perf_counter() - t0
    out["cpu_baseline_fanout"] = {"value": done / dt, "unit": "frame-pairs/s", "cores": 2, "kind": "port",
                                  "sample": f"{done} frame pairs, 12 || 21 matching on two threads like the reference, {dt:.1f} s"}
    nthr = min(32, os.cpu_count() or 1)

    def one(i):
        fr = frames[i % len(frames)]
        m12, _ = orc.match(fr["prev_desc"], fr["curr_desc"], 0.75, 1)
        pose(fr, m12)

    with ThreadPoolExecutor(nthr) as ex:  # (b)
        n_jobs = max(2 * nthr, int(budget_s * 90 * nthr / 2))
        t0 = time.perf_counter()
        list(ex.map(one, range(n_jobs)))
        dt = time.perf_counter() - t0
    out["cpu_baseline_threads"] = {"value": n_jobs / dt, "unit": "frame-pairs/s", "cores": nthr, "kind": "port",
                                   "sample": f"{n_jobs} independent frame pairs on {nthr} host threads, {dt:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frame pairs per step per GPU")
    ap.add_argument("--keypoints", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="run the pose kernel of batch s on a second stream beside the matching kernels of batch s+1 (+2 %%; "
                         "both kernels then share the CUs and per-kernel durations are no longer those of the kernel alone)")
    ap.add_argument("--no-overlap", action="store_true", help="accepted for compatibility: strict stream order is the default")
    args = ap.parse_args()

    import torch
    from stvo_amd import capi, synth
    from stvo_amd.ctypes_types import opt_params
    from stvo_amd.devbatch import TrackBatch

    from stvo_amd import shard
    world, rank, local_rank = shard.env_world()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    B, n = args.batch, args.keypoints
    max_pts = 2048 if n <= 2048 else None
    if max_pts is None:
        raise SystemExit("key-points per frame exceed STVO_POSE_MAX_POINTS (2048)")
    # rank r processes sequence r (SURVEY.md §8d config 5 sharding: sequence s -> GPU s mod G)
    frames = [synth.make_f2f_points(synth.frame_seed(rank, k), n=n) for k in range(B)]
    batch = TrackBatch(frames, max_pts=max_pts, max_lines=0, device=dev)
    prm = opt_params("kitti", has_lines=0)
    ctx = capi.Context(device_id=local_rank, max_rows=max_pts, max_batch=B)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    overlap = args.overlap and not args.no_overlap
    ctx.set_overlap(overlap)

    def step():
        ctx.track_batched(batch, synth.KITTI_CAM, prm, 0.75, 0.75, 1)

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()           # both of the context's streams
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # whole-job frame pairs (sum over ranks) and the slowest rank's time (max over ranks)
    frames_total, dt = shard.aggregate(dist, B * args.steps, dt, device=dev)

    # sanity: the timed work produced real poses
    res = batch.results()
    ok_frac = float((res["status"] == 0).mean())

    out = None
    if rank == 0:
        # The dominant kernel is the forward top-2 scan (K1m hamming_knn2_mfma_kernel<2, 0>, or K1 hamming_knn2_kernel with
        # STVO_KNN_MFMA=0): ONE launch per step, every prev row against every curr row.  The reverse (mutual) check is a
        # per-frame planning kernel plus two sparse scans of the same kernel family.
        # K1m is timed LIVE in a second pass over the same steps (same batch, same stream order): hipEvent pairs
        # around every launch, on the stream it is launched on.  The pass is separate from the one that produced
        # `value` because the event markers cost throughput (each is a barrier packet).
        ctx.set_kernel_timing(True)
        for _ in range(args.steps):
            step()
        ctx.synchronize()
        k1_ms, verify_ms, k1_calls = ctx.get_kernel_timing()
        ctx.set_kernel_timing(False)
        k1_solo_ms = ctx.time_stage(batch, synth.KITTI_CAM, prm, 0.75, 0, 10)  # same launch with the GPU to itself
        verify_solo_ms = ctx.time_stage(batch, synth.KITTI_CAM, prm, 0.75, 3, 10)
        pose_ms = ctx.time_stage(batch, synth.KITTI_CAM, prm, 0.75, 1, 10)
        n1v = batch.host["n_prev_pts"].astype(np.int64); n2v = batch.host["n_curr_pts"].astype(np.int64)
        nsel = ctx.last_reverse_counts(B).astype(np.int64)
        alg_bytes = float((32 * (n1v + n2v) + 8 * n1v).sum())   # descriptors in once, one packed top-2 per prev row out
        pairs = float((n1v * n2v).sum())                        # distance evaluations per launch
        achieved_gbs = alg_bytes / (k1_ms * 1e-3) / 1e9
        lane_ops = pairs * K1_LANE_OPS_PER_PAIR
        mfma = os.environ.get("STVO_KNN_MFMA", "2") != "0" and max_pts <= 8192   # the library's own rule (knn_mfma_qb)
        k1_name = "hamming_knn2_mfma_kernel<2, 0>" if mfma else "hamming_knn2_kernel"
        out = {
            "metric": "stereo frames/s (match+optimizePose)", "value": frames_total / dt, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8+f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 1241x376 stereo, 2000 ORB key-points/frame, "
                                   "brute-force mutual-NNR point match + optimizePose (config_kitti.yaml), "
                                   "batched independent frame pairs resident in HBM",
                       "frame_pairs_per_step_per_gpu": B, "keypoints_per_frame": n, "parallelism": f"seq-shard x{world}",
                       "committed_pose_fraction": ok_frac,
                       "pose_overlaps_next_match": overlap},
            "roofline": None, "hbm_view": None, "valu_roofline": None,
            "stage_ms": {"hamming_knn2": k1_ms, "reverse_check": verify_ms, "hamming_knn2_calls_timed": k1_calls,
                         "hamming_knn2_launches_per_step": 1, "hamming_knn2_solo": k1_solo_ms,
                         "reverse_scans_solo": verify_solo_ms, "pose_solo": pose_ms,
                         "claimed_column_fraction": float(nsel.sum()) / float(n2v.sum())},
        }
        traffic = committed_traffic(k1_name)
        traffic_src = ("bytes per launch = FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes, read from the committed "
                       "profiles/r01_f_hbm_counters.txt (not re-measured by this run)")
        timing = "hipEvent pairs around each launch, on the launch stream, over a second pass of the same K steps"
        hbm_view = {"kernel": k1_name, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "note": "brute-force matching does ~56 k distance bit-operations per compulsory byte: HBM is idle by construction"}
        if mfma:
            plan = ctx.last_reverse_plan(B).astype(np.int64)   # [claimed, light, heavy, |S|, tau] per frame pair
            out["stage_ms"]["reverse_plan"] = {
                "light_column_fraction": float(plan[1].sum()) / max(float(plan[0].sum()), 1.0),
                "heavy_column_fraction": float(plan[2].sum()) / max(float(plan[0].sum()), 1.0),
                "mean_rows_in_S": float(plan[3].mean()), "mean_tau": float(plan[4].mean()),
                "reverse_distance_evaluations_per_frame": float((plan[1] * plan[3] + plan[2] * n1v).mean()),
                "note": "reverse check: light columns are scanned against the |S| rows whose second-best forward distance is "
                        "within the cut tau, heavy columns against all rows (DESIGN.md §5)"}
            ops = pairs * K1M_OPS_PER_PAIR
            achieved_tops = ops / (k1_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": k1_name, "bound": "mfma", "achieved": achieved_tops, "peak": I8_MFMA_PEAK_TOPS,
                               "unit": "TFLOP/s", "unit_note": "int8 multiply-accumulate ops (TOP/s); 2 x 256 per distance",
                               "frac": achieved_tops / I8_MFMA_PEAK_TOPS, "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_ops_per_launch": ops, "algorithmic_bytes_per_launch": alg_bytes,
                               "avg_launch_ms": k1_ms, "avg_launch_ms_solo": k1_solo_ms, "timing": timing,
                               "frac_of_measured_mfma_floor": achieved_tops / I8_MFMA_MEASURED_FLOOR_TOPS,
                               "note": "K1m: all-pairs Hamming distance as an int8 Gram matrix on the matrix cores, top-2 fold "
                                       "(2 VALU ops per pair) in the shadow of the matrix instructions; one launch per step"}
            out["hbm_view"] = hbm_view
            del out["valu_roofline"]
        else:
            valu_meas = ctx.valu_peak()
            out["roofline"] = dict(hbm_view, bound="hbm", avg_launch_ms=k1_ms, timing=timing,
                                   note="K1 is integer-VALU bound (~530 lane-ops per compulsory byte); see valu_roofline")
            out["valu_roofline"] = {"kernel": k1_name, "lane_ops_per_launch": lane_ops,
                                    "achieved": lane_ops / (k1_ms * 1e-3) / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12,
                                    "unit": "T lane-ops/s", "frac": lane_ops / (k1_ms * 1e-3) / VALU_PEAK_LANE_OPS,
                                    "measured_peak_same_mix": valu_meas / 1e12,
                                    "frac_of_measured_peak": lane_ops / (k1_ms * 1e-3) / valu_meas}
            del out["hbm_view"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, prm)
            out.update(cpu_baseline_variants(frames, prm))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
