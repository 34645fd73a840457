#!/usr/bin/env python3
"""Randomised parity of the batched device path of BASELINE configs[1] / [3] (stvo_track_batched_dev: f2f mutual match of points and
lines + on-device record gather + optimizePose for B frame pairs; up to 256 pairs the latency pose kernel, beyond that pose2p_kernel,
the general form with 48-byte records) against per-pair oracle runs: ragged batches from 1 to 2048 rows and 0 to 320 key-lines per
pair, presets, optimizer modes, ratios, outlier levels.  Test infrastructure.  Run on a GPU box from the repo root:
    python tests/fuzz_track_batched.py [--seconds 120] [--seed 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from fuzz_entry_points import pose_sensitivity
from stvo_amd import capi, synth
from stvo_amd.ctypes_types import opt_params

CAM = synth.KITTI_CAM


def main(argv=None):
    import torch
    from stvo_amd.devbatch import TrackBatch
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    args = ap.parse_args(argv)
    orc = oracle_lib.load()
    t_end = time.time() + args.seconds
    case = bad = pairs = ill = 0
    while time.time() < t_end and (args.cases == 0 or case < args.cases):
        case += 1
        rng = np.random.default_rng([args.seed, case])
        B = int(rng.choice([1, 3, 9, 9, 40, 300]))
        preset = str(rng.choice(["kitti", "euroc"])); mode = int(rng.choice([0, 0, 1, 2])); nnr = float(rng.choice([0.75, 0.9]))
        lines = bool(rng.integers(0, 2))
        hi = int(rng.choice([20, 300, 2048])); lhi = int(rng.choice([5, 100, 320]))
        frames = []
        for k in range(B):
            n = int(rng.integers(1, hi + 1)); nl = int(rng.integers(1, lhi + 1))
            kw = dict(octave_probs=None if preset == "kitti" else [.5, .25, .15, .1], outlier_frac=float(rng.choice([0.0, 0.15, 0.4])))
            seed = int(rng.integers(1, 1 << 30))
            frames.append(synth.make_f2f_points_lines(seed, n=n, n_lines=nl, **kw) if lines else synth.make_f2f_points(seed, n=n, **kw))
        tag = f"seed {args.seed} case {case}: B {B} {preset} mode {mode} nnr {nnr} lines {lines} rows<={hi} key-lines<={lhi}"
        prm = opt_params(preset, mode=mode, has_lines=1 if lines else 0)
        ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
        try:
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            batch = TrackBatch(frames, max_pts=2048, max_lines=320 if lines else 0)
            for rep in range(2):   # (the second call runs on the scratch the first one left)
                ctx.track_batched(batch, CAM, prm, nnr, nnr, 1)
            torch.cuda.synchronize()
            res = batch.results(); mp_all = batch.m12_pts(); ip_all = batch.inlier_pts()
            ml_all = batch.m12_lines() if lines else None; il_all = batch.inlier_lines() if lines else None
            for b, fr in enumerate(frames):
                pairs += 1
                m12, _ = orc.match(fr["prev_desc"], fr["curr_desc"], nnr)
                sel = np.nonzero(m12 >= 0)[0]
                z3 = np.zeros((0, 3)); z2 = np.zeros((0, 2))
                rec = dict(P=fr["prev_P"][sel], pl_obs=fr["curr_pl"][m12[sel]], sigma2p=fr["prev_sigma2"][sel], inlier_p=np.ones(len(sel), np.int32),
                           sP=z3, eP=z3, le_obs=z3, spl=z2, epl=z2, sigma2l=np.zeros(0), inlier_l=np.zeros(0, np.int32))
                sl = np.zeros(0, np.int64); m12l = None
                if lines:
                    m12l, _ = orc.match(fr["prev_ldesc"], fr["curr_ldesc"], nnr)
                    sl = np.nonzero(m12l >= 0)[0]
                    rec.update(sP=fr["prev_sP"][sl], eP=fr["prev_eP"][sl], le_obs=fr["curr_le"][m12l[sl]], spl=fr["prev_spl"][sl], epl=fr["prev_epl"][sl],
                               sigma2l=fr["prev_sigma2l"][sl], inlier_l=np.ones(len(sl), np.int32))
                ref = orc.optimize_pose(np.eye(4), CAM, prm, rec)
                n1 = len(fr["prev_P"])
                what = None
                if not np.array_equal(mp_all[b, :n1], m12) or (lines and not np.array_equal(ml_all[b, :len(fr["prev_sP"])], m12l)):
                    what = "match indices"
                else:
                    e = -np.ones(n1, np.int32); e[sel] = ref["inlier_p"]
                    course = (res["status"][b], res["path"][b], tuple(res["iters"][b])) == (ref["status"], ref["path"], ref["iters"]) and \
                        np.array_equal(ip_all[b, :n1], e)
                    if lines and course:
                        e = -np.ones(len(fr["prev_sP"]), np.int32); e[sl] = ref["inlier_l"]
                        course = np.array_equal(il_all[b, :len(fr["prev_sP"])], e)
                    T = res["T"][b].reshape(4, 4)
                    dT = float(np.max(np.abs(T - ref["T"]))); derr = abs(res["err"][b] - ref["err"]) / max(abs(ref["err"]), 1e-300)
                    if not course or dT > 1e-8 or derr > 1e-8:
                        sT, serr, _ = pose_sensitivity(orc, np.eye(4), CAM, prm, rec, ref, trials=24 if not course else 8)
                        ill += 1
                        if course and not (dT <= max(1e-8, 100 * sT) and derr <= max(1e-8, 100 * serr)):
                            what = f"pose: dT {dT:.3g} derr {derr:.3g} | oracle's own sensitivity {sT:.3g} {serr:.3g}"
                        elif not course and np.isfinite(sT):
                            what = f"course: {(res['status'][b], res['path'][b], tuple(res['iters'][b]))} vs {(ref['status'], ref['path'], ref['iters'])}"
                if what:
                    bad += 1
                    print("MISMATCH", tag, f"| pair {b} ({n1} rows, {len(sel)} matched, {len(sl)} lines):", what, flush=True)
        finally:
            ctx.close()
    print(f"fuzz_track_batched: {case} cases, {pairs} frame pairs ({ill} judged by the oracle's sensitivity), {bad} findings, seed {args.seed}", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
