#!/usr/bin/env python3
"""Randomised parity of the per-stage entry points the handler mirror calls (stvo_match_nnr_mutual, stvo_match_grid_points,
stvo_normal_eq, stvo_optimize_pose) against the oracle: sizes from empty to the capacity, low-entropy descriptors (ties), every window
shape, optimizer modes and presets, outlier / noise levels that reach the failure paths.  Test infrastructure.  Run on a GPU box from
the repo root:   python tests/fuzz_entry_points.py [--seconds 150] [--seed 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from stvo_amd import capi, synth
from stvo_amd.ctypes_types import opt_params


def rand_desc(rng, n, entropy_bits=256):
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if entropy_bits < 256:
        keep = np.zeros(32, np.uint8)
        keep[: entropy_bits // 8] = 0xFF
        d &= keep
    return d


def pose_sensitivity(orc, T0, cam, prm, rec, base, trials=8):
    """The oracle's own move under a few roundings of its 3-D inputs (see tests/fuzz_pipeline.py); inf when its course changes."""
    cs = float(np.max(np.abs(base["cov"])))
    sT = serr = scov = 0.0
    for trial in range(trials):
        prng = np.random.default_rng(trial)
        r2 = dict(rec)
        for key in ("P", "sP", "eP"):
            if len(r2[key]):
                v = np.array(r2[key], float, copy=True)
                v *= 1.0 + 1e-15 * prng.choice([-1.0, 1.0], v.shape)
                r2[key] = v
        o2 = orc.optimize_pose(T0, cam, prm, r2)
        if (o2["status"], o2["path"], o2["iters"]) != (base["status"], base["path"], base["iters"]) or \
                not np.array_equal(o2["inlier_p"], base["inlier_p"]) or not np.array_equal(o2["inlier_l"], base["inlier_l"]):
            return np.inf, np.inf, np.inf
        sT = max(sT, float(np.max(np.abs(base["T"] - o2["T"]))))
        serr = max(serr, abs(base["err"] - o2["err"]) / max(abs(base["err"]), 1e-300))
        scov = max(scov, float(np.max(np.abs(base["cov"] - o2["cov"]))) / cs if cs > 0 else 0.0)
    return sT, serr, scov


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    args = ap.parse_args(argv)
    orc = oracle_lib.load()
    ctx = capi.Context(device_id=0, max_rows=4096, max_batch=4)
    t_end = time.time() + args.seconds
    case = bad = 0
    counts = dict(match=0, grid=0, normal_eq=0, pose=0, pose_ill_posed=0)
    while time.time() < t_end and (args.cases == 0 or case < args.cases):
        case += 1
        rng = np.random.default_rng([args.seed, case])
        tag = f"seed {args.seed} case {case}"
        # ---- StVO::match
        n1, n2 = int(rng.integers(0, 2049)), int(rng.integers(0, 2049))
        if rng.integers(0, 3) == 0:
            n1, n2 = int(rng.integers(0, 70)), int(rng.integers(0, 70))
        ent = int(rng.choice([8, 16, 24, 64, 256])); nnr = float(rng.choice([0.5, 0.75, 0.8, 0.9, 1.0])); mutual = int(rng.integers(0, 2))
        d2 = rand_desc(rng, n2, ent); d1 = rand_desc(rng, n1, ent)
        k = int(rng.integers(0, min(n1, n2) + 1))
        if k:
            d1[:k] = synth.flip_bits(rng, d2[rng.permutation(n2)[:k]], float(rng.choice([0.0, 0.02, 0.06, 0.12])))
        got, n = ctx.match(d1, d2, nnr, mutual)
        exp, en = orc.match(d1, d2, nnr, mutual)
        counts["match"] += 1
        if not np.array_equal(got, exp) or n != en:
            bad += 1
            print(f"MISMATCH match {tag}: {n1} x {n2} entropy {ent} nnr {nnr} mutual {mutual}: rows {np.nonzero(got != exp)[0][:8]}", flush=True)
        # ---- StVO::matchGrid (points)
        g1, g2 = int(rng.integers(1, 2049)), int(rng.integers(1, 2049))
        spread = float(rng.choice([0.1, 0.3, 1.0])); gent = int(rng.choice([8, 16, 32, 256])); ratio = float(rng.choice([0.6, 0.75, 0.9, 1.0]))
        w = tuple(int(x) for x in (rng.choice([0, 3, 10, 70]), rng.choice([0, 2, 10]), rng.choice([0, 1, 50]), rng.choice([0, 1, 50])))
        kp2 = np.stack([rng.uniform(0, 1241 * spread, g2), rng.uniform(0, 376 * spread, g2)], 1)
        kp1 = np.stack([rng.uniform(-30, 1241 * spread + 250, g1), rng.uniform(-10, 376 * spread + 10, g1)], 1)
        iw, ih = 64 / 1241.0, 48 / 376.0
        c1 = np.stack([(kp1[:, 0] * iw).astype(np.int32), (kp1[:, 1] * ih).astype(np.int32)], 1)
        c2 = np.stack([(kp2[:, 0] * iw).astype(np.int32), (kp2[:, 1] * ih).astype(np.int32)], 1)
        e1, e2 = rand_desc(rng, g1, gent), rand_desc(rng, g2, gent)
        start, items = orc.grid_build(c2)
        gm = int(rng.integers(0, 2))
        got, n = ctx.match_grid_points(c1, e1, start, items, e2, w, ratio, gm)
        exp, en = orc.match_grid_points(c1, e1, start, items, e2, w, ratio, gm)
        counts["grid"] += 1
        if not np.array_equal(got, exp) or n != en:
            bad += 1
            print(f"MISMATCH grid {tag}: {g1} x {g2} entropy {gent} window {w} ratio {ratio} spread {spread} mutual {gm}: rows {np.nonzero(got != exp)[0][:8]}", flush=True)
        # ---- StVO::matchGrid (lines): both end-point windows, the direction gate, the NaN direction of same-cell end points
        if case % 4 == 0:   # (the oracle rasterises every right line: every fourth case)
            l1, l2 = int(rng.integers(1, 513)), int(rng.integers(1, 513))
            lent = int(rng.choice([8, 24, 256])); lth = float(rng.choice([0.3, 0.75, 0.95])); lratio = float(rng.choice([0.6, 0.75, 1.0]))
            lw = tuple(int(x) for x in (rng.choice([0, 3, 10]), rng.choice([0, 2, 10]), rng.choice([0, 1, 5]), rng.choice([0, 1, 5])))

            def make_lines(n):
                a = np.stack([rng.uniform(0, 1241.0, n), rng.uniform(0, 376.0, n)], 1)
                b = a + rng.uniform(-150, 150, (n, 2))
                b[:, 0] = np.clip(b[:, 0], 0, 1240.0); b[:, 1] = np.clip(b[:, 1], 0, 375.0)
                return a.astype(np.float32), b.astype(np.float32)
            s1, t1 = make_lines(l1); s2, t2 = make_lines(l2)
            k = int(rng.integers(0, min(l1, 10) + 1))
            t1[:k] = s1[:k] + 1.0
            cl1 = np.concatenate([(s1[:, 0:1] * iw).astype(np.int32), (s1[:, 1:2] * ih).astype(np.int32),
                                  (t1[:, 0:1] * iw).astype(np.int32), (t1[:, 1:2] * ih).astype(np.int32)], 1)
            ent_xy, owner = [], []
            for j in range(l2):
                cells = orc.line_coords(s2[j, 0] * iw, s2[j, 1] * ih, t2[j, 0] * iw, t2[j, 1] * ih)
                ent_xy.append(cells); owner += [j] * len(cells)
            lstart, litems = orc.grid_build(np.concatenate(ent_xy), np.array(owner, np.int32))
            v = np.stack([(t2[:, 0] - s2[:, 0]).astype(np.float64) * iw, (t2[:, 1] - s2[:, 1]).astype(np.float64) * ih], 1)
            with np.errstate(invalid="ignore", divide="ignore"):
                dir2 = v / np.linalg.norm(v, axis=1, keepdims=True)
            f1, f2 = rand_desc(rng, l1, lent), rand_desc(rng, l2, lent)
            lm = int(rng.integers(0, 2))
            got, n = ctx.match_grid_lines(cl1, f1, lstart, litems, f2, dir2, lw, lratio, lth, lm)
            exp, en = orc.match_grid_lines(cl1, f1, lstart, litems, f2, dir2, lw, lratio, lth, lm)
            counts["grid_lines"] = counts.get("grid_lines", 0) + 1
            if not np.array_equal(got, exp) or n != en:
                bad += 1
                print(f"MISMATCH grid lines {tag}: {l1} x {l2} entropy {lent} window {lw} ratio {lratio} gate {lth} mutual {lm}: rows {np.nonzero(got != exp)[0][:8]}", flush=True)
        # ---- optimizeFunctions + optimizePose
        preset = str(rng.choice(["kitti", "euroc"])); mode = int(rng.choice([0, 0, 1, 2]))
        npts = int(rng.choice([0, 3, 12, 60, 300, 1200, 2048])); nl = int(rng.choice([0, 0, 5, 60, 300, 512]))
        npts = int(rng.integers(0, npts + 1)); nl = int(rng.integers(0, nl + 1))
        rec = synth.make_matched_records(int(rng.integers(1, 1 << 30)), n_pts=npts, n_lines=nl, octave_probs=[.5, .25, .15, .1] if preset == "euroc" else None,
                                         outlier_frac=float(rng.choice([0.0, 0.15, 0.4, 0.7])), noise_px=float(rng.choice([0.1, 0.5, 2.0])))
        prm = opt_params(preset, mode=mode)
        cam = synth.KITTI_CAM
        T0 = np.eye(4)
        robust = int(rng.integers(0, 2))
        H, g, e, nn = ctx.normal_eq(T0, cam, prm, rec, robust)
        oH, og, oe, on = orc.optimize_functions(T0, cam, prm, rec, robust)
        counts["normal_eq"] += 1
        sc = max(np.abs(oH).max(), 1e-300)
        # (robust weights: the MAD scale is truncated to float in the reference — a residual on a float rounding boundary moves it by an ulp)
        rt = 1e-6 if robust else 1e-10
        if not (np.allclose(H, oH, rtol=rt, atol=rt * 10 * sc) and np.allclose(g, og, rtol=rt, atol=rt * 10 * max(np.abs(og).max(), 1e-300)) and nn == on
                and ((np.isnan(e) and np.isnan(oe)) or np.isclose(e, oe, rtol=rt))):
            bad += 1
            print(f"MISMATCH normal_eq {tag}: {preset} pts {npts} lines {nl} robust {robust}: dH {np.abs(H - oH).max() / sc:.3g} n {nn} vs {on} e {e} vs {oe}", flush=True)
        out = ctx.optimize_pose(T0, cam, prm, rec)
        ref = orc.optimize_pose(T0, cam, prm, rec)
        counts["pose"] += 1
        same_course = (out["status"], out["path"], out["iters"]) == (ref["status"], ref["path"], ref["iters"]) and \
            np.array_equal(out["inlier_p"], ref["inlier_p"]) and np.array_equal(out["inlier_l"], ref["inlier_l"])
        cs = float(np.max(np.abs(ref["cov"])))
        dT = float(np.max(np.abs(out["T"] - ref["T"])))
        derr = abs(out["err"] - ref["err"]) / max(abs(ref["err"]), 1e-300)
        dcov = float(np.max(np.abs(out["cov"] - ref["cov"]))) / cs if cs > 0 else float(np.max(np.abs(out["cov"])))
        if not same_course or dT > 1e-8 or derr > 1e-8 or dcov > 1e-6:
            sT, serr, scov = pose_sensitivity(orc, T0, cam, prm, rec, ref, trials=24 if not same_course else 8)
            counts["pose_ill_posed"] += 1
            ok = (same_course or not np.isfinite(sT)) and dT <= max(1e-8, 100 * sT) and derr <= max(1e-8, 100 * serr) and dcov <= max(1e-6, 100 * scov) \
                if same_course else not np.isfinite(sT)
            if not ok:
                bad += 1
                print(f"MISMATCH pose {tag}: {preset} mode {mode} pts {npts} lines {nl}: course {(out['status'], out['path'], out['iters'])} vs "
                      f"{(ref['status'], ref['path'], ref['iters'])} dT {dT:.3g} derr {derr:.3g} dcov {dcov:.3g} | oracle's own sensitivity {sT:.3g} {serr:.3g} {scov:.3g}", flush=True)
    ctx.close()
    print(f"fuzz_entry_points: {case} cases ({counts}), {bad} findings, seed {args.seed}", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
