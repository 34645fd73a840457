#!/usr/bin/env python3
"""Randomised parity of the device-resident per-frame pipeline (stvo_seq_*: grid stereo match, tails, f2f mutual match, optimizePose)
against the oracle-driven pipeline (tests/pipeline_ref.py), with the comparison of tests/test_gpu_seq.py::run_and_compare: random
batch sizes (the latency pose kernel, the batch pose kernels, the many-sequences copy-back path), feature counts from empty to full,
presets, optimizer modes, motion model, noise / outlier / distractor levels that reach the failure paths, clustered descriptors.
Test infrastructure.  Run on a GPU box from the repo root:   python tests/fuzz_pipeline.py [--seconds 150] [--seed 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from stvo_amd import synth
import np_model
import pipeline_ref
from stvo_amd.ctypes_types import match_params, opt_params


def oracle_sensitivity(oracle, frames, cam, mp, op, motion_model, idx, trials=8):
    """How far the oracle's result of frame pair `idx` moves when the points handed to its optimizer change by a few roundings
    (relative 1e-15, random signs; the largest move over `trials` patterns).  Two things make a pair that sensitive: a dozen features
    and an iteration that does not settle, and — with robust weights — the MAD scale, whose deviations the reference truncates to
    float (auxiliar.cpp:401,453): a residual near a float rounding boundary moves the scale by a float ulp (6e-8) when it moves at all."""
    base = pipeline_ref.run_sequence(oracle, frames, cam, mp, op, motion_model=motion_model)[idx]
    orig = oracle.optimize_pose
    cs = float(np.max(np.abs(base["cov"])))
    sT = serr = scov = 0.0
    for trial in range(trials):
        prng = np.random.default_rng(trial)

        def perturbed(T0, cam_, prm, rec, *a, **kw):
            rec = dict(rec)
            for key in ("P", "sP", "eP"):   # the 3-D points and the key-lines' 3-D end points
                if key in rec and len(rec[key]):
                    v = np.array(rec[key], float, copy=True)
                    v *= 1.0 + 1e-15 * prng.choice([-1.0, 1.0], v.shape)
                    rec[key] = v
            if motion_model and not np.array_equal(np.asarray(T0), np.eye(4)):
                # with the motion model a pair starts from the previous pair's result, which the comparison accepts to 1e-8: the start
                # of the later pairs of a chain may differ by that much between device and oracle, so the sensitivity includes it
                T0 = np.array(T0, float, copy=True)
                T0[:3, 3] += 1e-9 * prng.standard_normal(3)
            return orig(T0, cam_, prm, rec, *a, **kw)
        oracle.optimize_pose = perturbed
        try:
            other = pipeline_ref.run_sequence(oracle, frames, cam, mp, op, motion_model=motion_model)[idx]
        finally:
            oracle.optimize_pose = orig
        if (other["status"], other["path"], other["iters"], other["n_inliers_pt"], other["n_inliers_ls"]) != \
                (base["status"], base["path"], base["iters"], base["n_inliers_pt"], base["n_inliers_ls"]):
            return np.inf, np.inf, np.inf   # (a few roundings change the optimizer's course: anything goes)
        sT = max(sT, float(np.max(np.abs(base["T"] - other["T"]))))
        serr = max(serr, abs(base["err"] - other["err"]) / max(abs(base["err"]), 1e-300))
        scov = max(scov, float(np.max(np.abs(base["cov"] - other["cov"]))) / cs if cs > 0 else 0.0)
    return sT, serr, scov


def run_and_compare(oracle, seqs, cam, preset, mode=0, has_lines=1, max_kp=2048, max_kl=320, motion_model=False, slots=False):
    """tests/test_gpu_seq.py::run_and_compare with the numeric tolerances tied to the conditioning of the pair's normal equations:
    counts, status, path, iteration counts and inlier counts must be IDENTICAL; pose, error and covariance agree to 1e-8 / 1e-8 /
    1e-6 — or to 1e-13 x cond(cov) where a pair with a handful of features makes the 6 x 6 system that ill-conditioned (the device
    adds its partial sums in another order than the oracle's loop).  Returns the worst deviations seen."""
    from stvo_amd import capi
    B, nf = len(seqs), len(seqs[0])
    mp = match_params(preset)
    op = opt_params(preset, mode=mode, has_lines=has_lines)
    cams = [cam] * B if isinstance(cam, dict) else list(cam)
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=max(B, 1))
    dev = capi.Sequences(ctx, B, max_kp, max_kl, cam, mp, op)
    worst = dict(dT=0.0, dcov=0.0, derr=0.0, cond=0.0)
    try:
        if motion_model:
            dev.set_motion_model(True)
        if slots:   # what bench.py does: every frame resident in its own slot (stvo_seq_set_slots / upload), then steps that ping-pong through them
            S = nf
            dev.set_slots(S)
            for k in range(S):
                dev.upload(k, [seqs[b][k] for b in range(B)])
            visit = (list(range(S)) + list(range(S - 2, 0, -1))) * 2
            visit = visit[:S + 3]
            seqs = [[seqs[b][k] for k in visit] for b in range(B)]   # the frames in the order they are visited: the oracle's sequence
            nf = len(visit)
        refs = [pipeline_ref.run_sequence(oracle, seqs[b], cams[b], mp, op, motion_model=motion_model) for b in range(B)]
        ref0 = [pipeline_ref.stereo_frame(oracle, seqs[b][0], cams[b], mp, True, bool(has_lines)) for b in range(B)]
        for k in range(nf):
            if slots:
                dev.step_dev(visit[k])
                res, counts = dev.read()
            else:
                res, counts = dev.push([seqs[b][k] for b in range(B)])
            for b in range(B):
                if k == 0:
                    assert counts[b, 0] == len(ref0[b]["P"]) and counts[b, 1] == len(ref0[b]["sP"]), (b, k, "first frame counts")
                    continue
                o, r = refs[b][k - 1], res[b]
                assert counts[b, 0] == o["n_stereo_pt"] and counts[b, 1] == o["n_stereo_ls"], (b, k, "stereo counts", counts[b], o["n_stereo_pt"], o["n_stereo_ls"])
                assert r["n_matched_pt"] == o["n_matched_pt"] and r["n_matched_ls"] == o["n_matched_ls"], (b, k, "matched counts")
                if (r["status"], r["path"], tuple(r["iters"]), r["n_inliers_pt"], r["n_inliers_ls"]) != \
                        (o["status"], o["path"], o["iters"], o["n_inliers_pt"], o["n_inliers_ls"]):
                    # another course of the optimizer: a finding, unless a few roundings change the ORACLE's course at this pair as well
                    unstable = not np.isfinite(oracle_sensitivity(oracle, seqs[b], cams[b], mp, op, motion_model, k - 1, trials=24)[0])
                    assert unstable, (b, k, "status / path / iterations / inliers", r["status"], o["status"], r["path"], o["path"], tuple(r["iters"]), o["iters"],
                                      r["n_inliers_pt"], o["n_inliers_pt"], r["n_inliers_ls"], o["n_inliers_ls"])
                    worst["unstable_pairs"] = worst.get("unstable_pairs", 0) + 1
                    break   # (with the motion model the later frames of this sequence start elsewhere: nothing to compare)
                T, cov = r["T"].reshape(4, 4), r["cov"].reshape(6, 6)
                cscale = float(np.max(np.abs(o["cov"])))
                cond = float(np.linalg.cond(o["cov"])) if cscale > 0 and np.all(np.isfinite(o["cov"])) else 1.0
                cond = cond if np.isfinite(cond) else 1e16
                dT = float(np.max(np.abs(T - o["T"])))
                dcov = float(np.max(np.abs(cov - o["cov"]))) / cscale if cscale > 0 else float(np.max(np.abs(cov)))
                derr = abs(r["err"] - o["err"]) / max(abs(o["err"]), 1e-300)
                for key, v in (("dT", dT), ("dcov", dcov), ("derr", derr), ("cond", cond)):
                    worst[key] = max(worst[key], v)
                if dT <= max(1e-8, 1e-13 * cond) and derr <= max(1e-8, 1e-13 * cond) and dcov <= max(1e-6, 1e-13 * cond):
                    continue
                # beyond the plain tolerances: is the PROBLEM that sensitive?  The oracle again with its optimizer inputs changed by one
                # rounding (1e-15 relative, eight random patterns): a pair whose own answer moves by as much under that is ill-posed (a dozen features, an
                # iteration that does not settle), not wrong — the device adds its partial sums in another order than the oracle's loop
                sT, serr, scov = oracle_sensitivity(oracle, seqs[b], cams[b], mp, op, motion_model, k - 1)
                worst["ill_posed_pairs"] = worst.get("ill_posed_pairs", 0) + 1
                assert dT <= max(1e-8, 100 * sT), (b, k, "pose", dT, "oracle's own sensitivity", sT, cond)
                assert derr <= max(1e-8, 100 * serr), (b, k, "err", derr, "oracle's own sensitivity", serr, cond)
                assert dcov <= max(1e-6, 100 * scov), (b, k, "cov", dcov, "oracle's own sensitivity", scov, cond)
    finally:
        dev.close()
        ctx.close()
    return worst


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    args = ap.parse_args(argv)
    orc = oracle_lib.load()
    t_end = time.time() + args.seconds
    case = bad = 0
    worst = dict(dT=0.0, dcov=0.0, derr=0.0, cond=0.0)
    while time.time() < t_end and (args.cases == 0 or case < args.cases):
        case += 1
        rng = np.random.default_rng([args.seed, case])
        B = int(rng.choice([1, 1, 2, 3, 5, 8, 17, 40, 130]))
        preset = str(rng.choice(["kitti", "euroc"]))
        cam = synth.KITTI_CAM if preset == "kitti" else synth.EUROC_CAM
        cams = cam if rng.integers(0, 2) or preset == "euroc" else [synth.config5_cam(int(s)) for s in rng.integers(0, 8, B)]
        mode = int(rng.choice([0, 0, 1, 2])); has_lines = int(rng.integers(0, 2)); mm = bool(rng.integers(0, 2))
        nf = int(rng.integers(3, 6))
        kw = dict(distract=float(rng.choice([0.0, 0.2, 0.6, 1.0])), flip_p=float(rng.choice([0.0, 0.03, 0.08, 0.15])),
                  noise_px=float(rng.choice([0.1, 0.3, 1.0, 3.0])), outlier_frac=float(rng.choice([0.0, 0.05, 0.2, 0.5])))
        if rng.integers(0, 4) == 0:
            kw["cluster_kw"] = dict(cluster_frac=float(rng.choice([0.6, 0.9])), cluster_size=int(rng.choice([8, 16])), spread_p=0.05)
        # (distractors are extra key-points / key-lines: the totals stay within the capacities of 2048 rows and 320 key-lines)
        pts_hi = int(rng.choice([12, 100, 700, 2000])); lines_hi = int(rng.choice([0, 10, 100, 300])) if has_lines else 0
        pts_hi = min(pts_hi, int(2000 / (1.0 + kw["distract"]))); lines_hi = min(lines_hi, int(300 / (1.0 + kw["distract"])))
        seqs = []
        for b in range(B):
            c = cams[b] if isinstance(cams, list) else cams
            seqs.append(synth.make_stereo_sequence(int(rng.integers(1, 1 << 30)), n_frames=nf, n_pts=int(rng.integers(0, pts_hi + 1)),
                                                   n_lines=int(rng.integers(0, lines_hi + 1)), cam=c, **kw))
        use_slots = bool(rng.integers(0, 3) == 0)
        tag = f"seed {args.seed} case {case}: slots {use_slots} B {B} {preset} mode {mode} lines {has_lines} mm {mm} frames {nf} pts<={pts_hi} lines<={lines_hi} {kw}"
        try:
            w = run_and_compare(orc, seqs, cams, preset, mode=mode, has_lines=has_lines, motion_model=mm, slots=use_slots)
            for key, v in w.items():
                worst[key] = worst.get(key, 0) + v if key.endswith("_pairs") else max(worst.get(key, 0.0), v)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", tag, "|", str(e)[:400], flush=True)
        except Exception as e:  # an error status from the library is a finding too
            bad += 1
            print("ERROR", tag, "|", repr(e)[:300], flush=True)
    print(f"fuzz_pipeline: {case} cases, {bad} findings, seed {args.seed}; worst deviations of the passing cases (incl. the pairs the oracle itself is unstable on): {worst}", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
