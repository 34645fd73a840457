#!/usr/bin/env python3
"""Randomised parity of images in -> poses out on the device (ORB [+ LSD + LBD] -> device ingest -> per-frame pipeline: stvo_orb_detect_dev,
stvo_lsd_detect_dev, stvo_lbd_compute_dev, stvo_keylines_xy_dev, stvo_seq_upload_dev, stvo_seq_step_dev) against the same chain on the
CPU (ORB / LSD / LBD oracles -> oracle-driven pipeline): random image sizes, pyramid levels, FAST thresholds, stream counts, scene
motions, with and without key-lines.  Test infrastructure.  Run on a GPU box:  python tests/fuzz_images.py [--seconds 120] [--seed 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import pipeline_ref
from fuzz_pipeline import oracle_sensitivity
from stvo_amd import capi, images, synth
from stvo_amd.ctypes_types import match_params, opt_params


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    args = ap.parse_args(argv)
    orc = oracle_lib.load()
    t_end = time.time() + args.seconds
    case = bad = pairs = 0
    while time.time() < t_end and (args.cases == 0 or case < args.cases):
        case += 1
        rng = np.random.default_rng([args.seed, case])
        cols, rows = int(rng.integers(320, 900)), int(rng.integers(160, 420))
        cam = dict(synth.KITTI_CAM, width=cols, height=rows, cx=cols / 2.0 - 3.3, cy=rows / 2.0 + 1.7)
        B = int(rng.integers(1, 4)); nf = int(rng.integers(3, 5)); lines = bool(rng.integers(0, 2))
        nlevels = int(rng.choice([1, 1, 2, 4])); nfeat = int(rng.choice([300, 1000, 2000])); th = int(rng.choice([10, 20, 30]))
        nlines = int(rng.choice([30, 100])); min_len = 0.025 * min(cols, rows)
        mp = match_params("kitti"); op = opt_params("kitti", has_lines=1 if lines else 0)
        seqs = [synth.make_stereo_image_sequence(int(rng.integers(1, 1 << 30)), nf, cam, shift_per_disp=float(rng.uniform(0.15, 0.4))) for _ in range(B)]
        tag = f"seed {args.seed} case {case}: {cols}x{rows} B {B} frames {nf} levels {nlevels} nfeatures {nfeat} fast {th} lines {lines} ({nlines})"
        ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
        pipe = images.ImagePipeline(ctx, B, cam, mp, op, max_kp=2048, nfeatures=nfeat, fast_threshold=th, nlevels=nlevels,
                                    lsd=capi.lsd_params(min_length=min_len, nfeatures=nlines) if lines else None, max_kl=128)
        try:
            pattern = pipe.orb.pattern()
            lopts = orc.lsd_opts(min_length=min_len, nfeatures=nlines)
            frames = []
            for b in range(B):
                fr_b = []
                for left, right in seqs[b]:
                    fr = {}
                    for side, img in (("l", left), ("r", right)):
                        o = orc.orb_detect_levels(img, nfeatures=nfeat, nlevels=nlevels, fast_th=th, pattern=pattern, cap=2048)   # (the capacity of the device pipeline: ties at the retainBest cut can exceed nfeatures)
                        fr["kp_" + side] = o["kp"]; fr["desc_" + side] = o["desc"]
                        if side == "l":
                            fr["oct_l"] = o["octave"]
                        if lines:
                            kl = orc.lsd_detect(img, lopts)
                            rec = np.stack([kl["sx"], kl["sy"], kl["ex"], kl["ey"], kl["angle"]], axis=1).astype(np.float32)
                            fr["kl_" + side] = np.ascontiguousarray(rec[:, :4]); fr["ldesc_" + side] = orc.lbd_compute(img, rec, kl["num_pixels"])
                            if side == "l":
                                fr["ang_l"] = np.ascontiguousarray(rec[:, 4]); fr["oct_ll"] = np.zeros(len(rec), np.int32)
                        else:
                            fr["kl_" + side] = np.zeros((0, 4), np.float32); fr["ldesc_" + side] = np.zeros((0, 32), np.uint8)
                            if side == "l":
                                fr["oct_ll"] = np.zeros(0, np.int32)
                    fr_b.append(fr)
                frames.append(fr_b)
            # (the handler's adaptive FAST threshold is the handler's: here the threshold stays fixed, on both sides)
            refs = [pipeline_ref.run_sequence(orc, frames[b], cam, mp, op, fast=dict(adaptive=False, th0=th, mn=th, mx=th, inc=0, feat=0, err=0.0)) for b in range(B)]
            for k in range(nf):
                res, counts = pipe.push_images(np.stack([seqs[b][k][0] for b in range(B)]), np.stack([seqs[b][k][1] for b in range(B)]))
                if k == 0:
                    continue
                for b in range(B):
                    pairs += 1
                    o, r = refs[b][k - 1], res[b]
                    what = None
                    if (counts[b, 0], counts[b, 1], r["n_matched_pt"], r["n_matched_ls"]) != (o["n_stereo_pt"], o["n_stereo_ls"], o["n_matched_pt"], o["n_matched_ls"]):
                        what = f"counts {tuple(counts[b])} / matched {(r['n_matched_pt'], r['n_matched_ls'])} vs {(o['n_stereo_pt'], o['n_stereo_ls'], o['n_matched_pt'], o['n_matched_ls'])}"
                    else:
                        course = (r["status"], r["path"], tuple(r["iters"]), r["n_inliers_pt"], r["n_inliers_ls"]) == \
                            (o["status"], o["path"], o["iters"], o["n_inliers_pt"], o["n_inliers_ls"])
                        dT = float(np.max(np.abs(r["T"].reshape(4, 4) - o["T"])))
                        if not course or dT > 1e-8:
                            sT = oracle_sensitivity(orc, frames[b], cam, mp, op, False, k - 1, trials=24 if not course else 8)[0]
                            if (not course and np.isfinite(sT)) or (course and dT > max(1e-8, 100 * sT)):
                                what = f"course {course} dT {dT:.3g} | oracle's own sensitivity {sT:.3g}"
                    if what:
                        bad += 1
                        print("MISMATCH", tag, f"| stream {b} frame {k}:", what, flush=True)
        finally:
            pipe.close(); ctx.close()
    print(f"fuzz_images: {case} cases, {pairs} frame pairs, {bad} findings, seed {args.seed}", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
