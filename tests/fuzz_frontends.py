#!/usr/bin/env python3
"""Randomised parity of the image front-ends (ORB, LSD incl. lsd_refine 1, LBD on LSD's key-lines) against the oracle, through the
C-ABI: random image sizes (every residue of the width, small and large), random content (scenes, blurred scenes, noise, ramps,
checkerboards, saturated blocks, flat), random parameters.  Test infrastructure (round 6: a blurred scene found a one-off in
numOfPixels that the fixed test images never met).  Run on a GPU box from the repo root:
    python tests/fuzz_frontends.py [--seconds 120] [--seed 1]
Prints one line per mismatch with everything needed to replay it; exit code 1 if any."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from scipy.ndimage import gaussian_filter
from stvo_amd import capi, synth


def make_image(rng, cols, rows):
    kind = rng.integers(0, 8)
    seed = int(rng.integers(1, 1 << 30))
    if kind <= 1:
        img = synth.make_image(seed, cols, rows, n_rects=int(rng.integers(5, 200)), n_discs=int(rng.integers(0, 60)), noise=float(rng.uniform(0, 6)))
    elif kind == 2:
        img = synth.make_image(seed, cols, rows)
        img = np.clip(np.rint(gaussian_filter(img.astype(float), float(rng.uniform(0.6, 2.5)))), 0, 255)
    elif kind == 3:
        img = rng.integers(0, 256, (rows, cols))
    elif kind == 4:
        yy, xx = np.mgrid[0:rows, 0:cols]
        img = (xx * rng.uniform(-1, 1) + yy * rng.uniform(-1, 1)) % 256 + rng.normal(0, rng.uniform(0, 3), (rows, cols))
    elif kind == 5:
        p = int(rng.integers(2, 40))
        yy, xx = np.mgrid[0:rows, 0:cols]
        img = (((xx // p) + (yy // p)) % 2) * rng.uniform(60, 255) + rng.normal(0, rng.uniform(0, 2), (rows, cols))
    elif kind == 6:
        img = np.kron(rng.integers(0, 2, ((rows + 15) // 16, (cols + 15) // 16)) * 255, np.ones((16, 16)))[:rows, :cols]
        if rng.integers(0, 2):
            img = gaussian_filter(img.astype(float), 1.0)
    else:
        img = np.full((rows, cols), int(rng.integers(0, 256)))
    return np.clip(np.rint(np.asarray(img, float)), 0, 255).astype(np.uint8)


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    args = ap.parse_args(argv)
    orc = oracle_lib.load()
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=8)
    t_end = time.time() + args.seconds
    case, bad = 0, 0
    counts = {"orb": 0, "lsd": 0, "lbd": 0}
    while time.time() < t_end and (args.cases == 0 or case < args.cases):
        case += 1
        rng = np.random.default_rng([args.seed, case])
        cols, rows = int(rng.integers(64, 900)), int(rng.integers(64, 500))
        if rng.integers(0, 6) == 0:   # the sizes of the shipped configurations (and one beyond LSD's 2^20 pixels: ORB only)
            cols, rows = [(1241, 376), (1226, 370), (752, 480), (1280, 720)][int(rng.integers(0, 4))]
        B = int(rng.integers(1, 6))
        if rng.integers(0, 8) == 0:   # the detector's other batch forms: several images per XCD (9 .. 128), one wave per image (beyond)
            B = int(rng.choice([12, 40, 150]))
            cols, rows = min(cols, 400), min(rows, 260)
        imgs = np.stack([make_image(rng, cols, rows) for _ in range(B)])
        tag = f"seed {args.seed} case {case} {cols}x{rows} B {B}"
        # ---- ORB
        nlev = int(rng.integers(1, 5)); th = int(rng.integers(5, 50)); nf = int(rng.choice([50, 300, 1000, 2000])); cap = int(rng.choice([256, 1024, 4096]))
        sf = float(rng.choice([1.2, 1.5, 2.0]))
        try:
            orb = capi.Orb(ctx, B, cols, rows, max_keypoints=cap, nfeatures=nf, fast_threshold=th, nlevels=nlev, scale_factor=sf)
        except capi.StvoError:
            orb = None   # (levels too small for the patch: refused, fine)
        if orb is not None:
            try:
                out = orb.detect(imgs)
                for b in range(B):
                    ref = orc.orb_detect_levels(imgs[b], nfeatures=nf, nlevels=nlev, scale_factor=sf, fast_th=th, cap=cap) if nlev > 1 else \
                        orc.orb_detect(imgs[b], nfeatures=nf, fast_th=th, cap=cap)
                    ok = all(same_bits(out[b][k], ref[k]) for k in ("kp", "response", "angle", "desc"))
                    counts["orb"] += 1
                    if not ok:
                        bad += 1
                        print(f"MISMATCH orb {tag} b {b} levels {nlev} sf {sf} th {th} nf {nf} cap {cap}: n {len(out[b]['kp'])} vs {len(ref['kp'])}", flush=True)
            finally:
                orb.close()
        # ---- LSD (+ LBD on its key-lines)
        scale = float(rng.choice([0.8, 1.0, 1.2])); refine = int(rng.integers(0, 2)); nb = int(rng.choice([64, 1024, 2048]))
        dth = float(rng.choice([0.6, 0.7, 0.85])); minlen = float(rng.uniform(0, 12)); nfl = int(rng.choice([0, 20, 100, 300])); K = int(rng.choice([64, 512, 2048]))
        if (cols * scale) * (rows * scale) <= (1 << 20):
            prm = capi.lsd_params(min_length=minlen, nfeatures=nfl, scale=scale, refine=refine, n_bins=nb)
            prm.density_th = dth
            opts = orc.lsd_opts(min_length=minlen, nfeatures=nfl, scale=scale, refine=refine, n_bins=nb)
            opts.density_th = dth
            lsd = capi.Lsd(ctx, B, cols, rows, prm, max_keylines=K)
            try:
                segs, n = lsd.segments(imgs)
                dets = lsd.detect(imgs)
                for b in range(B):
                    ref = orc.lsd_segments(imgs[b], opts)
                    counts["lsd"] += 1
                    if n[b] != len(ref) or not same_bits(segs[b], ref[:8192]):
                        bad += 1
                        print(f"MISMATCH lsd segments {tag} b {b} scale {scale} refine {refine} bins {nb} dth {dth}: n {n[b]} vs {len(ref)}", flush=True)
                        continue
                    kl = orc.lsd_detect(imgs[b], opts)
                    rec, resp = dets[b]
                    if len(kl) <= K and len(ref) <= 8192:   # (beyond the capacity the device keeps the K strongest: covered by tests/test_gpu_lsd.py)
                        ok = len(rec) == len(kl) and all(np.array_equal(rec[f], kl[f]) for f in ("sx", "sy", "ex", "ey", "num_pixels")) and \
                            np.array_equal(resp, kl["response"])
                        if not ok:
                            bad += 1
                            print(f"MISMATCH lsd keylines {tag} b {b} scale {scale} refine {refine} minlen {minlen} nf {nfl} K {K}: n {len(rec)} vs {len(kl)}", flush=True)
                # ---- LBD on image 0's key-lines
                rec, _ = dets[0]
                if 0 < len(rec) <= 512:
                    lines = np.stack([rec["sx"], rec["sy"], rec["ex"], rec["ey"], rec["angle"]], 1).astype(np.float32)
                    npx = rec["num_pixels"].astype(np.int32)
                    lbd = capi.Lbd(ctx, 1, cols, rows, max_keylines=512)
                    try:
                        got, got_f = lbd.compute(imgs[:1], [lines], [npx], want_float=True)
                        ref_d, ref_f = orc.lbd_compute(imgs[0], lines, npx, want_float=True)
                        counts["lbd"] += 1
                        if not (same_bits(got_f[0], ref_f) and np.array_equal(got[0], ref_d)):
                            bad += 1
                            print(f"MISMATCH lbd {tag} lines {len(lines)}", flush=True)
                    finally:
                        lbd.close()
            finally:
                lsd.close()
    ctx.close()
    print(f"fuzz_frontends: {case} cases ({counts}), {bad} mismatches, seed {args.seed}", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
