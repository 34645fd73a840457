#!/usr/bin/env python3
"""Randomised parity of the C++ host mirror (StereoFrameHandler on the device pipeline and on one synchronous call per stage; the
imagesStVO loop of stvo-pl_amd/app) against the oracle-driven pipeline: random sequences (feature counts, noise, outliers, distractors,
dropped right cameras), presets, optimizer modes, motion model, key-frame decisions.  Feature counts are kept at >= 150 key-points so
that the fixed tolerances of tests/test_gpu_handler.py::compare hold (tests/fuzz_pipeline.py covers the ill-posed pairs).  Test
infrastructure.  Run on a GPU box from the repo root:   python tests/fuzz_handler.py [--seconds 120] [--seed 1]"""
import argparse, os, pathlib, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import pipeline_ref
import test_gpu_handler as th
from stvo_amd import synth
from stvo_amd.ctypes_types import match_params, opt_params


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    args = ap.parse_args(argv)
    orc = oracle_lib.load()
    tmp = pathlib.Path(tempfile.mkdtemp())
    t_end = time.time() + args.seconds
    case = bad = 0
    while time.time() < t_end and (args.cases == 0 or case < args.cases):
        case += 1
        rng = np.random.default_rng([args.seed, case])
        preset = str(rng.choice(["kitti", "euroc"]))
        cam = synth.KITTI_CAM if preset == "kitti" else synth.EUROC_CAM
        mode = int(rng.choice([0, 0, 1, 2])); mm = bool(rng.integers(0, 2)); kf = bool(rng.integers(0, 3) == 0); pipe = bool(rng.integers(0, 2))
        nf = int(rng.integers(3, 9))
        distract = float(rng.choice([0.0, 0.2, 0.6]))
        kw = dict(distract=distract, flip_p=float(rng.choice([0.0, 0.03, 0.08])), noise_px=float(rng.choice([0.1, 0.3, 1.0])),
                  outlier_frac=float(rng.choice([0.0, 0.05, 0.2])))
        n_pts = int(rng.integers(150, int(2000 / (1 + distract)) + 1)); n_lines = int(rng.integers(0, int(300 / (1 + distract)) + 1))
        frames = synth.make_stereo_sequence(int(rng.integers(1, 1 << 30)), n_frames=nf, n_pts=n_pts, n_lines=n_lines, cam=cam,
                                            octave_probs=[.5, .25, .15, .1] if preset == "euroc" else None,
                                            depth=(1.0, 8.0) if preset == "euroc" else (4.0, 60.0), **kw)
        if rng.integers(0, 4) == 0 and nf >= 5:   # the right camera drops out for a frame: two pairs fail in band
            k = int(rng.integers(2, nf - 1))
            z2 = np.zeros((0, 2), np.float32); zd = np.zeros((0, 32), np.uint8); z4 = np.zeros((0, 4), np.float32)
            frames[k] = dict(frames[k], kp_r=z2, desc_r=zd, kl_r=z4, ldesc_r=zd)
        cfg = tmp / "cfg.yaml"
        cfg.write_text(("use_motion_model : true\n" if mm else "") + ("max_kf_t_dist : 1.5\n" if kf else ""))
        extra = ("-c", str(cfg)) + (("--keyframes",) if kf else ())
        tag = f"seed {args.seed} case {case}: {preset} mode {mode} mm {mm} keyframes {kf} pipeline {pipe} frames {nf} pts {n_pts} lines {n_lines} {kw}"
        try:
            res, _ = th.run_app(tmp, frames, cam, preset, mode=mode, extra=extra, pipeline=pipe)
            fast = dict(adaptive=True, th0=20, mn=5, mx=50, inc=5, feat=50, err=0.5) if preset == "euroc" else \
                dict(adaptive=True, th0=20, mn=7, mx=30, inc=5, feat=50, err=0.5)   # config_euroc.yaml / config_kitti.yaml: the adaptive FAST threshold's range
            ref = pipeline_ref.run_sequence(orc, frames, cam, match_params(preset), opt_params(preset, mode=mode), fast, motion_model=mm,
                                            keyframes=dict(min_entropy_ratio=0.85, max_kf_t_dist=1.5, max_kf_r_dist=15.0) if kf else None)
            th.compare(res, ref)
            if kf:
                assert [int(r["pad"]) for r in res] == [o["new_kf"] for o in ref], "key-frame decisions"
        except AssertionError as e:
            bad += 1
            import traceback
            where = [l.strip() for l in traceback.format_exc().splitlines() if l.strip().startswith("assert")]
            print("MISMATCH", tag, "|", (where[-1] if where else "")[:160], "|", str(e)[:300].replace("\n", " "), flush=True)
    print(f"fuzz_handler: {case} cases, {bad} findings, seed {args.seed}", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
