#!/usr/bin/env python3
"""Writes a synthetic KITTI-shaped stereo feature sequence for stvo-pl_amd/bin/imagesStVO_synth.
    python tools/make_sequence.py out.bin [--frames 51] [--points 1650] [--lines 0] [--cam kitti|euroc]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stvo-pl_amd", "python"))
from stvo_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--frames", type=int, default=51)
ap.add_argument("--points", type=int, default=1650)  # + 20 % distractors ~ 2000 key-points per image
ap.add_argument("--lines", type=int, default=0)
ap.add_argument("--cam", default="kitti")
ap.add_argument("--seed", type=int, default=synth.SEED0)
ap.add_argument("--config4", action="store_true",
                help="BASELINE configs[3] statistics: depth 0.5-8 m, octaves {.5,.25,.15,.1}, 40 %% point outliers (use with --cam euroc)")
a = ap.parse_args()
cam = synth.KITTI_CAM if a.cam == "kitti" else synth.EUROC_CAM
extra = dict(depth=(0.5, 8.0), octave_probs=[.5, .25, .15, .1], outlier_frac=0.4) if a.config4 else {}
frames = synth.make_stereo_sequence(a.seed, n_frames=a.frames, n_pts=a.points, n_lines=a.lines, cam=cam, **extra)
synth.write_sequence(a.out, frames, cam)
print(f"wrote {a.out}: {a.frames} frames, {len(frames[0]['kp_l'])} key-points, {len(frames[0]['kl_l'])} key-lines per image")
