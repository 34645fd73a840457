#!/usr/bin/env python3
"""Images in, poses out (bench.py's `images_to_poses` leg on its own): the ORB point front-end feeding the device-resident
per-frame pipeline for B stereo streams of KITTI-size images.
    python tools/bench_images.py [--streams 128] [--steps 8]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))

import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=128)
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    print(json.dumps(bench.images_leg(0, B=a.streams, steps=a.steps)))
