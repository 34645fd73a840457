#!/usr/bin/env python3
"""single-stream stvo_seq_push latency (median ms per frame) of the library STVO_LIB points at: KITTI-shaped and EuRoC-shaped, three runs each"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))
import latency_variants as lv
import torch  # noqa: F401
from stvo_amd import synth
kitti = synth.make_stereo_sequence(synth.frame_seed(77, 0), n_frames=61, n_pts=1650, n_lines=85, cam=synth.KITTI_CAM)
euroc = synth.make_stereo_sequence(synth.frame_seed(78, 0), n_frames=61, n_pts=660, n_lines=250, cam=synth.EUROC_CAM,
                                   depth=(0.5, 8.0), octave_probs=[.5, .25, .15, .1], outlier_frac=0.4)
print(os.environ.get("STVO_LIB"), "kitti", [round(lv.run(kitti, synth.KITTI_CAM, "kitti"), 4) for _ in range(3)],
      "euroc", [round(lv.run(euroc, synth.EUROC_CAM, "euroc"), 4) for _ in range(3)], flush=True)
