#!/usr/bin/env python3
"""Regression pin of oracle/stvo_lsd_oracle.c (NOT a reference vector: the detector core is OpenCV's, which neither this image nor
/root/reference holds — parity with it stays unpinned).  Writes tests/golden/lsd_oracle_320x200.npz: a synthetic image,
the segments of the detector core and the key-lines after the wrapper + top-50 cut, as the oracle of the commit that wrote it produced
them.  tests/test_oracle_lsd.py::test_oracle_output_is_pinned compares; re-run this script only when the restatement is changed on
purpose.      python tests/golden/make_lsd_golden.py"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "stvo-pl_amd", "python"))
import oracle_lib
from stvo_amd import synth
o = oracle_lib.load()
img = synth.make_image(4711, 320, 200, n_rects=70, n_discs=15)
seg = o.lsd_segments(img, o.lsd_opts())
kl = o.lsd_detect(img, o.lsd_opts(min_length=0.025 * 200, nfeatures=50))
seg1 = o.lsd_segments(img, o.lsd_opts(refine=1))   # lsd_refine = 1 (round 6)
np.savez_compressed(os.path.join(HERE, "lsd_oracle_320x200.npz"), img=img, segments=seg, keylines=kl, segments_refine1=seg1)
print(len(seg), "segments,", len(kl), "key-lines,", len(seg1), "segments with lsd_refine = 1")
