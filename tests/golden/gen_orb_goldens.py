"""Generates tests/golden/orb_goldens.npz from oracle/stvo_orb_oracle.c: seeded synthetic images and the oracle's ORB
front-end outputs on them (key-points, responses, angles, descriptors).  The oracle restates OpenCV's ORB from its published
algorithm (parity UNPINNED: OpenCV is absent from the reference tree and from this image), so these vectors pin the
restatement against accidental change and give the GPU test a fixed target; they are not OpenCV outputs.
    python tests/golden/gen_orb_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "stvo-pl_amd", "python"))
import oracle_lib  # noqa: E402
from stvo_amd import synth  # noqa: E402


def main():
    orc = oracle_lib.load()
    out = {}
    cases = [(11, 320, 200, 400, 20), (12, 256, 128, 150, 12), (13, 400, 150, 3000, 35)]   # seed, cols, rows, nfeatures, fast_th
    out["cases"] = np.array(cases, np.int32)
    for c, (seed, cols, rows, nf, th) in enumerate(cases):
        img = synth.make_image(seed, cols=cols, rows=rows, n_rects=40, n_discs=12)
        r = orc.orb_detect(img, nfeatures=nf, fast_th=th)
        out[f"img_{c}"] = img
        for k, v in r.items():
            out[f"{k}_{c}"] = v
    out["pattern"] = orc.orb_default_pattern()
    np.savez_compressed(os.path.join(HERE, "orb_goldens.npz"), **out)
    print("wrote orb_goldens.npz:", [len(out[f"kp_{c}"]) for c in range(len(cases))], "key-points")


if __name__ == "__main__":
    main()
