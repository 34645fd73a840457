"""GPU parity of the ORB point front-end (stvo_orb_*, csrc/orb_kernels.hip) against oracle/stvo_orb_oracle.c through the
C-ABI: key-point sets, FAST responses, orientation angles and 32-byte rBRIEF descriptors BIT-EXACT, on synthetic
1241 x 376 images (KITTI size, orb_nfeatures 2000, FAST threshold 20) and on the committed golden vectors.  The oracle
restates OpenCV's ORB (third-party, absent) from its published algorithm: parity with OpenCV itself is UNPINNED."""
import os

import numpy as np
import pytest

from stvo_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "orb_goldens.npz")


def same(got, ref):
    for k in ("kp", "response", "angle", "desc"):
        assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        assert np.array_equal(got[k].view(np.uint8), ref[k].view(np.uint8)), k   # bitwise, floats included


def test_orb_kitti_size_batch_bit_exact(hip, oracle):
    from stvo_amd import capi
    B = 3
    imgs = np.stack([synth.make_image(100 + b) for b in range(B)])
    orb = capi.Orb(hip, B, 1241, 376, max_keypoints=2048, nfeatures=2000, fast_threshold=20)
    try:
        assert np.array_equal(orb.pattern(), oracle.orb_default_pattern())
        out = orb.detect(imgs)
        for b in range(B):
            ref = oracle.orb_detect(imgs[b], nfeatures=2000, fast_th=20, cap=2048)
            assert 1500 < len(ref["kp"]) <= 2048
            same(out[b], ref)
        # a second call on other images reuses the scratch (histograms are reset by the kernels)
        imgs2 = imgs[::-1].copy()
        out2 = orb.detect(imgs2)
        for b in range(B):
            same(out2[b], out[B - 1 - b])
    finally:
        orb.close()


@pytest.mark.parametrize("nf,th,cap", [(300, 12, 512), (5000, 35, 1024), (50, 60, 64), (700, 5, 700)])
def test_orb_thresholds_cuts_and_capacity(hip, oracle, nf, th, cap):
    """retainBest cut with ties, fewer corners than requested, and truncation at the capacity (row-major order)."""
    from stvo_amd import capi
    img = synth.make_image(7 + nf, cols=640, rows=200, n_rects=120, n_discs=30, noise=4.0)
    orb = capi.Orb(hip, 1, 640, 200, max_keypoints=cap, nfeatures=nf, fast_threshold=th)
    try:
        same(orb.detect(img[None])[0], oracle.orb_detect(img, nfeatures=nf, fast_th=th, cap=cap))
    finally:
        orb.close()


def test_orb_custom_pattern_and_goldens(hip, oracle):
    from stvo_amd import capi
    g = np.load(GOLD)
    for c, (seed, cols, rows, nf, th) in enumerate(g["cases"]):
        orb = capi.Orb(hip, 1, int(cols), int(rows), max_keypoints=4096, nfeatures=int(nf), fast_threshold=int(th))
        try:
            got = orb.detect(g[f"img_{c}"][None])[0]
            for k in ("kp", "response", "angle", "desc"):
                assert np.array_equal(got[k], g[f"{k}_{c}"]), (c, k)
            if c == 0:   # another test pattern (what a maintainer does with OpenCV's learned table)
                rng = np.random.default_rng(3)
                pat = rng.integers(-13, 14, (256, 4)).astype(np.int8)
                orb.set_pattern(pat)
                same(orb.detect(g["img_0"][None])[0], oracle.orb_detect(g["img_0"], nfeatures=int(nf), fast_th=int(th), pattern=pat))
                with pytest.raises(capi.StvoError):
                    orb.set_pattern(np.full((256, 4), 14, np.int8))   # would leave the border
        finally:
            orb.close()


def test_orb_rejects_bad_parameters(hip):
    from stvo_amd import capi
    with pytest.raises(capi.StvoError):
        capi.Orb(hip, 1, 640, 200, edge_threshold=10)     # patch would leave the image
    with pytest.raises(capi.StvoError):
        capi.Orb(hip, 1, 32, 32)
