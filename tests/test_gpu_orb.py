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
    with pytest.raises(capi.StvoError):
        capi.Orb(hip, 1, 640, 200, max_keypoints=5000)    # beyond what the ordering kernel sorts: refused, not truncated silently
    with pytest.raises(capi.StvoError):
        capi.Orb(hip, 1, 640, 200, nlevels=9)
    with pytest.raises(capi.StvoError):
        capi.Orb(hip, 1, 640, 200, nlevels=4, scale_factor=1.0)


def same_levels(got, ref):
    same(got, ref)
    assert np.array_equal(got["octave"], ref["octave"])


@pytest.mark.parametrize("cols,rows,nf,nlev,sf,th", [(752, 480, 600, 4, 1.2, 20), (1241, 376, 2000, 4, 1.2, 20), (640, 200, 300, 8, 1.2, 7),
                                                     (400, 300, 900, 3, 1.5, 12), (752, 480, 600, 1, 1.2, 20)])
def test_orb_pyramid_levels_bit_exact(hip, oracle, cols, rows, nf, nlev, sf, th):
    """orb_nlevels > 1 (config_euroc.yaml / src/config.cpp: 4 levels at 1.2): pyramid by 8-bit bilinear resize, the budget split over
    the levels, key-points of all levels in level order with octave and coordinates scaled back — bit for bit against the oracle."""
    from stvo_amd import capi
    B = 2
    imgs = np.stack([synth.make_image(300 + b + nf, cols=cols, rows=rows, n_rects=400, n_discs=100) for b in range(B)])
    orb = capi.Orb(hip, B, cols, rows, max_keypoints=2048, nfeatures=nf, fast_threshold=th, nlevels=nlev, scale_factor=sf)
    try:
        out = orb.detect(imgs)
        for b in range(B):
            ref = oracle.orb_detect_levels(imgs[b], nfeatures=nf, nlevels=nlev, scale_factor=sf, fast_th=th, cap=2048)
            same_levels(out[b], ref)
            if nlev > 1:
                assert len(np.unique(ref["octave"])) >= min(nlev, 3)   # the upper levels really contribute
                assert np.all(np.diff(ref["octave"]) >= 0)             # level order
    finally:
        orb.close()


def test_orb_masses_of_equal_responses_are_truncated_deterministically(hip, oracle):
    """A regular dot pattern gives thousands of corners with ONE response value: retainBest keeps all the ties, far more than the
    ordering kernel sorts in LDS.  The output must then be exactly the first max_keypoints of the row-major order (what the oracle
    emits), independent of the order in which the tiles appended their candidates, and n_total must report the uncapped count."""
    from stvo_amd import capi
    cols, rows = 1024, 512
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = np.where((yy % 7 == 0) & (xx % 7 == 0), 200, 40).astype(np.uint8)   # bright dots on a 7-pixel lattice: ~9.6 k equal corners
    ref = oracle.orb_detect(img, nfeatures=200, fast_th=20, cap=1 << 16)
    assert len(ref["kp"]) > 4096 and len(np.unique(ref["response"])) <= 4
    for cap in (4096, 1000):
        orb = capi.Orb(hip, 2, cols, rows, max_keypoints=cap, nfeatures=200, fast_threshold=20)
        try:
            out = orb.detect(np.stack([img, img[::-1].copy()]))
            a = out[0]
            for k in ("kp", "response", "angle", "desc"):
                assert np.array_equal(a[k].view(np.uint8), ref[k][:cap].view(np.uint8)), (cap, k)
            assert a["n_total"] == len(ref["kp"]) and len(a["kp"]) == cap
            ref2 = oracle.orb_detect(img[::-1].copy(), nfeatures=200, fast_th=20, cap=cap)
            same(out[1], ref2)
            again = orb.detect(np.stack([img, img[::-1].copy()]))   # run to run
            assert np.array_equal(again[0]["kp"], a["kp"]) and np.array_equal(again[0]["desc"], a["desc"])
        finally:
            orb.close()
