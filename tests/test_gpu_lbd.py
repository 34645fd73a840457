"""GPU parity of the LBD line descriptor (stvo_lbd_*, csrc/lbd_kernels.hip) against oracle/stvo_lbd_oracle.c through the C-ABI:
the 32-byte binary descriptors and the 72-float descriptors BIT-EXACT on synthetic images with random segments (incl. segments
that leave the image, one-pixel segments, and degenerate flat regions that produce NaNs).  The oracle cites the reference-held
source (3rdparty/line_descriptor/src/binary_descriptor_custom.cpp) line by line; parity with the reference binary stays
UNPINNED because that source needs OpenCV to build."""
import numpy as np
import pytest

from stvo_amd import synth

pytestmark = pytest.mark.gpu


def random_lines(rng, n, cols, rows, len_px=(15.0, 300.0), margin=-10.0):
    sx = rng.uniform(-margin, cols + margin, n); sy = rng.uniform(-margin, rows + margin, n)
    L = rng.uniform(*len_px, n); a = rng.uniform(-np.pi, np.pi, n)
    ex = sx + L * np.cos(a); ey = sy + L * np.sin(a)
    lines = np.stack([sx, sy, ex, ey, np.arctan2((ey - sy).astype(np.float32), (ex - sx).astype(np.float32))], 1).astype(np.float32)
    # cv::LineIterator count of the (clipped) 8-connected raster: what KeyLine::numOfPixels holds
    npx = (np.maximum(np.abs(np.round(ex) - np.round(sx)), np.abs(np.round(ey) - np.round(sy))) + 1).astype(np.int32)
    return lines, npx


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("cols,rows,n", [(1241, 376, 100), (752, 480, 300), (320, 200, 512),
                                         (258, 67, 40), (259, 66, 40), (35, 21, 12), (24, 23, 16)])  # (widths of every residue mod 4, images a few words wide: the borders of the four-pixel blur / Sobel)
def test_lbd_bit_exact(hip, oracle, cols, rows, n):
    from stvo_amd import capi
    B = 2
    rng = np.random.default_rng(cols + n)
    imgs = np.stack([synth.make_image(40 + b + n, cols=cols, rows=rows, n_rects=300, n_discs=80) for b in range(B)])
    sets = [random_lines(rng, n - 7 * b, cols, rows) for b in range(B)]
    lbd = capi.Lbd(hip, B, cols, rows, max_keylines=512)
    try:
        got, got_f = lbd.compute(imgs, [s[0] for s in sets], [s[1] for s in sets], want_float=True)
        for b in range(B):
            ref, ref_f = oracle.lbd_compute(imgs[b], sets[b][0], sets[b][1], want_float=True)
            assert same_bits(got_f[b], ref_f), b           # the 72 floats, NaN payloads included
            assert np.array_equal(got[b], ref), b
            assert 0.2 < np.unpackbits(ref, axis=1).mean() < 0.8   # real descriptors, not all-zero rows
        # descriptors of a line and of the same line seen again are identical; of the reversed line they differ
        again = lbd.compute(imgs, [s[0] for s in sets], [s[1] for s in sets])
        assert np.array_equal(again[0], got[0])
    finally:
        lbd.close()


def test_lbd_edge_cases(hip, oracle):
    """Segments far outside the image (every sample clamps to the border), one- and two-pixel segments, a constant image (all
    gradients zero: 0 / 0 in the normalisations -> NaN descriptors, compared bit for bit), zero key-lines in one image."""
    from stvo_amd import capi
    cols, rows = 400, 300
    img = synth.make_image(5, cols=cols, rows=rows, n_rects=80, n_discs=20)
    flat = np.full((rows, cols), 77, np.uint8)
    lines = np.array([[-500, -500, -300, -450, 0.25], [10, 10, 10, 10, 0.0], [50, 50, 51, 50, 0.0], [399, 299, 700, 600, 0.785],
                      [200, 150, 200, 20, -1.5707964], [5, 290, 395, 292, 0.005]], np.float32)
    npx = np.array([201, 1, 2, 302, 131, 391], np.int32)
    lbd = capi.Lbd(hip, 3, cols, rows, max_keylines=16)
    try:
        got, got_f = lbd.compute(np.stack([img, flat, img]), [lines, lines, lines[:0]], [npx, npx, npx[:0]], want_float=True)
        for b, im in ((0, img), (1, flat)):
            ref, ref_f = oracle.lbd_compute(im, lines, npx, want_float=True)
            assert same_bits(got_f[b], ref_f) and np.array_equal(got[b], ref), b
        assert np.isnan(got_f[1]).any()        # the flat image really exercises the NaN path
        assert got[2].shape == (0, 32)
    finally:
        lbd.close()
