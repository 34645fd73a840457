"""CPU checks of the LSD oracle (oracle/stvo_lsd_oracle.c) — the restatement of cv::LineSegmentDetector + LSDDetectorC::detectImpl + the
top-N cut of StereoFrame::detectLineFeatures that the HIP detector is compared with on the GPU (tests/test_gpu_lsd.py).  Parity with
OpenCV itself is unpinned (no OpenCV here, no vector in the reference); what can be checked on the CPU is that the restatement behaves
like a line-segment detector with the wrapper's contract."""
import numpy as np

from stvo_amd import synth


def test_sincos_det_is_within_an_ulp(oracle):
    xs = np.concatenate([np.linspace(-7.0, 10.0, 3001), [0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 1e-9, -1e-9]])
    for x in xs:
        s, c = oracle.sincos_det(x)
        assert abs(s - np.sin(x)) <= 2.3e-16 and abs(c - np.cos(x)) <= 2.3e-16


def test_rectangle_edges_are_found(oracle):
    """A bright rectangle on a flat background: exactly its four edges, each within a pixel of where it was drawn, end points ordered
    so that the brighter side lies to the same hand (the level-line direction), lengths close to the sides."""
    img = np.full((200, 320), 100, np.uint8)
    img[50:150, 80:240] = 200
    seg = oracle.lsd_segments(img, oracle.lsd_opts())
    assert len(seg) == 4
    horiz = sorted([s for s in seg if abs(s[1] - s[3]) < 1.0], key=lambda s: s[1])
    vert = sorted([s for s in seg if abs(s[0] - s[2]) < 1.0], key=lambda s: s[0])
    assert len(horiz) == 2 and len(vert) == 2
    assert abs(horiz[0][1] - 49.5) < 1.0 and abs(horiz[1][1] - 149.5) < 1.0
    assert abs(vert[0][0] - 79.5) < 1.0 and abs(vert[1][0] - 239.5) < 1.0
    for s in horiz:
        assert abs(abs(s[0] - s[2]) - 160) < 4
    for s in vert:
        assert abs(abs(s[1] - s[3]) - 100) < 4
    # opposite edges run in opposite directions
    assert (horiz[0][2] - horiz[0][0]) * (horiz[1][2] - horiz[1][0]) < 0 and (vert[0][3] - vert[0][1]) * (vert[1][3] - vert[1][1]) < 0


def test_flat_and_scale_one(oracle):
    assert len(oracle.lsd_segments(np.full((120, 160), 77, np.uint8), oracle.lsd_opts())) == 0
    img = np.full((120, 160), 60, np.uint8)
    img[30:90, 40:120] = 180
    for scale in (1.0, 0.8, 1.2):
        seg = oracle.lsd_segments(img, oracle.lsd_opts(scale=scale))
        assert len(seg) == 4, scale


def test_wrapper_fields_and_top_n(oracle):
    """min_length, response = length / max(cols, rows), LineIterator count, angle = atan2 of the end points, and the cut: the
    lsd_nfeatures longest lines by descending response, ties in detection order; without the cut: detection order."""
    img = synth.make_image(321, 400, 240, n_rects=120, n_discs=30)
    cols, rows = 400, 240
    full = oracle.lsd_detect(img, oracle.lsd_opts(min_length=0.0, nfeatures=0))
    assert len(full) > 60
    d = np.hypot((full["sx"] - full["ex"]).astype(np.float64), (full["sy"] - full["ey"]).astype(np.float64)).astype(np.float32)
    assert np.array_equal(full["length"], d)
    assert np.array_equal(full["response"], full["length"] / np.float32(max(cols, rows)))
    assert np.allclose(full["angle"], np.arctan2((full["ey"] - full["sy"]).astype(np.float64), (full["ex"] - full["sx"]).astype(np.float64)), atol=1e-6)
    ix0, iy0, ix1, iy1 = (np.rint(full[k]).astype(int) for k in ("sx", "sy", "ex", "ey"))   # numpy rounds half to even like cvRound
    assert np.array_equal(full["num_pixels"], np.maximum(abs(ix1 - ix0), abs(iy1 - iy0)) + 1)
    assert (full["sx"] >= 0).all() and (full["ex"] <= cols - 1).all() and (full["sy"] >= 0).all() and (full["ey"] <= rows - 1).all()
    # min_length: exactly the lines longer than it, same order
    kept = oracle.lsd_detect(img, oracle.lsd_opts(min_length=12.0, nfeatures=0))
    sel = full[full["length"].astype(np.float64) > 12.0]
    assert kept.tobytes() == sel.tobytes()
    # the cut
    n = 25
    top = oracle.lsd_detect(img, oracle.lsd_opts(min_length=0.0, nfeatures=n))
    order = np.argsort(-full["response"], kind="stable")[:n]
    assert top.tobytes() == full[order].tobytes()
    assert np.all(np.diff(top["response"]) <= 0)


def test_segments_lie_on_edges(oracle):
    """The long segments of a synthetic scene lie on intensity edges: the gradient magnitude next to them is several times the image's
    MEDIAN gradient (the background is smooth; the rectangles' borders are the edges)."""
    img = synth.make_image(77, 320, 200, n_rects=60, n_discs=0, noise=1.0)
    seg = oracle.lsd_segments(img, oracle.lsd_opts())
    gy, gx = np.gradient(img.astype(np.float64))
    mag = np.hypot(gx, gy)
    long_ones = [s for s in seg if np.hypot(s[0] - s[2], s[1] - s[3]) > 15]
    assert len(long_ones) > 20
    hits = 0
    for s in long_ones:
        t = np.linspace(0.1, 0.9, 20)
        xs = np.clip(np.rint(s[0] + t * (s[2] - s[0])).astype(int), 1, 318); ys = np.clip(np.rint(s[1] + t * (s[3] - s[1])).astype(int), 1, 198)
        nb = np.max([mag[ys + dy, xs + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0)
        hits += np.median(nb) > 5 * np.median(mag)
    assert hits >= 0.95 * len(long_ones)


def test_oracle_output_is_pinned(oracle):
    """The restatement against its own committed output (tests/golden/make_lsd_golden.py): a regression pin, not a reference vector."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lsd_oracle_320x200.npz"))
    seg = oracle.lsd_segments(g["img"], oracle.lsd_opts())
    assert np.array_equal(seg, g["segments"])
    kl = oracle.lsd_detect(g["img"], oracle.lsd_opts(min_length=0.025 * 200, nfeatures=50))
    assert kl.tobytes() == g["keylines"].tobytes()
    seg1 = oracle.lsd_segments(g["img"], oracle.lsd_opts(refine=1))
    assert np.array_equal(seg1, g["segments_refine1"]) and not np.array_equal(seg1[:len(seg)], seg[:len(seg1)])


def test_detector_core_against_the_numpy_restatement(oracle):
    """tests/np_lsd.py — an independent statement of the detector core written from the published algorithm — gives the same segments,
    bit for bit and in the same order, as oracle/stvo_lsd_oracle.c at scale 1 (small images: it runs plain Python loops)."""
    import np_lsd
    rng = np.random.default_rng(5)
    total = changed = 0
    for k in range(4):
        img = np.full((72, 104), 90.0)
        for _ in range(6):
            w, h = rng.integers(10, 50), rng.integers(8, 40)
            x0, y0 = rng.integers(-5, 95), rng.integers(-5, 65)
            img[max(y0, 0):max(y0 + h, 0), max(x0, 0):max(x0 + w, 0)] = rng.uniform(20, 235)
        if k >= 2:   # slanted edges and texture: regions whose angle drifts while they grow
            yy, xx = np.mgrid[0:72, 0:104]
            img[(xx * 0.6 + yy * 0.8) % 37 < 11] += 55
            img += rng.normal(0, 2.5, img.shape)
        img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        ref = oracle.lsd_segments(img, oracle.lsd_opts(scale=1.0))
        got = np_lsd.segments(img)
        assert got.shape == ref.shape and np.array_equal(got, ref), k
        total += len(ref)
        # lsd_refine = 1 (refine_region / reduce_region_radius): the same agreement, and the mode is not a no-op on these images
        ref1 = oracle.lsd_segments(img, oracle.lsd_opts(scale=1.0, refine=1))
        got1 = np_lsd.segments(img, refine=1)
        assert got1.shape == ref1.shape and np.array_equal(got1, ref1), k
        changed += ref1.shape != ref.shape or not np.array_equal(ref1, ref)
    assert total > 30 and changed >= 1


def test_refine_with_density_threshold_zero_is_no_refinement(oracle):
    """lsd_refine = 1 with density_th = 0: every region is dense enough (density >= 0), so nothing is refined and the segments are those
    of lsd_refine = 0, bit for bit — at every scale the detector supports."""
    from stvo_amd import synth
    img = synth.make_image(77, 640, 360)
    for scale in (1.2, 1.0, 0.8):
        o1 = oracle.lsd_opts(scale=scale, refine=1)
        o1.density_th = 0.0
        a, b = oracle.lsd_segments(img, oracle.lsd_opts(scale=scale)), oracle.lsd_segments(img, o1)
        assert a.shape == b.shape and np.array_equal(a, b) and len(a) > 200


def test_line_iterator_count_clips_like_cliprect(oracle):
    """KeyLine::numOfPixels = cv::LineIterator(...).count: end points by cvRound, the line clipped to the image, 8-connected.  Against a
    plain statement of the clipping (exact rational intersection of the ideal line with the image's borders, truncated towards zero as
    cv::clipLine does with its double quotient) on random segments that start or end up to 40 pixels outside the image, on segments
    whose end point rounds to the first column / row outside (the case checkLineExtremes leaves behind: x in [cols - 0.5, cols)), and on
    segments entirely outside (count 0)."""
    from fractions import Fraction

    def count(cols, rows, sx, sy, ex, ey):
        rnd = lambda v: int(np.rint(np.float32(v)))   # cvRound: half to even
        x1, y1, x2, y2 = rnd(sx), rnd(sy), rnd(ex), rnd(ey)
        right, bottom = cols - 1, rows - 1
        code = lambda x, y: (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8
        trunc = lambda q: int(q) if q >= 0 else -int(-q)    # Fraction -> integer towards zero
        c1, c2 = code(x1, y1), code(x2, y2)
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1 & 12:
                a = 0 if c1 < 8 else bottom
                x1 += trunc(Fraction((a - y1) * (x2 - x1), y2 - y1)); y1 = a
                c1 = (x1 < 0) + (x1 > right) * 2
            if c2 & 12:
                a = 0 if c2 < 8 else bottom
                x2 += trunc(Fraction((a - y2) * (x2 - x1), y2 - y1)); y2 = a
                c2 = (x2 < 0) + (x2 > right) * 2
            if (c1 & c2) == 0 and (c1 | c2) != 0:
                if c1:
                    a = 0 if c1 == 1 else right
                    y1 += trunc(Fraction((a - x1) * (y2 - y1), x2 - x1)); x1 = a; c1 = 0
                if c2:
                    a = 0 if c2 == 1 else right
                    y2 += trunc(Fraction((a - x2) * (y2 - y1), x2 - x1)); x2 = a; c2 = 0
        if (c1 | c2) != 0:
            return 0
        assert 0 <= x1 <= right and 0 <= x2 <= right and 0 <= y1 <= bottom and 0 <= y2 <= bottom
        return max(abs(x2 - x1), abs(y2 - y1)) + 1

    rng = np.random.default_rng(12)
    cols, rows = 640, 360
    n_clipped = n_zero = 0
    for trial in range(4000):
        sx, sy = rng.uniform(-40, cols + 40), rng.uniform(-40, rows + 40)
        ex, ey = rng.uniform(-40, cols + 40), rng.uniform(-40, rows + 40)
        if trial % 5 == 0:     # what checkLineExtremes hands over: inside [0, cols) x [0, rows), an end within half a pixel of the far border
            sx, sy = rng.uniform(0, cols - 1), rng.uniform(0, rows - 1)
            ex, ey = (cols - rng.uniform(0.0, 0.5), rng.uniform(0, rows - 1)) if trial % 10 == 0 else (rng.uniform(0, cols - 1), rows - rng.uniform(0.0, 0.5))
        if trial % 7 == 0:     # entirely outside, on one side
            sx, ex = cols + rng.uniform(1, 30), cols + rng.uniform(1, 30)
        got = oracle.line_iterator_count(cols, rows, sx, sy, ex, ey)
        exp = count(cols, rows, sx, sy, ex, ey)
        # (the double quotient of cv::clipLine and the exact one truncate alike unless the quotient is an integer to within 1e-12: not in this sample)
        assert got == exp, (trial, sx, sy, ex, ey, got, exp)
        inside = all(0 <= int(np.rint(np.float32(v))) <= lim for v, lim in ((sx, cols - 1), (ex, cols - 1), (sy, rows - 1), (ey, rows - 1)))
        n_clipped += (not inside) and got > 0
        n_zero += got == 0
        if inside:
            assert got == max(abs(int(np.rint(np.float32(ex))) - int(np.rint(np.float32(sx)))), abs(int(np.rint(np.float32(ey))) - int(np.rint(np.float32(sy))))) + 1
    assert n_clipped > 500 and n_zero > 300


def test_refine_std_branches_against_the_numpy_restatement(oracle):
    """lsd_refine = 1 on images that make its branches run (a blurred scene: wide, sparse regions): regions grown again, regions cut
    back by radius over several steps, regions given up — oracle and numpy restatement agree bit for bit, at both shipped density
    thresholds and a strict one."""
    import np_lsd
    from scipy.ndimage import gaussian_filter
    from stvo_amd import synth
    base = synth.make_image(4711, 320, 200, n_rects=70, n_discs=15)[:120, :200]
    seen = {}
    for sigma, dth in ((0.0, 0.6), (1.0, 0.6), (2.0, 0.6), (2.0, 0.85)):
        img = base if sigma == 0 else np.clip(np.rint(gaussian_filter(base.astype(float), sigma)), 0, 255).astype(np.uint8)
        o = oracle.lsd_opts(scale=1.0, refine=1)
        o.density_th = dth
        ref = oracle.lsd_segments(img, o)
        st = {}
        got = np_lsd.segments(img, refine=1, density_th=dth, stats=st)
        assert got.shape == ref.shape and np.array_equal(got, ref), (sigma, dth)
        for k, v in st.items():
            seen[k] = seen.get(k, 0) + v
    assert seen.get("regrown", 0) >= 10 and seen.get("radius_steps", 0) >= 5 and seen.get("gone_after_regrowing", 0) + seen.get("gone_by_radius", 0) >= 1, seen


def test_scaled_image_sample_positions_follow_the_scale_not_the_rounded_size(oracle):
    """LSD scales with resize(img, Size(), scale, scale): cv::resize rounds only the OUTPUT size (cvRound(1241 * 1.2) = 1489) and keeps
    inv_scale = 1.2, so output column d samples source position (d + 0.5) / 1.2 - 0.5 — not (d + 0.5) * 1241 / 1489 - 0.5 (up to
    0.17 px apart at the right edge).  An independent numpy statement of the fixed-point bilinear pins both forms; sizes that scale
    to integers (640 x 200) give the same image either way."""
    cols, rows = 1241, 376
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    out = oracle.resize_linear_fxy(img, 1.2, 1.2)
    assert out.shape == (451, 1489)

    def model(img, dcols, drows, sx_, sy_):
        sc, sr = img.shape[1], img.shape[0]
        fx = (((np.arange(dcols) + 0.5) * sx_) - 0.5).astype(np.float32)
        x0 = np.floor(fx).astype(np.int64)
        fx = (fx - x0.astype(np.float32)).astype(np.float32)
        lo, hi = x0 < 0, x0 >= sc - 1
        fx[lo | hi] = 0
        x0[lo] = 0
        x0[hi] = sc - 1
        a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
        x1 = np.minimum(x0 + 1, sc - 1)
        a0[hi], a1[hi] = 2048, 0
        fy = (((np.arange(drows) + 0.5) * sy_) - 0.5).astype(np.float32)
        y0 = np.floor(fy).astype(np.int64)
        fy = (fy - y0.astype(np.float32)).astype(np.float32)
        b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
        b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
        r0, r1 = np.clip(y0, 0, sr - 1), np.clip(y0 + 1, 0, sr - 1)
        I = img.astype(np.int64)
        S0 = I[r0][:, x0] * a0 + I[r0][:, x1] * a1
        S1 = I[r1][:, x0] * a0 + I[r1][:, x1] * a1
        v = (((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2
        return np.clip(v, 0, 255).astype(np.uint8)

    assert np.array_equal(out, model(img, 1489, 451, 1.0 / 1.2, 1.0 / 1.2))
    ratio = oracle.resize_linear(img, 1489, 451)
    assert np.array_equal(ratio, model(img, 1489, 451, 1.0 / (1489 / 1241), 1.0 / (451 / 376)))
    assert np.count_nonzero(ratio != out) > 1000  # the two forms are different images at this size
    img2 = rng.integers(0, 256, (200, 640), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear_fxy(img2, 1.2, 1.2), oracle.resize_linear(img2, 768, 240))
