"""GPU parity of the LSD key-line detector (stvo_lsd_*, stvo-pl_amd/csrc/lsd_kernels.hip) against oracle/stvo_lsd_oracle.c:
what StereoFrame::detectLineFeatures gets from LSDDetectorC::detect + its top-N cut (/root/reference/src/stereoFrame.cpp:219-240;
3rdparty/line_descriptor/src/LSDDetector_custom.cpp:227-325).  The detector core (cv::LineSegmentDetector) is third-party code
the reference does not hold: the oracle restates the published algorithm, parity unpinned — what is pinned here is
that the HIP path reproduces the oracle bit for bit: every segment, in detection order."""
import os

import numpy as np
import pytest

from stvo_amd import synth

pytestmark = pytest.mark.gpu


def clean_image(cols, rows, seed):
    rng = np.random.default_rng(seed)
    img = np.full((rows, cols), 90, np.uint8)
    for _ in range(12):
        w, h = rng.integers(20, cols // 3), rng.integers(20, rows // 3)
        x0, y0 = rng.integers(0, cols - w), rng.integers(0, rows - h)
        img[y0:y0 + h, x0:x0 + w] = rng.integers(20, 240)
    return img


def check_keylines(got, ref, cols, rows):
    rec, resp = got
    assert len(rec) == len(ref)
    for f in ("sx", "sy", "ex", "ey"):
        assert np.array_equal(rec[f], ref[f]), f
    assert np.array_equal(rec["num_pixels"], ref["num_pixels"])
    assert np.array_equal(resp, ref["response"])
    # KeyLine::angle = atan2 in double, rounded to float: the device's atan2 and the host's may differ in the last place
    assert np.allclose(rec["angle"], ref["angle"], rtol=0, atol=4e-7)


@pytest.mark.parametrize("cols,rows,seed", [(1241, 376, 500), (752, 480, 501), (640, 480, 502)])
def test_lsd_segments_bit_exact(hip, oracle, cols, rows, seed):
    from stvo_amd import capi
    imgs = np.stack([synth.make_image(seed, cols, rows), clean_image(cols, rows, seed + 1)])
    prm = capi.lsd_params(min_length=0.025 * min(cols, rows), nfeatures=300)
    lsd = capi.Lsd(hip, 2, cols, rows, prm, max_keylines=512)
    try:
        segs, n = lsd.segments(imgs)
        dets = lsd.detect(imgs)
        for b in range(2):
            ref = oracle.lsd_segments(imgs[b], oracle.lsd_opts())
            assert n[b] == len(ref) and len(ref) > 10
            assert np.array_equal(segs[b], ref)
            kl = oracle.lsd_detect(imgs[b], oracle.lsd_opts(min_length=0.025 * min(cols, rows), nfeatures=300))
            check_keylines(dets[b], kl, cols, rows)
    finally:
        lsd.close()


def test_lsd_plain_growth_is_the_same(hip, oracle, switches):
    """STVO_LSD_GROW=0 (debug_switches.h): the plain form of lsd_grow_kernel — candidates taken one after the other, region2rect's
    sums through v_readlane — against the oracle on a scene, a clean image, noise and a flat image; the default form (guess +
    verification, sums from LDS) is what every other test of this file runs."""
    from stvo_amd import capi
    switches({"STVO_LSD_GROW": "0", "STVO_LSD_WAVES": "0"})
    cols, rows = 752, 480
    rng = np.random.default_rng(41)
    imgs = np.stack([synth.make_image(610, cols, rows), clean_image(cols, rows, 611), rng.integers(0, 255, (rows, cols), dtype=np.uint8),
                     np.full((rows, cols), 77, np.uint8)])
    for scale in (1.2, 1.0):
        lsd = capi.Lsd(hip, 4, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=scale), max_keylines=2048)
        try:
            segs, n = lsd.segments(imgs)
            for b in range(4):
                ref = oracle.lsd_segments(imgs[b], oracle.lsd_opts(scale=scale))
                assert n[b] == len(ref)
                assert np.array_equal(segs[b], ref)
        finally:
            lsd.close()


@pytest.mark.parametrize("waves", ["", "0"])
def test_lsd_many_waves_and_one_wave_per_image_are_the_same(hip, oracle, switches, waves):
    """Batches of <= 8 images run one image per XCD by default — a committing wave, a dispatcher and a feeder in one workgroup,
    speculating workgroups of four waves on the other CUs, pending regions validated at their seed's turn (lsd_kernels.hip:
    lsd_grow_xcd_kernel; CPU replay of the scheme: tools/experiments/lsd_waves_sim.c); STVO_LSD_WAVES=0 runs such a batch on the
    one-wave kernel of the large batches.  Both must give the oracle's segments in the oracle's order."""
    from stvo_amd import capi
    switches({"STVO_LSD_WAVES": waves} if waves else {})
    cols, rows = 752, 480
    rng = np.random.default_rng(43)
    imgs = np.stack([synth.make_image(620, cols, rows), clean_image(cols, rows, 621), rng.integers(0, 255, (rows, cols), dtype=np.uint8),
                     np.full((rows, cols), 77, np.uint8)])
    for scale in (1.2, 1.0):
        lsd = capi.Lsd(hip, 4, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=scale), max_keylines=2048)
        try:
            for _ in range(2):  # (the second call runs on the scratch the first one left)
                segs, n = lsd.segments(imgs)
                refs = [oracle.lsd_segments(imgs[b], oracle.lsd_opts(scale=scale)) for b in range(4)]
                assert list(n) == [len(r) for r in refs], (scale, _)
                for b in range(4):
                    assert np.array_equal(segs[b], refs[b]), (scale, _, b)
        finally:
            lsd.close()


def test_lsd_first_call_of_a_new_detector(hip, oracle):
    """A detector's FIRST call right after its creation, many times over: the creation zeroes the detector's device memory, and that
    fill has to be over before the first images arrive (it once ran on the null stream, which the context's non-blocking stream does
    not wait for: on some boxes the first image of a new detector came back without a segment in about every second process)."""
    from stvo_amd import capi
    cols, rows, B = 640, 360, 20
    rng = np.random.default_rng(57)
    imgs = np.stack([rng.integers(0, 255, (rows, cols), dtype=np.uint8) if b % 5 == 0 else synth.make_image(680 + b, cols, rows) for b in range(B)])
    refs = [len(oracle.lsd_segments(imgs[b], oracle.lsd_opts(scale=0.8))) for b in range(B)]
    for rep in range(12):
        lsd = capi.Lsd(hip, B, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=0.8), max_keylines=2048)
        try:
            _, n = lsd.segments(imgs)
            assert list(n) == refs, rep
        finally:
            lsd.close()


def test_lsd_mid_batch_several_images_per_xcd(hip, oracle):
    """Batches of 9 .. 128 images: several images per XCD, each with its committer's workgroup and as many speculating workgroups as
    the XCD's 32 CUs allow (the pseudo-ordering then comes from the segmented sort of the large batches): every image against the oracle."""
    from stvo_amd import capi
    cols, rows, B = 640, 360, 20
    rng = np.random.default_rng(53)
    imgs = np.stack([synth.make_image(640 + b, cols, rows) if b % 5 else rng.integers(0, 255, (rows, cols), dtype=np.uint8) for b in range(B)])
    lsd = capi.Lsd(hip, B, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=0.8), max_keylines=2048)
    try:
        for _ in range(2):
            segs, n = lsd.segments(imgs)
            for b in range(B):
                ref = oracle.lsd_segments(imgs[b], oracle.lsd_opts(scale=0.8))
                assert n[b] == len(ref), b
                assert np.array_equal(segs[b], ref), b
    finally:
        lsd.close()


def test_lsd_two_detectors_at_once(oracle):
    """Two contexts (two HIP streams), each with its own small-batch detector, called from two threads at the same time: the
    many-waves launches of both compete for the CUs of the same XCDs — a speculating workgroup may start late or next to another
    launch's, a committer may run with fewer helpers — and every call must still return the oracle's segments (the committer alone
    is the sequential search; every wait in the kernel is bounded)."""
    import threading
    from stvo_amd import capi
    cols, rows = 640, 360
    imgs = [np.stack([synth.make_image(660 + 2 * t, cols, rows), synth.make_image(661 + 2 * t, cols, rows)]) for t in range(2)]
    refs = [[oracle.lsd_segments(im[b], oracle.lsd_opts(scale=0.8)) for b in range(2)] for im in imgs]
    ctxs = [capi.Context(device_id=0, max_rows=2048, max_batch=4) for _ in range(2)]
    lsds = [capi.Lsd(c, 2, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=0.8), max_keylines=2048) for c in ctxs]
    out, err = [None, None], []

    def work(t):
        try:
            res = []
            for _ in range(6):
                res.append(lsds[t].segments(imgs[t]))
            out[t] = res
        except Exception as e:  # noqa: BLE001
            err.append(e)

    try:
        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join(timeout=120)
        assert not err, err
        for t in range(2):
            assert out[t] is not None
            for segs, n in out[t]:
                for b in range(2):
                    assert n[b] == len(refs[t][b])
                    assert np.array_equal(segs[b], refs[t][b])
    finally:
        for l in lsds:
            l.close()
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("knobs", [{"STVO_LSD_XCD_BLOCKS": "0"}, {"STVO_LSD_XCD_BLOCKS": "1", "STVO_LSD_FEED_AHEAD": "0"},
                                   {"STVO_LSD_XCD_BLOCKS": "31", "STVO_LSD_SEP": "4", "STVO_LSD_AHEAD": "1000000", "STVO_LSD_FEED_AHEAD": "100000"}])
def test_lsd_xcd_kernel_under_hostile_settings(hip, oracle, switches, knobs):
    """The many-waves form must be exact whatever the speculation does: no speculating workgroup at all (the committer grows every
    region itself, through its LDS bitmap), a feeder that is never ahead (every region through the table), and 124 speculating waves
    crowding each other (seeds 4 px apart, no limit on their lead: most regions fail validation and are grown again) — repeated, so
    that different interleavings are seen."""
    from stvo_amd import capi
    switches(knobs)
    cols, rows = 640, 360
    rng = np.random.default_rng(47)
    imgs = np.stack([synth.make_image(630, cols, rows), rng.integers(0, 255, (rows, cols), dtype=np.uint8)])
    refs = [oracle.lsd_segments(imgs[b], oracle.lsd_opts(scale=0.8)) for b in range(2)]
    lsd = capi.Lsd(hip, 2, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=0.8), max_keylines=2048)
    try:
        for _ in range(4):
            segs, n = lsd.segments(imgs)
            for b in range(2):
                assert n[b] == len(refs[b])
                assert np.array_equal(segs[b], refs[b])
    finally:
        lsd.close()


@pytest.mark.parametrize("n_bins", [1, 7, 1000, 2048])
def test_lsd_pseudo_ordering_by_bin_count(hip, oracle, n_bins):
    """The pseudo-ordering is a counting sort of the library's own (lsd_hist / lsd_scan / lsd_scatter_kernel): the defined pixels by bin
    of the gradient norm, highest bin first, row-major inside a bin.  The order of the seeds decides every region, so the segments
    in detection order are its test — for one bin (pure row-major order), a handful, a count that is not a power of two and the
    largest the library takes; a flat image (no defined pixel at all) and one of noise (nearly every pixel defined) ride along."""
    from stvo_amd import capi
    cols, rows = 640, 480
    rng = np.random.default_rng(59)
    imgs = np.stack([synth.make_image(700, cols, rows), clean_image(cols, rows, 701), np.full((rows, cols), 90, np.uint8),
                     rng.integers(0, 255, (rows, cols), dtype=np.uint8)])
    lsd = capi.Lsd(hip, 4, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, n_bins=n_bins), max_keylines=2048)
    try:
        segs, n = lsd.segments(imgs)
        for b in range(4):
            ref = oracle.lsd_segments(imgs[b], oracle.lsd_opts(n_bins=n_bins))
            assert n[b] == len(ref), b
            assert np.array_equal(segs[b], ref), b
    finally:
        lsd.close()


def test_lsd_keep_all_flat_and_unscaled(hip, oracle):
    """nfeatures = 0 keeps every line in detection order; a flat image and pure noise give none / few; scale 1 skips the blur."""
    from stvo_amd import capi
    cols, rows = 320, 200
    rng = np.random.default_rng(9)
    imgs = np.stack([np.full((rows, cols), 128, np.uint8), clean_image(cols, rows, 3),
                     rng.integers(0, 255, (rows, cols), dtype=np.uint8), synth.make_image(77, cols, rows, n_rects=80, n_discs=20)])
    for scale in (1.2, 1.0, 0.8):
        prm = capi.lsd_params(min_length=5.0, nfeatures=0, scale=scale)
        lsd = capi.Lsd(hip, 4, cols, rows, prm, max_keylines=1024)
        try:
            dets = lsd.detect(imgs)
            for b in range(4):
                kl = oracle.lsd_detect(imgs[b], oracle.lsd_opts(min_length=5.0, nfeatures=0, scale=scale))
                check_keylines(dets[b], kl, cols, rows)
            assert len(dets[0][0]) == 0 and len(dets[1][0]) >= 20
        finally:
            lsd.close()


@pytest.mark.parametrize("B", [4, 20])
def test_lsd_refine_std_vs_oracle(hip, oracle, B):
    """lsd_refine = 1 (LSD_REFINE_STD: a region too sparse for its rectangle gives its pixels back, is grown again under a tolerance
    from the angles near its seed, then cut back by radius — oracle/stvo_lsd_oracle.c: refine_region / reduce_region_radius) runs on its
    own one-wave kernel for every batch size (lsd_kernels.hip: lsd_grow_refine_kernel): segments in the oracle's order, bit for bit, the
    wrapper's key-lines too; and the mode changes something (the refinement branches ran)."""
    from stvo_amd import capi
    cols, rows = 640, 360
    rng = np.random.default_rng(71)
    from scipy.ndimage import gaussian_filter

    def blurred(seed, sigma):  # wide, sparse regions: the ones the refinement cuts back by radius (tests/test_oracle_lsd.py counts the branches)
        return np.clip(np.rint(gaussian_filter(synth.make_image(seed, cols, rows).astype(float), sigma)), 0, 255).astype(np.uint8)
    imgs = np.stack([synth.make_image(700 + b, cols, rows) if b % 5 == 0 else blurred(700 + b, 1.0 + 0.25 * (b % 4)) if b % 5 == 1 else
                     clean_image(cols, rows, 700 + b) if b % 5 == 2 else rng.integers(0, 255, (rows, cols), dtype=np.uint8) if b % 5 == 3 else
                     np.full((rows, cols), 50 + b, np.uint8) for b in range(B)])
    changed = 0
    for scale, dth in ((1.2, 0.6), (1.0, 0.6), (0.8, 0.6), (1.0, 0.85)):
        prm = capi.lsd_params(min_length=4.0, nfeatures=0, scale=scale, refine=1)
        prm.density_th = dth
        opts = oracle.lsd_opts(min_length=4.0, nfeatures=0, scale=scale, refine=1)
        opts.density_th = dth
        lsd = capi.Lsd(hip, B, cols, rows, prm, max_keylines=2048)
        try:
            for _ in range(2):
                segs, n = lsd.segments(imgs)
                refs = [oracle.lsd_segments(imgs[b], opts) for b in range(B)]
                assert list(n) == [len(r) for r in refs], (scale, dth, _)
                for b in range(B):
                    assert np.array_equal(segs[b], refs[b]), (scale, dth, _, b)
            for b in (0, 1):
                plain = oracle.lsd_segments(imgs[b], oracle.lsd_opts(scale=scale))
                changed += plain.shape != refs[b].shape or not np.array_equal(plain, refs[b])
            dets = lsd.detect(imgs)
            for b in range(min(B, 5)):
                check_keylines(dets[b], oracle.lsd_detect(imgs[b], opts), cols, rows)
        finally:
            lsd.close()
    assert changed >= 6
    # density_th = 0: nothing is ever refined — the refinement kernel (one wave per image, plain form) must then give exactly what the
    # default forms of the search give without refinement (many waves per image here): two different kernels, one answer
    prm0 = capi.lsd_params(min_length=4.0, nfeatures=0, scale=1.2, refine=1)
    prm0.density_th = 0.0
    a = capi.Lsd(hip, B, cols, rows, prm0, max_keylines=2048)
    b = capi.Lsd(hip, B, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=1.2), max_keylines=2048)
    try:
        (sa, na), (sb, nb) = a.segments(imgs), b.segments(imgs)
        assert list(na) == list(nb) and all(np.array_equal(x, y) for x, y in zip(sa, sb)) and sum(na) > 1000
    finally:
        a.close(); b.close()


def test_lsd_keylines_whose_end_rounds_outside_the_image(hip, oracle):
    """checkLineExtremes (LSDDetector_custom.cpp:75-100) leaves an end point in [cols - 0.5, cols) or [rows - 0.5, rows) as it is, cvRound
    puts it on the first position OUTSIDE the image and cv::LineIterator clips the line: numOfPixels is one less than the rounded end
    points say (found in round 6 by the refinement test's images; until then the device counted without clipping)."""
    from scipy.ndimage import gaussian_filter
    from stvo_amd import capi
    cols, rows = 640, 360

    def blurred(seed, sigma):
        return np.clip(np.rint(gaussian_filter(synth.make_image(seed, cols, rows).astype(float), sigma)), 0, 255).astype(np.uint8)
    imgs = np.stack([blurred(701, 1.25), synth.make_image(702, cols, rows), blurred(703, 1.75)])
    outside = 0
    for scale in (1.2, 1.0):
        lsd = capi.Lsd(hip, 3, cols, rows, capi.lsd_params(min_length=4.0, nfeatures=0, scale=scale), max_keylines=2048)
        try:
            dets = lsd.detect(imgs)
            for b in range(3):
                kl = oracle.lsd_detect(imgs[b], oracle.lsd_opts(min_length=4.0, nfeatures=0, scale=scale))
                check_keylines(dets[b], kl, cols, rows)
                outside += int(np.sum((np.rint(kl["sx"]) >= cols) | (np.rint(kl["ex"]) >= cols) | (np.rint(kl["sy"]) >= rows) | (np.rint(kl["ey"]) >= rows)))
        finally:
            lsd.close()
    assert outside >= 4


def test_lsd_rejects_what_is_not_built(hip):
    from stvo_amd import capi
    from stvo_amd.capi import StvoError
    with pytest.raises(StvoError):
        capi.Lsd(hip, 1, 320, 200, capi.lsd_params(refine=2))          # LSD_REFINE_ADV (rect_improve + the NFA test): not built
    with pytest.raises(StvoError):
        capi.Lsd(hip, 1, 2000, 1000, capi.lsd_params())                # more than 2^20 pixels after scaling


def test_lsd_capacity_cut_keeps_the_strongest_and_is_reported(hip, oracle):
    """lsd_nfeatures = 0 ("keep all") against a capacity of 40 key-lines: the 40 lines of highest response come back in the order of the
    top-N cut (what nfeatures = 40 gives), not the first 40 in detection order, and stvo_lsd_counts reports how many passed min_length."""
    from stvo_amd import capi
    cols, rows = 640, 240
    img = synth.make_image(950, cols=cols, rows=rows, n_rects=80, n_discs=10)
    full = oracle.lsd_detect(img, oracle.lsd_opts(min_length=8.0, nfeatures=0))
    want = oracle.lsd_detect(img, oracle.lsd_opts(min_length=8.0, nfeatures=40))
    assert len(full) > 60
    lsd = capi.Lsd(hip, 1, cols, rows, capi.lsd_params(min_length=8.0, nfeatures=0), max_keylines=40)
    try:
        det = lsd.detect(img[None])[0]
        check_keylines(det, want, cols, rows)
        n_seg, n_pass = lsd.counts()
        assert n_pass[0] == len(full) and n_seg[0] >= n_pass[0]
    finally:
        lsd.close()
