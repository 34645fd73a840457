"""Oracle (oracle/stvo_oracle.c) matching half vs the independent numpy model and analytic cases.
Each test covers one quirk of SURVEY.md Appendix A (matching 1-12)."""
import numpy as np
import pytest

import np_model
from stvo_amd import synth


def rand_desc(rng, n, entropy_bits=256):
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if entropy_bits < 256:  # entropy-starved descriptors force many distance ties
        keep = np.zeros(32, np.uint8)
        keep[: entropy_bits // 8] = 0xFF
        d &= keep
    return d


def test_distance_is_popcount(oracle):  # A.1
    rng = np.random.default_rng(1)
    a = rand_desc(rng, 200); b = rand_desc(rng, 200)
    D = np_model.hamming_matrix(a, b)
    for i in range(0, 200, 7):
        for j in range(0, 200, 11):
            assert oracle.distance(a[i], b[j]) == D[i, j]
    z = np.zeros(32, np.uint8); f = np.full(32, 255, np.uint8)
    assert oracle.distance(z, z) == 0 and oracle.distance(z, f) == 256


@pytest.mark.parametrize("n1,n2,ent", [(37, 53, 256), (64, 64, 16), (130, 5, 8), (1, 2, 256), (300, 257, 24)])
def test_knn2_ties_lowest_index_first(oracle, n1, n2, ent):  # A.5
    rng = np.random.default_rng(n1 * 1000 + n2)
    q = rand_desc(rng, n1, ent); t = rand_desc(rng, n2, ent)
    i0, d0, d1 = oracle.knn2(q, t)
    ri0, rd0, rd1 = np_model.knn2(np_model.hamming_matrix(q, t))
    assert np.array_equal(i0, ri0) and np.array_equal(d0, rd0) and np.array_equal(d1, rd1)


def test_float_ratio_table(oracle):  # A.2: (float)d0 < (float)d1 * (float)nnr
    # build 2-row train sets with exact distances (d0, d1) from an all-zero query
    def desc_with_bits(k):
        bits = np.zeros(256, np.uint8); bits[:k] = 1
        return np.packbits(bits, bitorder="little")
    q = np.zeros((1, 32), np.uint8)
    for nnr in (0.75, 0.9, 0.6, 1.0):
        for d0 in range(0, 257, 3):
            for d1 in (d0, d0 + 1, min(256, int(d0 / nnr) + 1), min(256, int(d0 / nnr)), 256):
                if d1 < d0:
                    continue
                t = np.stack([desc_with_bits(d0), desc_with_bits(d1)])
                m, n = oracle.match_nnr(q, t, nnr)
                exp = np.float32(d0) < np.float32(d1) * np.float32(nnr)
                assert (m[0] == 0) == bool(exp), (nnr, d0, d1)
    # 0.9f is 0.89999998: d0 = 9, d1 = 10 -> 9 < 8.9999998 false in float, (9 < 9.0 false in double too);
    # d0=90,d1=100: float 90 < 100*0.9f = 89.99999.. -> false
    t = np.stack([desc_with_bits(90), desc_with_bits(100)])
    assert oracle.match_nnr(q, t, 0.9)[0][0] == -1


def test_match_nnr_degenerate_sizes(oracle):  # A.6
    rng = np.random.default_rng(3)
    q = rand_desc(rng, 5)
    m, n = oracle.match_nnr(q, rand_desc(rng, 1), 0.75)
    assert n == 0 and np.all(m == -1)
    m, n = oracle.match_nnr(q, np.zeros((0, 32), np.uint8), 0.75)
    assert n == 0 and np.all(m == -1)
    m, n = oracle.match(np.zeros((0, 32), np.uint8), q, 0.75)
    assert n == 0 and len(m) == 0


@pytest.mark.parametrize("n1,n2,nnr,ent", [(200, 180, 0.75, 256), (150, 150, 0.9, 256), (120, 90, 0.9, 24),
                                            (64, 300, 0.75, 40), (257, 129, 0.8, 256)])
def test_match_mutual_vs_numpy(oracle, n1, n2, nnr, ent):  # A.7
    rng = np.random.default_rng(n1 + 7 * n2)
    d2 = rand_desc(rng, n2, ent)
    d1 = rand_desc(rng, n1, ent)
    k = min(n1, n2) // 2
    d1[:k] = synth.flip_bits(rng, d2[rng.permutation(n2)[:k]], 0.05)
    for best_lr in (1, 0):
        m, n = oracle.match(d1, d2, nnr, best_lr)
        ref = np_model.match(d1, d2, nnr, bool(best_lr))
        assert np.array_equal(m, ref)
        assert n == int((ref >= 0).sum())


def test_match_config2_shape(oracle):
    fr = synth.make_f2f_points(synth.frame_seed(0, 1), n=400)
    m, n = oracle.match(fr["prev_desc"], fr["curr_desc"], 0.75)
    good = (m >= 0) & (m == fr["true_m12"])
    assert good.sum() >= 0.95 * (fr["true_m12"] >= 0).sum()
    assert ((m >= 0) & (m != fr["true_m12"])).sum() <= 2


def grid_case(rng, n1, n2, ent=256, spread=1.0):
    kp2 = np.stack([rng.uniform(0, 1241 * spread, n2), rng.uniform(0, 376 * spread, n2)], 1)
    kp1 = np.stack([rng.uniform(-5, 1241 * spread + 20, n1), rng.uniform(0, 376 * spread, n1)], 1)
    iw, ih = 64 / 1241.0, 48 / 376.0
    c1 = np.stack([(kp1[:, 0] * iw).astype(np.int32), (kp1[:, 1] * ih).astype(np.int32)], 1)
    c2 = np.stack([(kp2[:, 0] * iw).astype(np.int32), (kp2[:, 1] * ih).astype(np.int32)], 1)
    return c1, c2, rand_desc(rng, n1, ent), rand_desc(rng, n2, ent)


def window_candidates(c2, x, y, w):
    lo_x, hi_x = max(0, x - w[0]), min(64, x + w[1] + 1)
    lo_y, hi_y = max(0, y - w[2]), min(48, y + w[3] + 1)
    sel = (c2[:, 0] >= lo_x) & (c2[:, 0] < hi_x) & (c2[:, 1] >= lo_y) & (c2[:, 1] < hi_y)
    return np.nonzero(sel)[0]


@pytest.mark.parametrize("n1,n2,ent,w,ratio,spread", [(300, 320, 256, (10, 0, 0, 0), 0.75, 1.0),
                                                       (400, 400, 16, (10, 0, 0, 0), 0.9, 0.3),
                                                       (250, 200, 24, (10, 0, 0, 1), 0.75, 0.2),
                                                       (100, 500, 8, (3, 2, 1, 1), 1.0, 0.1)])
def test_match_grid_points_vs_numpy(oracle, n1, n2, ent, w, ratio, spread):  # A.3, A.8, A.9, A.10
    rng = np.random.default_rng(n1 * 3 + n2)
    c1, c2, d1, d2 = grid_case(rng, n1, n2, ent, spread)
    start, items = oracle.grid_build(c2)
    cands = [window_candidates(c2, x, y, w) for x, y in c1]
    for best_lr in (1, 0):
        m, n = oracle.match_grid_points(c1, d1, start, items, d2, w, ratio, best_lr)
        ref = np_model.match_grid(cands, d1, d2, ratio, bool(best_lr))
        assert np.array_equal(m, ref)
        assert n == int((ref >= 0).sum())


def test_match_grid_single_candidate_accepted_and_empty_rejected(oracle):  # A.3
    d1 = np.zeros((2, 32), np.uint8); d2 = np.full((1, 32), 255, np.uint8)
    c1 = np.array([[5, 5], [40, 40]], np.int32); c2 = np.array([[3, 5]], np.int32)
    start, items = oracle.grid_build(c2)
    m, n = oracle.match_grid_points(c1, d1, start, items, d2, (10, 0, 0, 0), 0.75, 1)
    assert list(m) == [0, -1] and n == 1  # one eligible candidate: best_d2 = INT_MAX -> accepted


def test_match_grid_lines_vs_numpy(oracle):  # A.4, A.11, A.12
    rng = np.random.default_rng(77)
    n1, n2 = 120, 140
    W, H = 1241.0, 376.0
    iw, ih = 64 / W, 48 / H
    def lines(n):
        s = np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n)], 1)
        e = s + rng.uniform(-150, 150, (n, 2))
        e[:, 0] = np.clip(e[:, 0], 0, W - 1); e[:, 1] = np.clip(e[:, 1], 0, H - 1)
        return s.astype(np.float32), e.astype(np.float32)
    s1, e1 = lines(n1); s2, e2 = lines(n2)
    e1[:10] = s1[:10] + 1.0  # same-cell end points -> NaN direction -> gate never skips
    c1 = np.concatenate([(s1[:, 0:1] * iw).astype(np.int32), (s1[:, 1:2] * ih).astype(np.int32),
                         (e1[:, 0:1] * iw).astype(np.int32), (e1[:, 1:2] * ih).astype(np.int32)], 1)
    ent, owner = [], []
    for j in range(n2):
        cells = oracle.line_coords(s2[j, 0] * iw, s2[j, 1] * ih, e2[j, 0] * iw, e2[j, 1] * ih)
        ent.append(cells); owner += [j] * len(cells)
    ent = np.concatenate(ent); owner = np.array(owner, np.int32)
    start, items = oracle.grid_build(ent, owner)
    v = np.stack([(e2[:, 0] - s2[:, 0]).astype(np.float64) * iw, (e2[:, 1] - s2[:, 1]).astype(np.float64) * ih], 1)
    dir2 = v / np.linalg.norm(v, axis=1, keepdims=True)
    d1 = rand_desc(rng, n1, 24); d2 = rand_desc(rng, n2, 24)
    w = (10, 0, 0, 0)
    inb = (ent[:, 0] >= 0) & (ent[:, 0] < 64) & (ent[:, 1] >= 0) & (ent[:, 1] < 48)
    entb, ownb = ent[inb], owner[inb]
    def cand(i):
        out = []
        for x, y in ((c1[i, 0], c1[i, 1]), (c1[i, 2], c1[i, 3])):
            lo_x, hi_x = max(0, x - w[0]), min(64, x + w[1] + 1)
            sel = (entb[:, 0] >= lo_x) & (entb[:, 0] < hi_x) & (entb[:, 1] == y) & (0 <= y < 48)
            out += list(ownb[sel])
        return sorted(set(out))
    cands = [cand(i) for i in range(n1)]
    def gate(i1, i2):
        vv = np.array([c1[i1, 2] - c1[i1, 0], c1[i1, 3] - c1[i1, 1]], float)
        with np.errstate(all="ignore"):
            vv = vv / np.sqrt(vv @ vv)
            return not (abs(vv @ dir2[i2]) < 0.75)
    for best_lr in (1, 0):
        m, n = oracle.match_grid_lines(c1, d1, start, items, d2, dir2, w, 0.75, 0.75, best_lr)
        ref = np_model.match_grid(cands, d1, d2, 0.75, bool(best_lr), gate)
        assert np.array_equal(m, ref)
    assert (m[:10] >= -1).all()


def test_pipeline_reference_thread_fanout_gives_identical_results(oracle):
    """tests/pipeline_ref.run_sequence with the reference's thread structure (points || lines, 12 || 21; what bench.py times as the
    CPU path at the reference's fan-out) is the same computation as the sequential one: identical matches, inliers and poses."""
    from concurrent.futures import ThreadPoolExecutor
    import pipeline_ref
    from stvo_amd import synth
    from stvo_amd.ctypes_types import match_params, opt_params
    cam = synth.KITTI_CAM
    seq = synth.make_stereo_sequence(4242, n_frames=4, n_pts=300, n_lines=40, cam=cam)
    mp = match_params("kitti"); op = opt_params("kitti")
    a = pipeline_ref.run_sequence(oracle, seq, cam, mp, op)
    with ThreadPoolExecutor(6) as ex:
        b = pipeline_ref.run_sequence(oracle, seq, cam, mp, op, fanout=ex)
    for x, y in zip(a, b):
        for k in ("status", "path", "iters", "n_matched_pt", "n_matched_ls", "n_inliers_pt", "n_inliers_ls", "n_stereo_pt", "n_stereo_ls"):
            assert x[k] == y[k], k
        assert np.array_equal(x["T"], y["T"]) and np.array_equal(x["inlier_p"], y["inlier_p"]) and np.array_equal(x["inlier_l"], y["inlier_l"])
