"""CPU checks of the ORB front-end restatement (oracle/stvo_orb_oracle.c): against the committed golden vectors, and piece
by piece against independent numpy statements of the same definitions (FAST-9 score by its definition, the fixed-point
Gaussian blur, the angle polynomial, the rotated BRIEF tests).  The restatement itself is unpinned against OpenCV (absent)."""
import os

import numpy as np

from stvo_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "orb_goldens.npz")
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2),
          (-1, 3)]


def test_orb_oracle_vs_committed_goldens(oracle):
    g = np.load(GOLD)
    assert np.array_equal(oracle.orb_default_pattern(), g["pattern"])
    for c, (seed, cols, rows, nf, th) in enumerate(g["cases"]):
        img = g[f"img_{c}"]
        assert np.array_equal(img, synth.make_image(int(seed), cols=int(cols), rows=int(rows), n_rects=40, n_discs=12))
        r = oracle.orb_detect(img, nfeatures=int(nf), fast_th=int(th))
        for k in ("kp", "response", "angle", "desc"):
            assert np.array_equal(r[k], g[f"{k}_{c}"]), (c, k)


def test_fast_score_is_the_largest_threshold_of_the_definition(oracle):
    img = synth.make_image(5, cols=96, rows=64, n_rects=14, n_discs=5, noise=6.0)
    sc = oracle.fast_scores(img, 10)
    I = img.astype(np.int32)
    n_corner = 0
    for y in range(3, 64 - 3):
        for x in range(3, 96 - 3):
            ring = np.array([I[y + dy, x + dx] for dx, dy in CIRCLE]) - I[y, x]
            ext = np.concatenate([ring, ring[:8]])

            def corner(t):
                b = ext > t; d = ext < -t
                return any(b[k:k + 9].all() or d[k:k + 9].all() for k in range(16))
            best = -1
            for t in range(0, 255):
                if corner(t):
                    best = t
                else:
                    break
            exp = best if best >= 10 else 0
            assert sc[y, x] == exp, (x, y, sc[y, x], exp)
            n_corner += exp > 0
    assert n_corner > 20 and sc[:3].max() == 0 and sc[:, :3].max() == 0


def test_blur_and_angle_pieces(oracle):
    img = synth.make_image(6, cols=80, rows=50, n_rects=10, n_discs=4)
    k = np.exp(-(np.arange(7) - 3.0) ** 2 / 8.0)
    ki = np.rint((k / k.sum()).astype(np.float32) * 256.0).astype(np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")   # numpy 'reflect' = BORDER_REFLECT_101
    h = sum(ki[i] * pad[:, i:i + 80] for i in range(7))
    v = sum(ki[i] * h[i:i + 50, :] for i in range(7))
    assert np.array_equal(oracle.gaussian_blur7(img), np.clip((v + 32768) >> 16, 0, 255).astype(np.uint8))
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.normal(0, 1e4, 2).astype(np.float32)
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        got = oracle.fast_atan2(float(y), float(x))
        assert min(abs(got - ref), 360 - abs(got - ref)) < 0.3   # OpenCV documents ~0.3 degrees
    assert oracle.fast_atan2(0.0, 0.0) == 0.0


def test_descriptor_bits_from_an_independent_statement(oracle):
    img = synth.make_image(7, cols=200, rows=120, n_rects=30, n_discs=8)
    r = oracle.orb_detect(img, nfeatures=60, fast_th=15)
    blur = oracle.gaussian_blur7(img).astype(np.int32)
    pat = oracle.orb_default_pattern().astype(np.float32)
    assert len(r["kp"]) >= 30
    # key-points: strict 3x3 maxima of the score image inside the border, row-major, cut with ties
    sc = oracle.fast_scores(img, 15).astype(np.int32)
    cand = []
    for y in range(19, 120 - 19):
        for x in range(19, 200 - 19):
            s = sc[y, x]
            nb = sc[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
            if s > 0 and s > nb.max():
                cand.append((y, x, s))
    resp = sorted((c[2] for c in cand), reverse=True)
    cut = resp[59] if len(resp) >= 60 else 1
    exp = [(x, y, s) for y, x, s in cand if s >= cut]
    assert [(int(k[0]), int(k[1])) for k in r["kp"]] == [(x, y) for x, y, _ in exp]
    assert np.array_equal(r["response"], np.array([s for _, _, s in exp], np.float32))
    for n in range(len(r["kp"])):
        x, y = int(r["kp"][n, 0]), int(r["kp"][n, 1])
        rad = np.float32(r["angle"][n]) * np.float32(np.pi / 180.0)
        a, b = np.float32(np.cos(np.float64(rad))), np.float32(np.sin(np.float64(rad)))
        ix0 = np.rint(pat[:, 0] * a - pat[:, 1] * b).astype(int); iy0 = np.rint(pat[:, 0] * b + pat[:, 1] * a).astype(int)
        ix1 = np.rint(pat[:, 2] * a - pat[:, 3] * b).astype(int); iy1 = np.rint(pat[:, 2] * b + pat[:, 3] * a).astype(int)
        bits = (blur[y + iy0, x + ix0] < blur[y + iy1, x + ix1]).astype(np.uint8)
        assert np.array_equal(np.packbits(bits, bitorder="little"), r["desc"][n]), n


def test_image_sequences_drive_the_oracle_chain(oracle):
    """The CPU side of tests/test_gpu_images.py on its own: ORB oracle on synthetic layered stereo image sequences, its key-points
    through the oracle pipeline — stereo disparities are the layers', and committed poses translate along x."""
    import pipeline_ref
    from stvo_amd import synth
    from stvo_amd.ctypes_types import match_params, opt_params
    cam = dict(synth.KITTI_CAM, width=640, height=240)
    mp = match_params("kitti"); op = opt_params("kitti", has_lines=0)
    pairs = synth.make_stereo_image_sequence(50, 4, cam)
    frames = []
    for left, right in pairs:
        l, r = oracle.orb_detect(left), oracle.orb_detect(right)
        frames.append(dict(kp_l=l["kp"], oct_l=np.zeros(len(l["kp"]), np.int32), desc_l=l["desc"], kp_r=r["kp"], desc_r=r["desc"],
                           kl_l=np.zeros((0, 4), np.float32), oct_ll=np.zeros(0, np.int32), ldesc_l=np.zeros((0, 32), np.uint8),
                           kl_r=np.zeros((0, 4), np.float32), ldesc_r=np.zeros((0, 32), np.uint8)))
    st = pipeline_ref.stereo_frame(oracle, frames[0], cam, mp, True, False)
    disp = np.round(cam["fx"] * cam["b"] / st["P"][:, 2]).astype(int)
    assert np.isin(disp, [8, 12, 16, 24, 32]).mean() > 0.9 and len(st["P"]) > 300   # a few wrong associations survive the filters
    res = pipeline_ref.run_sequence(oracle, frames, cam, mp, op)
    good = [o for o in res if o["status"] == 0]
    assert len(good) >= 2
    for o in good:
        assert 0.05 < abs(o["T"][0, 3]) < 0.3 and abs(o["T"][1, 3]) < 0.05


def test_lbd_oracle_pieces_against_independent_statements(oracle):
    """oracle/stvo_lbd_oracle.c piece by piece against numpy statements of the definitions: the fixed-point 5 x 5 blur, the Sobel
    derivatives, the weight tables (with the source's integer divisions), and the binary form of a float descriptor."""
    from stvo_amd import synth
    img = synth.make_image(3, cols=97, rows=61, n_rects=30, n_discs=8)
    k = np.exp(-(np.arange(5) - 2.0) ** 2 / 2.0); k /= k.sum()
    ki = np.rint(k.astype(np.float32).astype(np.float64) * 256.0).astype(np.int64)
    pad = np.pad(img.astype(np.int64), 2, mode="reflect")
    h = sum(ki[i] * pad[:, i:i + 97] for i in range(5))
    v = sum(ki[i] * h[i:i + 61, :] for i in range(5))
    assert np.array_equal(oracle.gaussian_blur5(img), np.clip((v + (1 << 15)) >> 16, 0, 255).astype(np.uint8))
    p = np.pad(img.astype(np.int64), 1, mode="reflect")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    odx, ody = oracle.sobel3(img)
    assert np.array_equal(odx, dx) and np.array_equal(ody, dy)
    cl, cg = oracle.lbd_tables()
    assert np.allclose(cl, np.exp(-(np.arange(21) - 10.0) ** 2 / (2 * 7.0 ** 2)), rtol=1e-15)   # u = 20 // 2, sigma = 15 // 2
    assert np.allclose(cg, np.exp(-(np.arange(63) - 31.0) ** 2 / (2 * 31.0 ** 2)), rtol=1e-15)
    # binary form: byte c = sum 2^i [f(band a)[i] > f(band b)[i]] over the 32 pairs of the source's table
    rng = np.random.default_rng(1)
    lines = np.array([[20, 20, 80, 45, np.arctan2(25, 60)]], np.float32)
    desc, df = oracle.lbd_compute(img, lines, np.array([61], np.int32), want_float=True)
    pairs = [(a, b) for a in range(9) for b in range(a + 1, 9) if not (a < 2 and b > 6)]
    assert len(pairs) == 32
    f = df[0].reshape(9, 8)
    want = [sum((1 << i) for i in range(8) if f[a][i] > f[b][i]) for a, b in pairs]
    assert list(desc[0]) == want and abs(np.linalg.norm(df[0]) - 1.0) < 1e-5
