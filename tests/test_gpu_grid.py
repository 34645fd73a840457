"""GPU parity: K3 grid-windowed matchers (points and lines) through the C-ABI vs the oracle. Bit-exact."""
import numpy as np
import pytest

from stvo_amd import synth
from stvo_amd.ctypes_types import match_params

pytestmark = pytest.mark.gpu


def rand_desc(rng, n, entropy_bits=256):
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if entropy_bits < 256:
        keep = np.zeros(32, np.uint8)
        keep[: entropy_bits // 8] = 0xFF
        d &= keep
    return d


def grid_case(rng, n1, n2, ent=256, spread=1.0):
    kp2 = np.stack([rng.uniform(0, 1241 * spread, n2), rng.uniform(0, 376 * spread, n2)], 1)
    kp1 = np.stack([rng.uniform(-30, 1241 * spread + 250, n1), rng.uniform(-10, 376 * spread + 10, n1)], 1)
    iw, ih = 64 / 1241.0, 48 / 376.0
    c1 = np.stack([(kp1[:, 0] * iw).astype(np.int32), (kp1[:, 1] * ih).astype(np.int32)], 1)
    c2 = np.stack([(kp2[:, 0] * iw).astype(np.int32), (kp2[:, 1] * ih).astype(np.int32)], 1)
    return c1, c2, rand_desc(rng, n1, ent), rand_desc(rng, n2, ent)


@pytest.mark.parametrize("n1,n2,ent,w,ratio,spread", [(2000, 2000, 256, (10, 0, 0, 0), 0.75, 1.0),
                                                       (2000, 1900, 16, (10, 0, 0, 0), 0.9, 0.3),
                                                       (1500, 1200, 24, (10, 0, 0, 1), 0.75, 0.2),
                                                       (300, 500, 8, (3, 2, 1, 1), 1.0, 0.1),
                                                       (200, 200, 16, (70, 70, 50, 50), 0.8, 1.0),
                                                       (65, 63, 256, (10, 0, 0, 0), 0.75, 1.0),
                                                       (1, 1, 256, (64, 64, 48, 48), 0.75, 1.0),
                                                       (4096, 4096, 32, (10, 0, 0, 0), 0.75, 1.0)])
def test_grid_points_bit_exact(hip, oracle, n1, n2, ent, w, ratio, spread):
    rng = np.random.default_rng(n1 * 3 + n2)
    c1, c2, d1, d2 = grid_case(rng, n1, n2, ent, spread)
    start, items = oracle.grid_build(c2)
    for mutual in (1, 0):
        got, n = hip.match_grid_points(c1, d1, start, items, d2, w, ratio, mutual)
        exp, en = oracle.match_grid_points(c1, d1, start, items, d2, w, ratio, mutual)
        assert np.array_equal(got, exp), (mutual, np.nonzero(got != exp)[0][:10])
        assert n == en


def test_grid_points_duplicate_and_out_of_range_items(hip, oracle):
    """A train index listed in several cells must count once (unordered_set); ids outside [0,n2) are skipped."""
    rng = np.random.default_rng(9)
    n1, n2 = 400, 300
    c1, c2, d1, d2 = grid_case(rng, n1, n2, 16, 0.3)
    ent = np.concatenate([c2, c2 + np.array([1, 0]), rng.integers(0, 20, (50, 2))]).astype(np.int32)
    owner = np.concatenate([np.arange(n2), np.arange(n2), rng.integers(-5, n2 + 20, 50)]).astype(np.int32)
    start, items = oracle.grid_build(ent, owner)
    got, n = hip.match_grid_points(c1, d1, start, items, d2, (10, 0, 0, 0), 0.9, 1)
    exp, en = oracle.match_grid_points(c1, d1, start, items, d2, (10, 0, 0, 0), 0.9, 1)
    assert np.array_equal(got, exp) and n == en


def test_grid_points_empty(hip):
    start = np.zeros(64 * 48 + 1, np.int32)
    d = np.zeros((3, 32), np.uint8)
    m, n = hip.match_grid_points(np.zeros((3, 2), np.int32), d, start, np.zeros(0, np.int32), np.zeros((0, 32), np.uint8), (10, 0, 0, 0), 0.75)
    assert list(m) == [-1, -1, -1] and n == 0
    m, n = hip.match_grid_points(np.zeros((0, 2), np.int32), np.zeros((0, 32), np.uint8), start, np.zeros(0, np.int32), d, (10, 0, 0, 0), 0.75)
    assert len(m) == 0 and n == 0


def make_lines(rng, n, W=1241.0, H=376.0):
    s = np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n)], 1)
    e = s + rng.uniform(-150, 150, (n, 2))
    e[:, 0] = np.clip(e[:, 0], 0, W - 1); e[:, 1] = np.clip(e[:, 1], 0, H - 1)
    return s.astype(np.float32), e.astype(np.float32)


@pytest.mark.parametrize("n1,n2,ent,th", [(120, 140, 24, 0.75), (300, 300, 256, 0.75), (300, 280, 8, 0.3), (100, 64, 16, 0.95)])
def test_grid_lines_bit_exact(hip, oracle, n1, n2, ent, th):
    rng = np.random.default_rng(77 + n1)
    iw, ih = 64 / 1241.0, 48 / 376.0
    s1, e1 = make_lines(rng, n1); s2, e2 = make_lines(rng, n2)
    e1[:10] = s1[:10] + 1.0  # same-cell end points -> NaN direction -> the gate never skips
    c1 = np.concatenate([(s1[:, 0:1] * iw).astype(np.int32), (s1[:, 1:2] * ih).astype(np.int32),
                         (e1[:, 0:1] * iw).astype(np.int32), (e1[:, 1:2] * ih).astype(np.int32)], 1)
    ent_xy, owner = [], []
    for j in range(n2):
        cells = oracle.line_coords(s2[j, 0] * iw, s2[j, 1] * ih, e2[j, 0] * iw, e2[j, 1] * ih)
        ent_xy.append(cells); owner += [j] * len(cells)
    start, items = oracle.grid_build(np.concatenate(ent_xy), np.array(owner, np.int32))
    v = np.stack([(e2[:, 0] - s2[:, 0]).astype(np.float64) * iw, (e2[:, 1] - s2[:, 1]).astype(np.float64) * ih], 1)
    dir2 = v / np.linalg.norm(v, axis=1, keepdims=True)
    d1 = rand_desc(rng, n1, ent); d2 = rand_desc(rng, n2, ent)
    for mutual in (1, 0):
        got, n = hip.match_grid_lines(c1, d1, start, items, d2, dir2, (10, 0, 0, 0), 0.75, th, mutual)
        exp, en = oracle.match_grid_lines(c1, d1, start, items, d2, dir2, (10, 0, 0, 0), 0.75, th, mutual)
        assert np.array_equal(got, exp), (mutual, np.nonzero(got != exp)[0][:10])
        assert n == en
