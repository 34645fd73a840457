"""GPU parity: K1/K2 (hamming_knn2 + nnr_mutual) through the C-ABI vs the oracle.  Bit-exact."""
import numpy as np
import pytest

from stvo_amd import synth

pytestmark = pytest.mark.gpu


def rand_desc(rng, n, entropy_bits=256):
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if entropy_bits < 256:
        keep = np.zeros(32, np.uint8)
        keep[: entropy_bits // 8] = 0xFF
        d &= keep
    return d


@pytest.mark.parametrize("n1,n2,nnr,ent", [(2000, 2000, 0.75, 256), (1999, 1777, 0.9, 256), (800, 800, 0.9, 24),
                                            (300, 257, 0.75, 16), (64, 64, 0.8, 8), (65, 63, 0.75, 256),
                                            (1, 2, 0.75, 256), (2, 1, 0.75, 256), (5, 3, 1.0, 8), (3, 5, 0.5, 256),
                                            (4096, 4095, 0.75, 256)])
def test_match_bit_exact(hip, oracle, n1, n2, nnr, ent):
    rng = np.random.default_rng(n1 * 7919 + n2)
    d2 = rand_desc(rng, n2, ent)
    d1 = rand_desc(rng, n1, ent)
    k = min(n1, n2) // 2
    if k:
        d1[:k] = synth.flip_bits(rng, d2[rng.permutation(n2)[:k]], 0.06)
    for mutual in (1, 0):
        got, n = hip.match(d1, d2, nnr, mutual)
        exp, en = oracle.match(d1, d2, nnr, mutual)
        assert np.array_equal(got, exp), (n1, n2, mutual, np.nonzero(got != exp)[0][:10])
        assert n == en


def test_match_empty_and_errors(hip):
    from stvo_amd.capi import StvoError
    z = np.zeros((0, 32), np.uint8)
    d = np.zeros((4, 32), np.uint8)
    m, n = hip.match(z, d, 0.75)
    assert len(m) == 0 and n == 0
    m, n = hip.match(d, z, 0.75)
    assert np.all(m == -1) and n == 0
    with pytest.raises(StvoError):
        hip.match(d, d, 1.5)  # nnr > 1 is outside the order-independent regime
    with pytest.raises(StvoError):
        hip.match(np.zeros((5000, 32), np.uint8), d, 0.75)  # capacity of the fixture context is 4096


def test_match_config2_frames(hip, oracle):
    for frame in range(3):
        fr = synth.make_f2f_points(synth.frame_seed(0, frame), n=2000)
        got, n = hip.match(fr["prev_desc"], fr["curr_desc"], 0.75)
        exp, en = oracle.match(fr["prev_desc"], fr["curr_desc"], 0.75)
        assert np.array_equal(got, exp) and n == en
        assert 1300 <= n <= 1520  # SURVEY.md §8(d) config 2 expectation


def test_match_all_identical_descriptors_no_match(hip, oracle):
    d = np.tile(np.arange(32, dtype=np.uint8), (300, 1))
    got, n = hip.match(d, d, 0.9)
    exp, en = oracle.match(d, d, 0.9)
    assert np.array_equal(got, exp) and n == en == 0  # best == second fails the ratio test


def test_match_size_independent_properties(hip):
    """Full-size property checks that need no oracle run: permutation equivariance and symmetry."""
    rng = np.random.default_rng(5)
    fr = synth.make_f2f_points(123, n=2000)
    d1, d2 = fr["prev_desc"], fr["curr_desc"]
    m12, _ = hip.match(d1, d2, 0.75)
    m21, _ = hip.match(d2, d1, 0.75)
    # mutual matching is symmetric: (i -> j) in 12  <=>  (j -> i) in 21
    idx = np.nonzero(m12 >= 0)[0]
    assert np.array_equal(m21[m12[idx]], idx)
    assert (m21 >= 0).sum() == len(idx)
    # permuting the train rows permutes the answer
    perm = rng.permutation(len(d2))
    mp, _ = hip.match(d1, d2[perm], 0.75)
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    exp = np.where(m12 >= 0, inv[np.maximum(m12, 0)], -1)
    assert np.array_equal(mp, exp)


def test_match_verify_shortcut_adversarial(hip, oracle):
    """The mutual check is a range query that stops a row after 128 bits when no lane can be blocked.  Inputs built
    to sit on every edge of that short cut: several claimants per column, rows that agree with a column in the first
    half only (lower bound 0, true distance large), in the second half only, and distances straddling T = d0 / nnr."""
    rng = np.random.default_rng(77)
    n2 = 640
    d2 = rand_desc(rng, n2)
    rows = []
    for j in range(0, 320):
        base = d2[j].copy()
        kind = j % 8
        if kind == 0:      # three claimants of column j at distances 3, 3 and 9 (tie between the first two)
            for nb in (3, 3, 9):
                r = base.copy(); bits = rng.permutation(256)[:nb]
                for b in bits: r[b >> 3] ^= np.uint8(1 << (b & 7))
                rows.append(r)
        elif kind == 1:    # flips only in the second half: partial distance 0
            r = base.copy(); bits = 128 + rng.permutation(128)[: int(rng.integers(1, 60))]
            for b in bits: r[b >> 3] ^= np.uint8(1 << (b & 7))
            rows.append(r)
        elif kind == 2:    # flips only in the first half
            r = base.copy(); bits = rng.permutation(128)[: int(rng.integers(1, 60))]
            for b in bits: r[b >> 3] ^= np.uint8(1 << (b & 7))
            rows.append(r)
        elif kind == 3:    # a claimant at d0 and a second row exactly at / just above the blocking threshold
            d0 = int(rng.integers(4, 40))
            for nb in (d0, int(np.floor(d0 / 0.75)) + int(rng.integers(-1, 2))):
                r = base.copy(); bits = rng.permutation(256)[:nb]
                for b in bits: r[b >> 3] ^= np.uint8(1 << (b & 7))
                rows.append(r)
        elif kind == 4:    # first half identical, second half random (lower bound 0, distance ~64)
            r = base.copy(); r[16:] = rng.integers(0, 256, 16, dtype=np.uint8)
            rows.append(r)
            r2 = base.copy(); bits = rng.permutation(256)[:45]
            for b in bits: r2[b >> 3] ^= np.uint8(1 << (b & 7))
            rows.append(r2)
        else:
            rows.append(synth.flip_bits(rng, base[None, :], 0.08)[0])
    d1 = np.stack(rows)[rng.permutation(len(rows))]
    for nnr in (0.75, 0.9, 1.0, 0.3):
        for a, b in ((d1, d2), (d2, d1)):
            got, n = hip.match(a, b, nnr, 1)
            exp, en = oracle.match(a, b, nnr, 1)
            assert np.array_equal(got, exp), (nnr, np.nonzero(got != exp)[0][:10])
            assert n == en


def test_match_matrix_core_key_edges(hip, oracle):
    """K1m builds its packed (distance, index) keys inside the matrix-core accumulator, tile-relative and rebased per
    32-row tile.  Exercise what that encoding could get wrong: distances 0 and 256, equal distances inside one tile and
    across tiles (lowest train index wins), copies of one row on both sides of every tile edge, ragged last tiles."""
    rng = np.random.default_rng(2024)
    for n2 in (2, 31, 32, 33, 63, 64, 65, 95, 97, 129, 517):
        d2 = rand_desc(rng, n2)
        # the same train row planted on both sides of tile edges: equal distances, the lower index must be reported
        for a, b in ((0, n2 - 1), (31, 32), (30, 33), (63, 64), (1, 96)):
            if b < n2:
                d2[b] = d2[a]
        if n2 > 40:
            d2[40] = 0x00    # all clear
            d2[7] = 0xFF     # all set: distance 256 to row 40
        q = [d2[j].copy() for j in range(0, n2, 3)]
        q.append(np.zeros(32, np.uint8)); q.append(np.full(32, 0xFF, np.uint8))
        q += list(synth.flip_bits(rng, d2[rng.integers(0, n2, 40)], 0.05))
        d1 = np.stack(q)
        for nnr in (0.75, 1.0):
            for mutual in (1, 0):
                for x, y in ((d1, d2), (d2, d1)):
                    got, n = hip.match(x, y, nnr, mutual)
                    exp, en = oracle.match(x, y, nnr, mutual)
                    assert np.array_equal(got, exp), (n2, nnr, mutual, np.nonzero(got != exp)[0][:10])
                    assert n == en


def test_match_low_entropy_many_ties(hip, oracle):
    """Descriptors with 6 random bits: almost every distance is shared by dozens of train rows, so the answer is
    decided by the tie order alone (strict '<' over ascending train index, oracle/stvo_oracle.c orc_knn2)."""
    rng = np.random.default_rng(99)
    d1 = np.zeros((700, 32), np.uint8); d2 = np.zeros((900, 32), np.uint8)
    d1[:, 5] = rng.integers(0, 64, 700); d2[:, 5] = rng.integers(0, 64, 900)
    d1[:, 20] = rng.integers(0, 4, 700) << 3; d2[:, 20] = rng.integers(0, 4, 900) << 3
    for mutual in (1, 0):
        got, n = hip.match(d1, d2, 1.0, mutual)
        exp, en = oracle.match(d1, d2, 1.0, mutual)
        assert np.array_equal(got, exp) and n == en


def test_match_beyond_matrix_core_index_range(oracle):
    """More than 8192 rows per side do not fit the 13 index bits of K1m's keys: the context must take the VALU kernels
    (K1 + K1v) and still agree with the oracle."""
    from stvo_amd import capi
    ctx = capi.Context(device_id=0, max_rows=8300, max_batch=1)
    try:
        rng = np.random.default_rng(8300)
        d2 = rand_desc(rng, 8300)
        d1 = rand_desc(rng, 8250)
        d1[:3000] = synth.flip_bits(rng, d2[rng.permutation(8300)[:3000]], 0.06)
        got, n = ctx.match(d1, d2, 0.75, 1)
        exp, en = oracle.match(d1, d2, 0.75, 1)
        assert np.array_equal(got, exp) and n == en
    finally:
        ctx.close()


def test_match_at_the_matrix_core_index_limit(oracle):
    """Exactly 8192 rows per side: the largest frame K1m can index (13 index bits, tile-relative keys rebased 256 times),
    with the true matches planted in the LAST rows so that the highest indices carry the answers."""
    from stvo_amd import capi
    ctx = capi.Context(device_id=0, max_rows=8192, max_batch=1)
    try:
        rng = np.random.default_rng(8192)
        d2 = rand_desc(rng, 8192)
        d1 = rand_desc(rng, 8192)
        d1[-2500:] = synth.flip_bits(rng, d2[-2500:][rng.permutation(2500)], 0.05)
        d1[:5] = d2[-5:]                      # distance 0 to the very last rows
        for mutual in (1, 0):
            got, n = ctx.match(d1, d2, 0.75, mutual)
            exp, en = oracle.match(d1, d2, 0.75, mutual)
            assert np.array_equal(got, exp) and n == en
            assert (got[-2500:] >= 8192 - 2500).sum() > 2000
    finally:
        ctx.close()


@pytest.mark.parametrize("qb", ["0", "1", "4"])
def test_match_other_kernel_forms_still_agree(qb):
    """The VALU matcher (K1 + K1v, STVO_KNN_MFMA=0) is kept for sizes K1m cannot index and as the comparison point of
    the profiles; K1m's instantiations with one and with four query blocks per wave (STVO_KNN_MFMA=1 / 4; the library's choice is
    2) share the forward kernel's text — the top-2 fold reads matrix-core accumulators through inline asm, one matrix instruction
    behind their last write when a wave holds a single block.  Run this file's parity cases against each in a child process (the
    switch is read once per process)."""
    import os, subprocess, sys
    if os.environ.get("STVO_KNN_MFMA") is not None:
        pytest.skip("already the child run")
    env = dict(os.environ, STVO_KNN_MFMA=qb)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q",
                        "-k", "bit_exact or adversarial or key_edges or many_ties or fuzz"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _flip(rng, row, nbits, lo=0, hi=256):
    r = row.copy()
    for b in lo + rng.permutation(hi - lo)[:nbits]:
        r[b >> 3] ^= np.uint8(1 << (b & 7))
    return r


def test_match_reverse_plan_adversarial(hip, oracle):
    """The matrix-core reverse check evaluates D(i', j) only for rows i' whose SECOND-best forward distance is within the
    column's blocking threshold T_j; every other potential blocker must have j among its own top-2.  Build the cases that
    separate the two routes: clusters of near-identical train rows (a blocker whose two best entries are OTHER members of the
    cluster, so column j is in nobody's top-2 but its claimant's), blockers exactly at T and T + 1, claimants that are
    themselves members of S, weak claims with large thresholds (heavy columns) next to strong ones (light columns)."""
    rng = np.random.default_rng(4242)
    seen_heavy = False
    for trial, (nnr, n_fill) in enumerate(((0.75, 900), (0.9, 300), (1.0, 64), (0.5, 1500))):
        train, query = [], []
        for c in range(60):
            base = rand_desc(rng, 1)[0]
            kind = c % 6
            if kind == 0:    # cluster j, j1, j2; i* near j; i' near j1/j2 and at distance <= T of j without j in its top-2
                j = base; j1 = _flip(rng, base, 12, 0, 64); j2 = _flip(rng, j1, 1, 64, 128)
                train += [j, j1, j2]
                d0 = int(rng.integers(6, 14))
                query.append(_flip(rng, j, d0, 128, 256))              # claimant of j
                query.append(_flip(rng, j1, int(rng.integers(0, 3)), 128, 256))  # near j1, j2; D(., j) ~ 12-14
            elif kind == 1:  # second row exactly at T / T + 1 / T - 1 of the column (T = largest blocking distance)
                d0 = int(rng.integers(3, 60)); f0 = np.float32(d0); T = d0
                while T < 256 and not (f0 < np.float32(T + 1) * np.float32(nnr)):
                    T += 1
                train.append(base)
                query.append(_flip(rng, base, d0))
                query.append(_flip(rng, base, min(256, max(d0, T + int(rng.integers(-1, 2))))))
            elif kind == 2:  # two train rows 3 bits apart, two queries each nearest to a different one: both in S
                t2 = _flip(rng, base, 3)
                train += [base, t2]
                query += [_flip(rng, base, 2, 128, 256), _flip(rng, t2, 2, 0, 128)]
            elif kind == 3:  # weak claim: distance 70-90 (heavy column when the rest of the frame is strong)
                train.append(base)
                query.append(_flip(rng, base, int(rng.integers(70, 90))))
            elif kind == 4:  # the same train row twice (tie in every query's top-2)
                train += [base, base.copy()]
                query += [_flip(rng, base, 5), _flip(rng, base, 9)]
            else:
                train.append(base)
                query.append(synth.flip_bits(rng, base[None, :], 0.07)[0])
        train += list(rand_desc(rng, n_fill)); query += list(rand_desc(rng, n_fill // 2))
        d2 = np.stack(train)[rng.permutation(len(train))]
        d1 = np.stack(query)[rng.permutation(len(query))]
        import os
        routes = np.zeros(3, np.int64)
        for a, b in ((d1, d2), (d2, d1)):
            got, n = hip.match(a, b, nnr, 1)
            exp, en = oracle.match(a, b, nnr, 1)
            assert np.array_equal(got, exp), (trial, nnr, np.nonzero(got != exp)[0][:10])
            assert n == en
            routes += hip.last_reverse_plan(1)[1:4, 0]
        if os.environ.get("STVO_KNN_MFMA") != "0":  # the inputs really took all three routes: light, heavy, non-empty S
            assert routes[0] > 0 and routes[2] > 0, routes
            seen_heavy = seen_heavy or routes[1] > 0
    if os.environ.get("STVO_KNN_MFMA") != "0":
        assert seen_heavy


def test_match_fuzz_sizes_and_entropy(hip, oracle):
    """Seeded sweep over ragged sizes (tile and segment edges of both matchers), descriptor entropies (from tie-dominated
    to random) and ratios; every case against the oracle, mutual and one-way."""
    rng = np.random.default_rng(31337)
    for case in range(120):
        n1 = int(rng.integers(1, 700)); n2 = int(rng.integers(1, 700))
        ent = int(rng.choice([8, 16, 24, 64, 256]))
        nnr = float(rng.choice([0.5, 0.75, 0.8, 0.9, 1.0]))
        d2 = rand_desc(rng, n2, ent); d1 = rand_desc(rng, n1, ent)
        k = int(rng.integers(0, min(n1, n2) + 1))
        if k:
            d1[:k] = synth.flip_bits(rng, d2[rng.permutation(n2)[:k]], float(rng.choice([0.0, 0.02, 0.06, 0.12])))
        mutual = int(case % 3 != 0)
        got, n = hip.match(d1, d2, nnr, mutual)
        exp, en = oracle.match(d1, d2, nnr, mutual)
        assert np.array_equal(got, exp), (case, n1, n2, ent, nnr, mutual, np.nonzero(got != exp)[0][:10])
        assert n == en


@pytest.mark.parametrize("kw", [dict(cluster_frac=0.6, cluster_size=8, spread_p=0.06), dict(cluster_frac=0.9, cluster_size=16, spread_p=0.04),
                                dict(cluster_frac=1.0, cluster_size=40, spread_p=0.02)])
def test_match_clustered_descriptors_2000x2000(hip, oracle, kw):
    """Descriptor rows in groups of near-duplicates (bench.py's `reverse_check_correlated` workload): second-best distances
    overlap the blocking thresholds, so the reverse check needs its light AND heavy scans with a non-empty row list S.
    Bit-exact against the oracle at 2000 x 2000, and the plan statistics show that the non-trivial routes were taken."""
    fr = synth.make_f2f_points(synth.frame_seed(9, int(kw["cluster_size"])), n=2000, desc_model="clustered", cluster_kw=kw)
    for nnr in (0.75, 0.9):
        m12, n = hip.match(fr["prev_desc"], fr["curr_desc"], nnr)
        exp, en = oracle.match(fr["prev_desc"], fr["curr_desc"], nnr)
        assert np.array_equal(m12, exp) and n == en
        plan = hip.last_reverse_plan(1)[:, 0]
        assert plan[0] > 0 and plan[3] > 0, plan          # claimed columns, rows in S
        assert plan[1] > 0 and plan[2] > 0, plan          # both light and heavy columns


@pytest.mark.parametrize("n", [5000, 8192])
def test_match_clustered_descriptors_beyond_2048_rows(oracle, n):
    """The reverse check's units beyond their comfortable size: with more than 2048 train rows a heavy column's segment is longer than one
    LDS-resident chunk (8 segments of n / 8 rows: several chunks one after the other), and with heavy clustering the row list S exceeds
    256 rows (several light segments).  Clustered rows on both sides, both ratios; bit-exact against the oracle."""
    from stvo_amd import capi
    ctx = capi.Context(device_id=0, max_rows=8192, max_batch=1)
    try:
        rng = np.random.default_rng(n)
        kw = dict(cluster_frac=0.8, cluster_size=12, spread_p=0.05)
        d2 = synth.clustered_desc(rng, n, **kw)
        d1 = synth.flip_bits(rng, d2[rng.permutation(n)], 0.06)
        d1[: n // 5] = synth.clustered_desc(rng, n // 5, **kw)      # rows without a partner
        # 600 query rows that sit between TWIN train rows (2 bits apart): no claim (the ratio test fails), but a second-best distance
        # of a few bits puts every one of them into the row list S of any useful cut — |S| > 256, several light segments
        tw = rand_desc(rng, 600)
        d2[0:1200:2] = tw
        d2[1:1200:2] = synth.flip_bits(rng, tw, 2.0 / 256)
        d1[n // 5:n // 5 + 600] = synth.flip_bits(rng, tw, 3.0 / 256)
        for nnr in (0.75, 0.95):
            got, cnt = ctx.match(d1, d2, nnr, 1)
            exp, en = oracle.match(d1, d2, nnr, 1)
            assert np.array_equal(got, exp) and cnt == en
            plan = ctx.last_reverse_plan(1)[:, 0]
            assert plan[2] > 0 and plan[1] > 0, plan                 # heavy and light columns
        assert plan[3] > 256, plan                                   # more rows in S than one LDS-resident chunk holds
    finally:
        ctx.close()
