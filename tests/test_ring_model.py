"""CPU: the band of scan_pairs_ring_kernel (tests/ring_model.py: the diagonal band of a threshold K cut into 64-row
blocks, "+1 per row" starts, +1 from an upstream outside the band, the score decode) is exact up to K and above K
otherwise, and ring_max_k(G) keeps consecutive tenants of a ring lane apart.  The model restates the kernel's rules,
not the kernel; the oracle is the judge."""
import random

import pytest

from ring_model import band, banded_nw, lives, ring_fits, ring_lanes_nw, ring_max_k

ACGT = b"ACGT"


def _mutate(rng, s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(ACGT)); out.append(ch); continue
        if x < rate:
            out.append(rng.choice(ACGT)); continue
        out.append(ch)
    return bytes(out) or b"A"


@pytest.mark.parametrize("G", [4, 8, 16, 21, 32, 64])
def test_ring_max_k_keeps_tenants_apart(G):
    rng = random.Random(G)
    K = ring_max_k(G)
    for _ in range(300):
        m = rng.randrange(1, 64 * 3 * G)
        T = max(1, m + rng.randrange(-min(K, m - 1), K + 1))
        assert ring_fits(m, T, K, G), (m, T, K, G)
        # and never more than G blocks at work in one STEP (block b updates column step - b)
        lv = lives(m, T, K)
        for step in (0, T // 3, T // 2, T - 1, T + len(lv) // 2):
            assert sum(1 for b, x in enumerate(lv) if x and x[0] <= step - b <= x[1]) <= G
    # the bound is tight: one diagonal more and the tenants of a lane collide (the whole-wave ring gives a lane away)
    assert not ring_fits(64 * 4 * G, 64 * 4 * G, 65 * G - 62, G)
    assert ring_fits(64 * 4 * G, 64 * 4 * G, 65 * G - 64, G)


def test_band_formula():
    assert band(100, 100, 10) == (-5, 5)
    assert band(100, 130, 40) == (-5, 35)
    assert band(130, 100, 40) == (-35, 5)
    assert lives(200, 200, 0) == [(0, 63), (64, 127), (128, 191), (192, 199)]


@pytest.mark.parametrize("seed", range(6))
def test_banded_score_is_exact_up_to_k_and_above_it_otherwise(oracle, seed):
    rng = random.Random(500 + seed)
    for _ in range(10):
        tn = rng.choice([30, 64, 65, 200, 500, 900])
        t = bytes(rng.choice(ACGT) for _ in range(tn))
        if rng.random() < 0.8:
            q = _mutate(rng, t, rng.choice([0.0, 0.03, 0.1, 0.3]))
        else:
            q = bytes(rng.choice(ACGT) for _ in range(max(1, int(tn * rng.uniform(0.5, 1.5)))))
        d = oracle.align(q, t, "NW", "distance", -1)["editDistance"]
        lo = abs(len(t) - len(q))
        for K in sorted({d, d + 1, d + 7, max(lo, d - 1), max(lo, d // 2), max(len(q), len(t))}):
            got = banded_nw(q, t, K)
            if d <= K:
                assert got == d, (len(q), len(t), d, K, got)
            else:
                assert got > K, (len(q), len(t), d, K, got)
        if lo > 0:
            assert banded_nw(q, t, lo - 1) is None


def test_config4_shape_on_the_21_lane_ring(oracle):
    """a 2.5 kb pair at ONT-like divergence with the threshold the 21-lane ring allows, and one just below its distance"""
    rng = random.Random(21)
    t = bytes(rng.choice(ACGT) for _ in range(2500))
    q = _mutate(rng, t, 0.12)
    d = oracle.align(q, t, "NW", "distance", -1)["editDistance"]
    assert banded_nw(q, t, ring_max_k(21)) == d
    assert banded_nw(q, t, d - 1) > d - 1
    assert ring_fits(len(q), len(t), ring_max_k(21), 21)


def _edge_hugging_pair(rng, p, core_len, nsub, upper):
    """the cheapest alignment runs `p` diagonals off the main one for the whole core: p symbols the other sequence cannot
    match on one end, p on the other -- distance 2 p + nsub, on the edge of the band of threshold 2 p"""
    core = bytes(rng.choice(ACGT) for _ in range(core_len))
    c2 = bytearray(core)
    for i in rng.sample(range(core_len), nsub):
        c2[i] = ord("A") if c2[i] != ord("A") else ord("C")
    a, b = b"T" * p + core, bytes(c2) + b"G" * p
    return (b, a) if upper else (a, b)                       # (query, target)


@pytest.mark.parametrize("G", [4, 8])
def test_lane_by_lane_ring_is_exact_with_the_band_filling_the_ring(oracle, G):
    """ring_lanes_nw restates the ring kernel's schedule lane by lane.  At K = ring_max_k(G) = 65 G - 64 every lane of the
    ring can be at work in one step: the block at the top of the band then has the band's BOTTOM block on the lane above
    it, and only stays exact because it stops listening once the block above has left the band.  Pairs whose cheapest path
    runs along the band's upper edge at distance K + 1 tell the difference: without the rule a substitution on the edge is
    forgiven and the scan reports K."""
    rng = random.Random(G)
    K = ring_max_k(G)
    assert K == 65 * G - 64
    wrong_without = 0
    for it in range(36):
        q, t = _edge_hugging_pair(rng, K // 2, 64 * G + 300, 1 if it % 3 else it % 4, upper=it % 6 != 5)
        d = oracle.align(q, t, "NW", "distance", -1)["editDistance"]
        got = ring_lanes_nw(q, t, K, G)
        assert (got == d) if d <= K else (got > K), (G, it, d, got)
        assert got == banded_nw(q, t, K)                              # lane by lane == column by column
        loose = ring_lanes_nw(q, t, K, G, upstream_rule=False)
        wrong_without += not ((loose == d) if d <= K else (loose > K))
        # one lane kept idle (the rule of rounds 1-4) needs no listening rule
        K0 = 64 * (G - 2)
        got0 = ring_lanes_nw(q, t, K0, G, upstream_rule=False)
        assert (got0 == d) if d <= K0 else (got0 > K0)
    assert wrong_without > 0


def test_lane_by_lane_ring_with_bands_of_a_few_diagonals():
    """K = 0, 1, 2: the block above leaves the band in the very step a block starts, or right after"""
    rng = random.Random(902)
    for G in (4, 8):
        for n in (64, 65, 200, 700):
            t = bytes(rng.choice(ACGT) for _ in range(n))
            for K in (0, 1, 2, 3):
                assert ring_lanes_nw(t, t, K, G) == 0
                q = bytearray(t); q[n // 2] = ord("A") if q[n // 2] != ord("A") else ord("C")
                got = ring_lanes_nw(bytes(q), t, K, G)
                assert got == 1 if K >= 1 else got > K
                if n > 70:
                    q2 = bytes(q[:n // 3] + q[n // 3 + 1:])              # one deletion as well: distance 2, one row short
                    got = ring_lanes_nw(q2, t, K, G)
                    assert (got == 2 if K >= 2 else (got is None or got > K)), (G, n, K, got)


@pytest.mark.parametrize("G", [4, 8])
def test_lane_by_lane_ring_inside_a_static_prefix_band(oracle, G):
    """SHW with a fixed k runs inside the static band [-K, K] (2 K + 1 diagonals: Batch::solveShwBanded); K = ring_max_k(G) / 2
    fills the ring.  The cheapest prefix alignment skips K target symbols and stays on the band's upper edge: the best
    bottom-row score the lanes see is the reference's SHW distance when that is <= K, and nothing <= K otherwise."""
    rng = random.Random(950 + G)
    K = ring_max_k(G) // 2
    for it in range(10):
        core = bytes(rng.choice(ACGT) for _ in range(64 * G + 250))
        c2 = bytearray(core)
        for j in rng.sample(range(len(core)), it % 3):
            c2[j] = ord("A") if c2[j] != ord("A") else ord("C")
        q, t = bytes(c2), b"T" * (K - (it % 4 == 3)) + core
        t = t[:len(q) + K]                                             # (columns past m + K cannot score <= K)
        want = oracle.align(q, t, "SHW", "distance", K)["editDistance"]
        rows = []
        ring_lanes_nw(q, t, K, G, geom=(-K, K), bottom_row=rows)
        best = min((v for _, v in rows), default=K + 1)
        assert (best == want) if want >= 0 else (best > K), (G, it, want, best)
        loose = []
        ring_lanes_nw(q, t, K, G, upstream_rule=False, geom=(-K, K), bottom_row=loose)
        assert min((v for _, v in loose), default=K + 1) <= best      # without the rule: never higher, sometimes too low


def test_lane_by_lane_ring_of_tall_lanes(oracle):
    """ring lanes of 128 rows (H = 2 in the kernel): the same rules with RH = 128, band limit (RH + 1) G - RH"""
    rng = random.Random(903)
    G, RH = 4, 128
    K = ring_max_k(G, RH)
    assert K == 129 * G - 128
    for it in range(10):
        q, t = _edge_hugging_pair(rng, K // 2, RH * G + 300, it % 3, upper=it % 4 != 3)
        d = oracle.align(q, t, "NW", "distance", -1)["editDistance"]
        assert ring_fits(len(q), len(t), K, G, RH)
        got = ring_lanes_nw(q, t, K, G, RH)
        assert (got == d) if d <= K else (got > K), (it, d, got)


def test_lane_by_lane_ring_on_random_pairs(oracle):
    rng = random.Random(901)
    for it in range(40):
        G = rng.choice([4, 4, 8])
        t = bytes(rng.choice(ACGT) for _ in range(rng.choice([300, 700, 1000])))
        q = _mutate(rng, t, rng.choice([0.05, 0.2, 0.35]))
        d = oracle.align(q, t, "NW", "distance", -1)["editDistance"]
        for K in (ring_max_k(G), d, d - 1):
            if K < abs(len(t) - len(q)) or K > ring_max_k(G):
                continue
            got = ring_lanes_nw(q, t, K, G)
            assert (got == d) if d <= K else (got > K), (G, len(q), len(t), d, K, got)


# ---------------------------------------------------------------- rings of 32-row words (ring32_kernels.hip)

@pytest.mark.parametrize("G", [4, 8, 16])
def test_ring32_keeps_tenants_apart_also_with_a_shared_geometry(G):
    """words of 32 rows: K <= 32 (G - 2) keeps the tenants of a ring lane apart, and so does ANY band with
    dmax - dmin <= 32 (G - 2) -- the geometry the units of a wave share (the extremes over the wave)"""
    rng = random.Random(100 + G)
    K = ring_max_k(G, 32)
    for _ in range(300):
        m = rng.randrange(1, 32 * 3 * G)
        T = max(1, m + rng.randrange(-min(K, m - 1), K + 1))
        assert ring_fits(m, T, K, G, 32), (m, T, K, G)
        lo, hi = band(m, T, K)
        # a neighbour in the wave with another T - m widens the band on either side
        lo2, hi2 = lo - rng.randrange(0, 12), hi + rng.randrange(0, 12)
        if hi2 - lo2 <= 32 * (G - 2):
            assert ring_fits(m, T, K, G, 32, (lo2, hi2)), (m, T, K, G, lo2, hi2)
            lv = [x for x in lives(m, T, K, 32, (lo2, hi2)) if x]
            for j in (0, T // 3, T // 2, T - 1):
                assert sum(1 for f, l in lv if f <= j <= l) <= G


def test_ring32_band_is_exact_up_to_k_also_when_widened(oracle):
    rng = random.Random(77)
    for it in range(60):
        m = rng.choice([1, 31, 32, 33, 64, 100, 257, 300, 700])
        q = bytes(rng.choice(ACGT) for _ in range(m))
        t = _mutate(rng, q, rng.choice([0.02, 0.1, 0.3]))
        if it % 6 == 0:
            t = bytes(rng.choice(ACGT) for _ in range(rng.randrange(1, m + 40)))
        want = oracle.align(q, t, "NW", "distance", -1)["editDistance"]
        for K in (want, max(want, abs(len(t) - m)) + 7, 128, 192):
            got = banded_nw(q, t, K, 32)
            if K < abs(len(t) - m):
                assert got is None
                continue
            assert (got == want) if want <= K else (got > K), (it, m, len(t), K, got, want)
            lo, hi = band(m, len(t), K)
            wide = banded_nw(q, t, K, 32, (lo - rng.randrange(0, 20), hi + rng.randrange(0, 20)))
            assert (wide == want) if want <= K else (wide > K), (it, "widened", K, wide, want)
