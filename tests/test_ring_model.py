"""CPU: the band of scan_pairs_ring_kernel (tests/ring_model.py: the diagonal band of a threshold K cut into 64-row
blocks, "+1 per row" starts, +1 from an upstream outside the band, the score decode) is exact up to K and above K
otherwise, and ring_max_k(G) keeps consecutive tenants of a ring lane apart.  The model restates the kernel's rules,
not the kernel; the oracle is the judge."""
import random

import pytest

from ring_model import band, banded_nw, lives, ring_fits, ring_max_k

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
        # and never more than G blocks alive in one column
        lv = [x for x in lives(m, T, K) if x]
        for j in (0, T // 3, T // 2, T - 1):
            assert sum(1 for f, l in lv if f <= j <= l) <= G
    # the bound is not vacuous: far above it the tenants of a lane do collide
    assert not ring_fits(64 * 4 * G, 64 * 4 * G, 66 * G, G)


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
