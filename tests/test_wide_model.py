"""CPU: the data path of scan_pairs_wide_kernel (tests/wide_model.py: strips of L blocks cut out of Ukkonen's diagonal band,
the 16-column granules of bottom-row deltas between them, +1 per column beyond the upper strip's life, the start score
taken over from the strip above, codes folded into block scores, the NW decode and the last-column dump) against the
oracle: exact up to K, above K otherwise, semi-global answers equal, dumped cells exact where they matter.  The model
restates the kernel's rules, not the kernel; the oracle is the judge."""
import random

import pytest

from wide_model import M64, WB, geom, popc, strip_range, wide_scan

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


def _rand(rng, n):
    return bytes(rng.choice(ACGT) for _ in range(n))


@pytest.mark.parametrize("L", [1, 2, 3])
def test_nw_exact_up_to_k(oracle, L):
    rng = random.Random(100 + L)
    for it in range(60):
        T = rng.randrange(1, 900)
        t = _rand(rng, T)
        q = _mutate(rng, t, rng.choice([0.02, 0.1, 0.3])) if rng.random() < 0.8 else _rand(rng, rng.randrange(1, 900))
        d = oracle.align(q, t, "NW", "distance", -1)["editDistance"]
        for K in sorted({max(d // 2, 0), max(d - 1, 0), d, d + 1, d + rng.randrange(1, 200), max(len(q), T)}):
            r = wide_scan(q, t, 0, K, L=L)
            if K < abs(T - len(q)):
                assert r is None
                continue
            score = r[0]
            if d <= K:
                assert score == d, (it, len(q), T, K, d, score)
            else:
                assert score is None or score > K, (it, len(q), T, K, d, score)


def _dp_column(q, t):
    """D[i][T] for i = 0..m (textbook NW)"""
    prev = list(range(len(q) + 1))
    for j, ch in enumerate(t):
        cur = [j + 1] + [0] * len(q)
        for i in range(1, len(q) + 1):
            cur[i] = min(prev[i] + 1, cur[i - 1] + 1, prev[i - 1] + (q[i - 1] != ch))
        prev = cur
    return prev


@pytest.mark.parametrize("L", [1, 2])
def test_hirschberg_half_dump(oracle, L):
    """a scan stopped at column tlen - 1 inside the band of the WHOLE piece (bandT): every dumped cell is an upper
    bound, and exact wherever the true value is <= K minus what the other half still has to pay (here: <= K)"""
    rng = random.Random(7 + L)
    for it in range(12):
        Tfull = rng.randrange(40, 420)
        tfull = _rand(rng, Tfull)
        q = _mutate(rng, tfull, 0.15)
        K = oracle.align(q, tfull, "NW", "distance", -1)["editDistance"]
        lw = Tfull // 2
        t = tfull[:lw]
        r = wide_scan(q, t, 0, K, L=L, bandT=Tfull)
        assert r is not None
        dump = r[4]
        col = _dp_column(q, t)
        seen_exact = 0
        for b, (P, Mv, sc) in dump.items():
            v = sc
            for bit in range(WB - 1, -1, -1):               # decode from the bottom of the word upwards
                i = WB * b + bit
                if i < len(q):
                    true = col[i + 1]
                    assert v >= true, (it, b, bit)
                    # cells on a path of total cost <= K: inside the band and exact
                    R = None
                    if true <= K:
                        dmin, dmax = geom(0, len(q), lw, Tfull, K)
                        if dmin <= (lw - 1) - i <= dmax:
                            R = true
                    if R is not None and true + abs((Tfull - lw) - (len(q) - i - 1)) <= K:
                        assert v == true, (it, b, bit, v, true)
                        seen_exact += 1
                v -= (P >> bit) & 1
                v += (Mv >> bit) & 1
        assert seen_exact > 0


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("L", [1, 2])
def test_semi_global_strips(oracle, mode, L):
    rng = random.Random(31 * mode + L)
    name = {1: "SHW", 2: "HW"}[mode]
    for it in range(40):
        T = rng.randrange(1, 700)
        t = _rand(rng, T)
        if rng.random() < 0.7:
            a = rng.randrange(0, T)
            q = _mutate(rng, t[a:a + rng.randrange(1, 400)], 0.1)
        else:
            q = _rand(rng, rng.randrange(1, 400))
        want = oracle.align(q, t, name, "distance", -1)
        k = len(q)                                           # the engine's "no threshold": every column <= m qualifies
        score, count, last, positions, _ = wide_scan(q, t, mode, k, L=L)
        ends = [e for e in want["endLocations"] if e >= 0]   # (the -1 position is the host's W rule, not the kernel's)
        if want["editDistance"] == len(q) and not ends:
            continue
        assert score == want["editDistance"], (it, name, len(q), T)
        assert positions == ends and count == len(ends) and last == ends[-1]


def test_strip_ranges_cover_the_band():
    rng = random.Random(5)
    for _ in range(300):
        m, T = rng.randrange(1, 5000), rng.randrange(1, 5000)
        K = abs(T - m) + rng.randrange(0, 3000)
        L = rng.choice([1, 2, 4, 64])
        dmin, dmax = geom(0, m, T, 0, K)
        nb = (m + WB - 1) // WB
        ns = (nb + L - 1) // L
        prev = None
        for s in range(ns):
            c0, c1 = strip_range(s, L, T, dmin, dmax)
            if c0 > c1:
                # nothing below has a column either
                assert all(strip_range(x, L, T, dmin, dmax)[0] > strip_range(x, L, T, dmin, dmax)[1] for x in range(s, ns))
                break
            # every in-band cell of the strip's rows lies in its column range
            for i in (WB * L * s, min(m, WB * L * (s + 1)) - 1):
                lo, hi = max(0, i + dmin), min(T - 1, i + dmax)
                if lo <= hi:
                    assert c0 <= lo and hi <= c1
            if prev is not None and c0 > 0:
                assert prev[0] <= c0 - 1 <= prev[1]          # the strip above passes column c0 - 1 (start score)
                assert prev[1] >= c0                          # and is still alive at c0 (first hin is a real delta)
            prev = (c0, c1)


@pytest.mark.parametrize("L", [1, 2, 3])
def test_shw_inside_the_band_of_a_threshold(oracle, L):
    """SHW with threshold K only needs the diagonals [-K, K] and the first m + K columns: same answer as the oracle with
    k = K (score, every end position), "none" when the distance is above K"""
    rng = random.Random(77 + L)
    for it in range(60):
        T = rng.randrange(1, 700)
        t = _rand(rng, T)
        q = _mutate(rng, t[:rng.randrange(1, T + 1)], rng.choice([0.02, 0.1, 0.3])) if rng.random() < 0.8 else _rand(rng, rng.randrange(1, 400))
        m = len(q)
        d = oracle.align(q, t, "SHW", "distance", -1)["editDistance"]
        for K in sorted({max(1, d // 2), max(1, d - 1), max(1, d), d + 1, d + 40}):
            if K >= m or T < m - K:
                continue                                       # (the host scans such units unbanded / answers "none" itself)
            want = oracle.align(q, t, "SHW", "distance", K)
            tcut = t[:m + K]
            score, count, last, positions, _ = wide_scan(q, tcut, 1, K, L=L, bandT=-1)
            ends = [e for e in (want["endLocations"] or []) if e >= 0]
            if want["editDistance"] < 0 or not ends:
                assert score == -1 or (want["editDistance"] == m), (it, m, T, K, d, score)
                continue
            assert score == want["editDistance"] and positions == ends, (it, m, T, K, d, score, positions, ends)


@pytest.mark.parametrize("L", [1, 2, 64])
def test_hw_inside_the_band_of_a_threshold(oracle, L):
    """HW with threshold K only needs the diagonals [-K, (T - m) + 2 K] (an alignment within K starts at a column in
    [0, T - m + K] and stays within K diagonals of it): same answer as the oracle with k = K (score, every end
    position), "none" when nothing is within K"""
    rng = random.Random(177 + L)
    for it in range(60):
        m = rng.randrange(1, 500)
        slack = rng.randrange(0, 120)
        t = _rand(rng, m + slack)
        lead = rng.randrange(0, slack + 1)
        q = _mutate(rng, t[lead:lead + m], rng.choice([0.02, 0.1, 0.3])) if rng.random() < 0.8 else _rand(rng, m)
        m, T = len(q), len(t)
        d = oracle.align(q, t, "HW", "distance", -1)["editDistance"]
        for K in sorted({max(1, d // 2), max(1, d - 1), max(1, d), d + 1, d + 25}):
            if K >= m or T < m - K:
                continue
            want = oracle.align(q, t, "HW", "distance", K)
            score, count, last, positions, _ = wide_scan(q, t, 2, K, L=L, bandT=-1)
            ends = [e for e in (want["endLocations"] or []) if e >= 0]
            if want["editDistance"] < 0 or not ends:
                assert score == -1 or (want["editDistance"] == m), (it, m, T, K, d, score)
                continue
            assert score == want["editDistance"] and positions == ends, (it, m, T, K, d, score, positions, ends)
