"""CPU: the wave-level band rules of scan_reads_banded_kernel (tests/band_model.py: growth at S <= k + c - 1, the
(A + B - span) / 2 shrink bounds, the height ladders of the 12 / 16 / 24 / 32-word groups, the per-lane bottom row of a
padded group, k-doubling) give the reference's HW distance and end columns.  The model restates the kernel's rules,
not the kernel: it guards the arithmetic of the rules against the oracle where no GPU is needed."""
import random

import pytest

from band_model import Wave, group_words, height_down, height_ok, height_up

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
    return bytes(out)


def _reads(rng, target, lengths, unrelated_every=4):
    out = []
    for i, m in enumerate(lengths):
        if i % unrelated_every == unrelated_every - 1:
            out.append(bytes(rng.choice(ACGT) for _ in range(m)))
            continue
        a = rng.randrange(0, len(target) - m - 40)
        r = _mutate(rng, target[a:a + m + 40], rng.choice([0.0, 0.02, 0.05, 0.12]))[:m]
        out.append(r + bytes(rng.choice(ACGT) for _ in range(m - len(r))))
    return out


def test_ladders():
    for nwd in (1, 2, 5, 8):
        assert all(height_ok(nwd, h) for h in range(1, nwd + 1))
    assert [h for h in range(1, 33) if height_ok(32, h)] == [1, 2, 3, 4, 6, 8, 12, 16, 24, 32]
    assert [h for h in range(1, 13) if height_ok(12, h)] == [1, 2, 3, 4, 6, 8, 12]
    for nwd in (12, 16, 24, 32):
        h, seen = 1, [1]
        while h < nwd:
            h = height_up(nwd, h); seen.append(h)
        assert seen == [x for x in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32) if x <= nwd]
        assert [height_down(nwd, x) for x in seen[1:]] == seen[:-1]
        # the bottom row of the group's shortest read (word nwd - 4, or nwd - 8 above 16 words) is outside every
        # height below the full one
        assert height_down(nwd, nwd) <= nwd - (8 if nwd > 16 else 4)
    assert [group_words(m) for m in (1, 32, 33, 256, 257, 384, 385, 512, 513, 768, 769, 1024)] == \
           [1, 1, 2, 8, 12, 12, 16, 16, 24, 24, 32, 32]


@pytest.mark.parametrize("nwd,lo,hi", [(1, 8, 32), (2, 33, 64), (3, 65, 96), (5, 129, 160), (8, 225, 256),
                                       (12, 257, 384), (16, 385, 512), (24, 513, 768), (32, 769, 1024)])
def test_band_rules_give_the_reference_answer(oracle, nwd, lo, hi):
    rng = random.Random(1000 + nwd)
    T = 2600 if nwd <= 8 else 3600
    target = bytes(rng.choice(ACGT) for _ in range(T))
    heights = set()
    for wave in range(3 if nwd <= 8 else 2):
        lengths = [lo, hi] + [rng.randrange(lo, hi + 1) for _ in range(4)]
        reads = _reads(rng, target, lengths)
        assert all(group_words(len(r)) == nwd for r in reads)
        w = Wave(reads, nwd, ACGT)
        log = []
        w.scan(target, [len(r) for r in reads], kcap=8, log=log)          # pass 1 alone: what the ladder does
        heights.update(log)
        got = w.solve(target)
        for r, (best, cols) in zip(reads, got):
            want = oracle.align(r, target, "HW", "locations", -1)
            assert best == want["editDistance"], (nwd, len(r))
            assert cols == [e for e in want["endLocations"] if e >= 0], (nwd, len(r))
    assert 1 in heights or nwd == 1                                       # the band does come down to one word ...
    if nwd > 8:
        assert nwd in heights and len(heights) >= 4                       # ... and climbs the ladder at the matches


def test_fixed_thresholds_and_repeats(oracle):
    """a tandem repeat keeps the band tall; thresholds below the distance leave a lane without a hit"""
    rng = random.Random(7)
    unit = bytes(rng.choice(ACGT) for _ in range(97))
    target = unit * 30
    reads = [(unit * 5)[11:11 + m] for m in (270, 300, 340, 384)]
    reads = [bytes(bytearray(r[:50]) + bytearray(b"T" if r[50:51] != b"T" else b"A") + bytearray(r[51:])) for r in reads]
    w = Wave(reads, 12, ACGT)
    for k in (0, 1, 5):
        got = w.scan(target, [k] * len(reads))
        for r, (best, cols) in zip(reads, got):
            want = oracle.align(r, target, "HW", "locations", k)
            if want["editDistance"] < 0:
                assert not cols
            else:
                assert best == want["editDistance"] and cols == [e for e in want["endLocations"] if e >= 0]


@pytest.mark.parametrize("nwd,lo,hi", [(1, 20, 32), (2, 33, 64), (3, 65, 96), (4, 97, 128), (5, 129, 160), (6, 161, 192),
                                       (7, 193, 224), (8, 225, 256), (12, 257, 384), (16, 385, 512), (24, 513, 768),
                                       (32, 769, 1024)])
def test_rules_are_tight_at_the_true_distance(oracle, nwd, lo, hi):
    """one lane per wave (no neighbour keeps the band open) and k = the read's own distance (+0 / +1): the match is
    found only if the band grows in time and never sheds a word that still holds a cell <= k.  (Moving the growth
    thresholds by 2-3 or the shrink bounds by 6-8 makes this test fail: it is what pins the constants of band_quad.)"""
    rng = random.Random(2000 + nwd)
    target = bytes(rng.choice(ACGT) for _ in range(3000))
    reads = _reads(rng, target, [rng.randrange(lo, hi + 1) for _ in range(20)], unrelated_every=1000)
    for i, r in enumerate(reads):
        want = oracle.align(r, target, "HW", "locations", -1)
        k = want["editDistance"] + (i & 1)
        (best, cols), = Wave([r], nwd, ACGT).scan(target, [k])
        assert best == want["editDistance"], (nwd, len(r), k)
        assert cols == [e for e in want["endLocations"] if e >= 0], (nwd, len(r), k)
