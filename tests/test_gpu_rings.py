"""-m gpu: the banded lane-ring NW kernels (scan_pairs_ring_kernel, 4 / 16 / 64 lanes per unit) and
the banded column store + traceback behind EDLIB_TASK_PATH, against the compiled reference (or the
oracle restatement where it did not travel).

Directed at what the generic fuzz only hits by chance: distances right at the level thresholds
(the rings' band limits of rounds 1-4: 128, 896, 1920, 3968, and round 5's: 196, 976, 2016, 4031; bands that FILL the
ring: the last tests of this file), block counts right at the ring sizes (4, 16, 32, 64 blocks), mixed batches whose units
resolve at different levels, batches large enough to take the prefix divergence probe, fixed k, rings
sharing a wave with idle rings, and paths that run along the edge of the band."""
import os
import random

import numpy as np
import pytest

from edlib_amd import synth

pytestmark = pytest.mark.gpu
SEED_SHIFT = int(os.environ.get("EDLIB_FUZZ_SEED", "0"))     # soak runs: other inputs of the same shapes
FIELDS = ("status", "editDistance", "endLocations", "startLocations", "numLocations",
          "alignment", "alignmentLength", "alphabetLength")


def _impl(ref, oracle):
    return ref if ref is not None else oracle


def _check(engine, impl, qs, ts, mode, task, k, what):
    got = engine.align_pairs(qs, ts, mode=mode, task=task, k=k, raw=True)
    bad = []
    for i, (q, t, g) in enumerate(zip(qs, ts, got)):
        want = impl.align(q, t, mode, task, k)
        if want["status"] == 2:                       # oracle restatement: Hirschberg regime unsupported
            continue
        if any(g[f] != want[f] for f in FIELDS):
            bad.append((i, len(q), len(t), {f: (g[f], want[f]) for f in FIELDS if g[f] != want[f] and f != "alignment"}))
    assert not bad, (what, mode, task, k, bad[:3])


def _pair_with_edits(rng, n, edits, indel_frac=0.5):
    """A random n-base target and a query exactly `edits` scattered edit operations away (the true
    distance is <= edits and usually equal for sparse edits)."""
    t = synth.random_dna(rng.randrange(1 << 30), n).tobytes()
    q = bytearray(t)
    for _ in range(edits):
        p = rng.randrange(len(q))
        x = rng.random()
        if x < indel_frac / 2 and len(q) > 1:
            del q[p]
        elif x < indel_frac:
            q.insert(p, rng.choice(b"ACGT"))
        else:
            q[p] = rng.choice([c for c in b"ACGT" if c != q[p]])
    return bytes(q), t


def test_distances_at_level_thresholds(engine, ref, oracle):
    """Distances just below / at / above 128, 896, 1920 and 3968 (the band limits of the four ring sizes until round 4; fixed k
    also at today's limits)."""
    rng = random.Random(4101 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    qs, ts = [], []
    for n, edits in ((6000, 120), (6000, 127), (6000, 128), (6000, 129), (6000, 135),
                     (9000, 880), (9000, 896), (9000, 900), (9000, 930),
                     (20000, 1500), (20000, 1915), (20000, 1920), (20000, 1923), (20000, 1990),
                     (30000, 3900), (30000, 3968), (30000, 3975), (30000, 4100)):
        for _ in range(2):
            q, t = _pair_with_edits(rng, n, edits, indel_frac=0.2)
            qs.append(q); ts.append(t)
    _check(engine, impl, qs, ts, "NW", "distance", -1, "thresholds")
    for k in (127, 128, 129, 196, 896, 976, 1920, 2016, 3968, 4031, 5000):      # (round 5's ring limits: 196, 976, 2016, 4031)
        _check(engine, impl, qs[:28], ts[:28], "NW", "distance", k, "thresholds fixed k")


def test_block_counts_at_ring_sizes(engine, ref, oracle):
    """Queries of 1..5, 15..17 and 63..65 blocks, similar and unrelated targets, distance and path."""
    rng = random.Random(4102 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    for task in ("distance", "path"):
        qs, ts = [], []
        for m in (1, 63, 64, 65, 255, 256, 257, 320, 960, 1023, 1024, 1025, 1088, 2040, 2048, 2049, 2112, 4032, 4096, 4097, 4160):
            if task == "path" and m > 1100:
                continue
            for rate in (0.0, 0.03, 0.25):
                t = synth.random_dna(rng.randrange(1 << 30), m + rng.randrange(-m // 8, m // 8 + 1) if m > 8 else m)
                q, _ = synth.mutate(t, rng.randrange(1 << 30), rate, rate / 3, rate / 3)
                q = q[:m] if len(q) >= m else np.concatenate([q, synth.random_dna(rng.randrange(1 << 30), m - len(q))])
                qs.append(q.tobytes()); ts.append(t.tobytes())
            # unrelated pair: distance close to max(m, T), the band is the whole matrix
            qs.append(synth.random_dna(rng.randrange(1 << 30), m).tobytes())
            ts.append(synth.random_dna(rng.randrange(1 << 30), max(1, m // 2)).tobytes())
        _check(engine, impl, qs, ts, "NW", task, -1, "ring sizes")


def test_mixed_levels_and_probe(engine, ref, oracle):
    """A batch of > 256 units (takes the divergence probe) whose units need different levels; the
    sample checked against the reference covers every kind."""
    rng = random.Random(4103 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    qs, ts, kinds = [], [], []
    for i in range(600):
        kind = i % 6
        n = (200, 1000, 1000, 3000, 8000, 300)[kind]
        rate = (0.02, 0.01, 0.3, 0.05, 0.15, 0.6)[kind]
        t = synth.random_dna(rng.randrange(1 << 30), n + rng.randrange(0, 50))
        q, _ = synth.mutate(t, rng.randrange(1 << 30), rate, rate / 4, rate / 4)
        qs.append(q.tobytes() or b"A"); ts.append(t.tobytes()); kinds.append(kind)
    got = engine.align_pairs(qs, ts, mode="NW", task="distance", k=-1, raw=True)
    for i in list(range(0, 600, 7)) + list(range(590, 600)):
        want = impl.align(qs[i], ts[i], "NW", "distance", -1)
        assert all(got[i][f] == want[f] for f in FIELDS), (i, kinds[i], got[i], want)
    # the same batch with paths for the short kinds only (1 kb paths are the traceback branch)
    sel = [i for i in range(600) if kinds[i] in (0, 1, 2, 5)][:200]
    _check(engine, impl, [qs[i] for i in sel], [ts[i] for i in sel], "NW", "path", -1, "mixed paths")


def test_paths_along_the_band_edge(engine, ref, oracle):
    """All edits of one kind at one end: the optimal path hugs the lowest / highest diagonal of the
    band, where a block's left neighbour column is outside the band and the diagonal neighbour is the
    bottom cell of the block above (traceback_kernel, ring layout)."""
    rng = random.Random(4104 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    qs, ts = [], []
    for n in (64, 128, 129, 640, 1000):
        t = synth.random_dna(rng.randrange(1 << 30), n).tobytes()
        for d in (1, 2, 63, 64, 65, 100):
            if d >= n:
                continue
            ins = bytes(rng.choice(b"ACGT") for _ in range(d))
            qs += [ins + t, t + ins, t[d:], t[:-d], t[:n // 2] + ins + t[n // 2:], ins + t[:-d], t[d:] + ins]
            ts += [t] * 7
    for mode in ("NW", "HW", "SHW"):
        _check(engine, impl, qs, ts, mode, "path", -1, "band edge")


def test_low_complexity_paths(engine, ref, oracle):
    """Tie-rich inputs (1-3 letter alphabets): every tie must break as in the reference (up > left > diagonal)."""
    rng = random.Random(4105 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    qs, ts = [], []
    for _ in range(120):
        sigma = rng.choice([1, 2, 2, 3])
        m, n = rng.randrange(1, 700), rng.randrange(1, 700)
        qs.append(bytes(rng.choice(b"ACG"[:sigma]) for _ in range(m)))
        ts.append(bytes(rng.choice(b"ACG"[:sigma]) for _ in range(n)))
    for mode in ("NW", "HW", "SHW"):
        _check(engine, impl, qs, ts, mode, "path", -1, "ties")
    _check(engine, impl, qs, ts, "NW", "path", 150, "ties fixed k")


def test_wide_alphabet_rings(engine, ref, oracle):
    """More than 32 target symbols: the ring kernels gather Peq from the HBM pool."""
    rng = random.Random(4106 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    qs, ts = [], []
    for _ in range(40):
        n = rng.randrange(50, 2500)
        t = bytes(rng.randrange(33, 120) for _ in range(n))
        q = bytearray(t)
        for _ in range(rng.randrange(0, n // 5 + 1)):
            q[rng.randrange(len(q))] = rng.randrange(33, 120)
        qs.append(bytes(q)); ts.append(t)
    _check(engine, impl, qs, ts, "NW", "distance", -1, "wide alphabet")
    _check(engine, impl, qs[:20], ts[:20], "NW", "path", -1, "wide alphabet")


def test_semiglobal_units_on_rings(engine, ref, oracle):
    """SHW / HW pairs whose queries fit 4- or 16-lane rings (and some that do not, in the same batch):
    distances, end locations (also more than the 16 kept per unit), start locations, position -1."""
    rng = random.Random(4107 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    qs, ts = [], []
    for m in (1, 2, 63, 64, 65, 150, 255, 256, 257, 700, 1024, 1025, 1500):
        for _ in range(3):
            n = rng.choice([m // 2 + 1, m, 2 * m + 7, 5 * m + 100])
            t = synth.random_dna(rng.randrange(1 << 30), n).tobytes()
            if n > m and rng.random() < 0.7:
                a = rng.randrange(0, n - m + 1)
                q, _ = synth.mutate(np.frombuffer(t[a:a + m], dtype=np.uint8), rng.randrange(1 << 30), 0.05, 0.02, 0.02)
                q = q.tobytes() or b"A"
            else:
                q = synth.random_dna(rng.randrange(1 << 30), m).tobytes()
            qs.append(q); ts.append(t)
    # many end locations, and the empty-prefix location -1
    qs += [b"A" * 30, b"ACGT" * 10, b"AA", b"AC" * 100, b"G" * 200]
    ts += [b"A" * 500, b"T" * 50, b"B", b"AC" * 400, b"G" * 64]
    for mode in ("HW", "SHW"):
        for task in ("distance", "locations", "path"):
            _check(engine, impl, qs, ts, mode, task, -1, "semi-global rings")
        for k in (0, 5, 40):
            _check(engine, impl, qs, ts, mode, "locations", k, "semi-global rings fixed k")


def test_switches_do_not_change_results(engine):
    """EDLIB_AMD_NWBAND / NOPROBE / PEQFULL / BAND only choose kernels: every result field stays the same."""
    import os
    rng = random.Random(4108 + SEED_SHIFT)
    qs, ts = [], []
    for i in range(320):
        n = rng.choice([150, 700, 1000, 2500, 6000])
        rate = rng.choice([0.0, 0.02, 0.1, 0.3])
        t = synth.random_dna(rng.randrange(1 << 30), n)
        q, _ = synth.mutate(t, rng.randrange(1 << 30), rate, rate / 3, rate / 3)
        qs.append(q.tobytes() or b"A"); ts.append(t.tobytes())
    target = synth.random_dna(5, 30000)
    reads = synth.illumina_reads(target, 256, m=150, seed=9)["reads"]

    def everything():
        out = []
        for mode, task in (("NW", "distance"), ("NW", "path"), ("HW", "locations"), ("SHW", "path")):
            sel = slice(0, 320) if task != "path" else slice(0, 120)
            out.append(engine.align_pairs(qs[sel], ts[sel], mode=mode, task=task, k=-1, raw=True))
        out.append(engine.align_batch([r.tobytes() for r in reads], target.tobytes(), mode="HW", task="path", k=-1, raw=True))
        return out

    base = everything()
    for var, val in (("EDLIB_AMD_NWBAND", "0"), ("EDLIB_AMD_NOPROBE", "1"), ("EDLIB_AMD_PEQFULL", "0")):
        os.environ[var] = val
        try:
            got = everything()
        finally:
            del os.environ[var]
        assert got == base, var


@pytest.mark.parametrize("m", [300, 700, 1100, 1500, 5000])
def test_hw_long_queries_are_cut_into_target_segments(engine, oracle, m):
    """a handful of HW units with long targets: kernel W cuts every target into segments (each a unit with a warm-up
    of 2m-1 columns that records nothing) and merges them; scores, ALL end locations (several equal copies planted,
    some straddling cuts), start locations and paths must be what one uncut scan gives (reference edlib.cpp:550-704)"""
    import numpy as np
    from edlib_amd import synth
    T = 260_000
    t = synth.random_dna(600 + m, T)
    q0 = t[1000:1000 + m].copy()
    # plant exact copies of the same stretch at several places (cut positions are multiples of T / S: cover a few)
    for at in (40_000, 65_000 - m // 2, 129_990, 200_000, T - m):
        t[at:at + m] = q0
    q, _ = synth.mutate(q0, 601 + m, 0.02, 0.005, 0.005)
    qs = [q, q0, synth.random_dna(602 + m, m)]             # edited copy, exact copy, unrelated
    for task in ("distance", "locations", "path"):
        got = engine.align_batch(qs, t, mode="HW", task=task, raw=True)
        for i, qq in enumerate(qs):
            want = oracle.align(qq.tobytes(), t.tobytes(), "HW", task, -1)
            if want["status"] == 2:                         # Hirschberg regime not restated by the C oracle
                continue
            for f in ("editDistance", "endLocations", "startLocations", "numLocations", "alignment", "alphabetLength"):
                assert got[i][f] == want[f], (m, task, i, f)
    one = engine.align_raw(q.tobytes(), t.tobytes(), "HW", "locations", 40)
    assert one == oracle.align(q.tobytes(), t.tobytes(), "HW", "locations", 40)


def test_round2_ring_sizes_8_and_21(engine, ref, oracle):
    """The rings added in round 2 (8 lanes: K = 384 then, 456 now; 21 lanes: K = 1216 then, 1301 now; both carried by
    ds_bpermute): distances just below / at / above those limits, block counts right at 8 and 21 blocks, batches of >= 256 long units that
    take the prefix probe and start on the 21-lane ring with a tail that climbs to 32, fixed k, and paths whose
    storing scan lands on those rings (ring store row = block % 21)."""
    rng = random.Random(4110 + SEED_SHIFT)
    impl = _impl(ref, oracle)
    qs, ts = [], []
    for n, edits in ((7000, 370), (7000, 384), (7000, 385), (7000, 400), (12000, 1190), (12000, 1216), (12000, 1217), (12000, 1260)):
        for _ in range(3):
            q, t = _pair_with_edits(rng, n, edits, indel_frac=0.2)
            qs.append(q); ts.append(t)
    _check(engine, impl, qs, ts, "NW", "distance", -1, "8 / 21 thresholds")
    for k in (383, 384, 385, 456, 457, 1215, 1216, 1217, 1301, 1302):
        _check(engine, impl, qs, ts, "NW", "distance", k, "8 / 21 thresholds fixed k")
    # block counts at the ring sizes, similar / divergent / unrelated
    qs, ts = [], []
    for m in (449, 511, 512, 513, 1280, 1343, 1344, 1345, 1408):
        for rate in (0.01, 0.2):
            t = synth.random_dna(rng.randrange(1 << 30), m + rng.randrange(-20, 21))
            q, _ = synth.mutate(t, rng.randrange(1 << 30), rate, rate / 3, rate / 3)
            q = q[:m] if len(q) >= m else np.concatenate([q, synth.random_dna(rng.randrange(1 << 30), m - len(q))])
            qs.append(q.tobytes()); ts.append(t.tobytes())
        qs.append(synth.random_dna(rng.randrange(1 << 30), m).tobytes()); ts.append(synth.random_dna(rng.randrange(1 << 30), m // 3 + 1).tobytes())
    _check(engine, impl, qs, ts, "NW", "distance", -1, "8 / 21 blocks")
    _check(engine, impl, [q for q in qs if len(q) < 1100], [t for q, t in zip(qs, ts) if len(q) < 1100], "NW", "path", -1, "8 / 21 blocks paths")
    # paths of 3 kb pairs at distance ~250 (storing scan on 8-lane rings) and ~1000 (21-lane rings): Hirschberg regime
    # (3000 x 3000 needs 20 * 47 * 3000 bytes > 1 MiB), so the leaves of the levels use the new rings too
    qs, ts = [], []
    for edits in (250, 330, 1000, 1150):
        for _ in range(2):
            q, t = _pair_with_edits(rng, 3000 if edits < 500 else 9000, edits, indel_frac=0.3)
            qs.append(q); ts.append(t)
    if ref is not None:
        _check(engine, ref, qs, ts, "NW", "path", -1, "8 / 21 paths (Hirschberg)")
    # >= 256 long units: probe, first level 21 lanes, a tail above its limit (every tenth pair: distance ~1380 > 1301)
    qs, ts = [], []
    for i in range(300):
        t = synth.random_dna(rng.randrange(1 << 30), 10000)
        rate = 0.038 if i % 10 else 0.046
        q, _ = synth.mutate(t, rng.randrange(1 << 30), rate, rate, rate)
        qs.append(q.tobytes()); ts.append(t.tobytes())
    got = engine.align_pairs(qs, ts, mode="NW", task="distance", k=-1, raw=True)
    for i in range(0, 300, 3):
        want = impl.align(qs[i], ts[i], "NW", "distance", -1)
        assert all(got[i][f] == want[f] for f in FIELDS), (i, got[i], want)


def test_big_batch_of_like_pairs(engine, checker):
    """20,000 NW pairs of 1,100 bases (beyond the flat path's 16 blocks; the NW distance levels over a batch whose unit
    selection, descriptors and records are 20,000 long): a strided sample against the reference, every unit's shape fields,
    and two runs agree (records are recycled across runs)"""
    from edlib_amd import synth
    import numpy as np
    qs, ts = synth.mutated_pairs(20000, 1100, seed=77, sub=0.02, ins=0.01, dele=0.01)
    b = engine.PairBatch(list(qs), list(ts), mode="NW", task="distance", k=-1)
    try:
        b.run()
        first = b.results_flat()["editDistance"].copy()
        b.run()
        got = b.results_flat()
    finally:
        b.close()
    assert np.array_equal(first, got["editDistance"]) and np.all(got["status"] == 0) and np.all(got["numLocations"] == 1)
    assert np.all(got["ends"] == 1099)
    for i in range(0, 20000, 67):
        want = checker.align(qs[i].tobytes(), ts[i].tobytes(), "NW", "distance", -1)
        assert got["editDistance"][i] == want["editDistance"] and got["alphabetLength"][i] == want["alphabetLength"], i


# ---------------------------------------------------------------- bands that fill the ring (round 5: ring_max_k = 65 G - 64)

RING_CAPS = {4: 196, 8: 456, 16: 976, 21: 1301, 32: 2016, 64: 4031}      # pair_kernels.hpp: ring_max_k


def _edge_pair(rng, p, core_len, nsub, upper, junk=(b"T", b"G")):
    """the cheapest alignment runs p diagonals off the main one for the whole core (p symbols nothing matches on one end
    of the target, p on the other end of the query): distance 2 p + nsub, ON the edge of the band of threshold 2 p --
    the upper edge is where a block of the ring has the band's bottom block on the lane above it"""
    core = synth.random_dna(rng.randrange(1 << 30), core_len).tobytes()
    c2 = bytearray(core)
    for i in rng.sample(range(core_len), nsub):
        c2[i] = ord("A") if c2[i] != ord("A") else ord("C")
    a, b = junk[0] * p + core, bytes(c2) + junk[1] * p
    return (b, a) if upper else (a, b)                       # (query, target)


@pytest.mark.parametrize("G", [4, 8, 16, 21, 32, 64])
def test_bands_that_fill_the_ring(engine, checker, G):
    """Fixed k = ring_max_k(G) (the band of that threshold puts a block on EVERY lane of the ring; rounds 1-4 kept one lane
    idle) on pairs whose cheapest path runs along the edge of exactly that band, at distances k - 1 .. k + 3: every unit
    against the reference, then the same batch with k = -1 (the levels climb through the rings), and one ring below /
    above.  More than 512 units, so that the batch takes the levels (a handful goes straight to whole-wave rings)."""
    rng = random.Random(5100 + G + SEED_SHIFT)
    cap = RING_CAPS[G]
    qs, ts = [], []
    for i in range(540):
        p = cap // 2 - (1 if i % 9 == 8 else 0)
        q, t = _edge_pair(rng, p, max(64 * G + 300, 1100) + rng.randrange(0, 130), (0, 1, 1, 2, 3)[i % 5], upper=i % 4 != 3)
        qs.append(q); ts.append(t)
    for k in (cap, cap - 1, cap + 1, -1):
        got = engine.align_pairs(qs, ts, mode="NW", task="distance", k=k, raw=True)
        bad = []
        for i in range(0, 540, 1 if k == cap else 5):
            want = checker.align(qs[i], ts[i], "NW", "distance", k)
            if any(got[i][f] != want[f] for f in FIELDS):
                bad.append((i, got[i]["editDistance"], want["editDistance"]))
        assert not bad, (G, k, bad[:5])


@pytest.mark.parametrize("G", [4, 8, 16])
def test_paths_inside_bands_that_fill_the_ring(engine, checker, G):
    """the storing scan of a PATH runs inside the band of the unit's own distance (edlib.cpp:1196-1199): pairs whose
    distance is ring_max_k(G) or just below land on G-lane rings with every lane at work, and the walk reads their store"""
    rng = random.Random(5200 + G + SEED_SHIFT)
    cap = RING_CAPS[G]
    qs, ts = [], []
    for i in range(24):
        q, t = _edge_pair(rng, cap // 2 - i % 3, 1100 + rng.randrange(0, 200), 0 if i % 2 else 1, upper=i % 4 != 3)
        qs.append(q); ts.append(t)
    _check(engine, checker, qs, ts, "NW", "path", -1, "paths at the ring's limit")
    _check(engine, checker, qs, ts, "NW", "path", cap, "paths at the ring's limit, fixed k")


@pytest.mark.parametrize("G", [4, 8, 16])
def test_prefix_bands_that_fill_the_ring(engine, checker, G):
    """SHW inside the static band [-K, K] of a fixed k (2 K + 1 diagonals, solveSemiGlobalUnits): K = ring_max_k(G) / 2 fills
    the ring; the cheapest prefix alignment skips K target symbols first and stays on the band's upper edge"""
    rng = random.Random(5300 + G + SEED_SHIFT)
    K = RING_CAPS[G] // 2
    qs, ts = [], []
    for i in range(300):
        core = synth.random_dna(rng.randrange(1 << 30), max(64 * G + 200, 1100)).tobytes()
        c2 = bytearray(core)
        for j in rng.sample(range(len(core)), i % 3):
            c2[j] = ord("A") if c2[j] != ord("A") else ord("C")
        skip = K - (i % 7 == 6)
        qs.append(bytes(c2)); ts.append(b"T" * skip + core + synth.random_dna(rng.randrange(1 << 30), rng.randrange(0, 40)).tobytes())
    for k in (K, K + 1):
        got = engine.align_pairs(qs, ts, mode="SHW", task="locations", k=k, raw=True)
        bad = []
        for i in range(0, 300, 2):
            want = checker.align(qs[i], ts[i], "SHW", "locations", k)
            if any(got[i][f] != want[f] for f in FIELDS):
                bad.append((i, got[i]["editDistance"], want["editDistance"]))
        assert not bad, (G, k, bad[:5])


def test_level_that_takes_every_unit_and_its_tail(engine, checker):
    """9,000 NW pairs of ~1,200 bases, most at 3 % divergence, every 40th at 30 %: the batch takes the divergence probe, its
    first ring level takes EVERY unit (descriptors written on the device, the Peq of all units built next to the probe:
    Batch::prepareLevelAll / runLevelAll), the divergent ones climb on.  Also with a fixed k below the tail, and a second
    run of the same batch (the resident specs are reused)."""
    rng = random.Random(5400 + SEED_SHIFT)
    qs, ts = [], []
    for i in range(9000):
        t = synth.random_dna(rng.randrange(1 << 30), 1150 + rng.randrange(0, 120))
        rate = 0.1 if i % 40 == 7 else 0.01
        q, _ = synth.mutate(t, rng.randrange(1 << 30), rate, rate, rate)
        qs.append(q.tobytes()); ts.append(t.tobytes())
    for k in (-1, 150):
        b = engine.PairBatch(qs, ts, mode="NW", task="distance", k=k)
        try:
            b.run()
            first = b.results_flat()["editDistance"].copy()
            b.run()
            got = b.results_flat()
        finally:
            b.close()
        assert np.array_equal(first, got["editDistance"])
        for i in list(range(0, 9000, 61)) + list(range(7, 9000, 40))[:60]:
            want = checker.align(qs[i], ts[i], "NW", "distance", k)
            assert got["editDistance"][i] == want["editDistance"] and got["alphabetLength"][i] == want["alphabetLength"], (k, i)
        if k == -1:
            assert sum(1 for i in range(7, 9000, 40) if got["editDistance"][i] > 196) > 100      # the tail did leave the first level


@pytest.mark.parametrize("H,K,m", [(2, 968, 2300), (4, 1928, 4400)])
def test_prefix_bands_that_fill_rings_of_tall_lanes(engine, checker, H, K, m):
    """the same on 16-lane rings whose lanes hold H = 2 / 4 blocks (queries above 32 / 64 blocks): the static SHW band of
    2 K = (64 H + 1) 16 - 64 H diagonals is the most such a ring holds (ring_max_k(16, H) = 1936 / 3856)"""
    rng = random.Random(5500 + H + SEED_SHIFT)
    qs, ts = [], []
    for i in range(24):
        core = synth.random_dna(rng.randrange(1 << 30), m + rng.randrange(0, 100)).tobytes()
        c2 = bytearray(core)
        for j in rng.sample(range(len(core)), i % 3):
            c2[j] = ord("A") if c2[j] != ord("A") else ord("C")
        skip = K - (i % 5 == 4)
        qs.append(bytes(c2)); ts.append(b"T" * skip + core + synth.random_dna(rng.randrange(1 << 30), rng.randrange(0, 40)).tobytes())
    for k in (K, K + 1, K - 1):
        _check(engine, checker, qs, ts, "SHW", "locations", k, "tall ring lanes at their band limit")


def test_hw_band_takes_only_the_units_that_qualify(engine, checker):
    """a mixed HW batch: queries in windows barely longer than themselves (the static HW band) next to queries in long
    windows (every block of every column, cut into target segments): each kind takes its own path (Batch::solveSemiGlobal
    partitions the units), every field as the reference has it"""
    rng = random.Random(5900 + SEED_SHIFT)
    qs, ts = [], []
    for i in range(12):
        core = synth.random_dna(rng.randrange(1 << 30), 900 + rng.randrange(0, 300)).tobytes()
        q = bytearray(core)
        for j in rng.sample(range(len(q)), 6):
            q[j] = ord("A") if q[j] != ord("A") else ord("C")
        pad = 60 if i % 3 else 30000 + rng.randrange(0, 20000)           # narrow window / long window
        left = synth.random_dna(rng.randrange(1 << 30), pad // 2).tobytes()
        right = synth.random_dna(rng.randrange(1 << 30), pad - pad // 2).tobytes()
        qs.append(bytes(q)); ts.append(left + core + right)
    for task in ("distance", "locations", "path"):
        for k in (20, -1):
            _check(engine, checker, qs, ts, "HW", task, k, "mixed HW batch")
