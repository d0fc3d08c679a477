"""The lane-per-pair level of big NW distance batches (edlib_amd/csrc/lanepair*.hpp, Batch::prepareLaneLevel / runLaneLevel;
reference: myersCalcEditDistanceNW with the k-doubling of edlibAlign, edlib.cpp:197-217, 730-928) on the GPU, through the C ABI,
against the compiled reference.  The window logic itself is pinned on the CPU (tests/test_lanepair_model.py: the same header,
host-compiled); here: the pack kernel, the wave-uniform trims of 64 different lanes, units the level cannot hold, the levels
that follow it."""
import os
import random

import numpy as np
import pytest

from edlib_amd import synth

pytestmark = pytest.mark.gpu
SEED_SHIFT = int(os.environ.get("EDLIB_AMD_TEST_SEED", "0"))


def _batch(rng, n, lo, hi, rate_of, alphabet=b"ACGT"):
    qs, ts = [], []
    for i in range(n):
        t = synth.random_dna(rng.randrange(1 << 30), lo + rng.randrange(0, hi - lo + 1))
        r = rate_of(i)
        if r is None:
            q = synth.random_dna(rng.randrange(1 << 30), lo + rng.randrange(0, hi - lo + 1))
        else:
            q, _ = synth.mutate(t, rng.randrange(1 << 30), r, r, r)
        qb, tb = q.tobytes(), t.tobytes()
        if alphabet != b"ACGT":
            tr = bytes.maketrans(b"ACGT", alphabet)
            qb, tb = qb.translate(tr), tb.translate(tr)
        qs.append(qb); ts.append(tb)
    return qs, ts


def _run(engine, qs, ts, k, env=None):
    old = {}
    for key, v in (env or {}).items():
        old[key] = os.environ.get(key); os.environ[key] = v
    try:
        b = engine.PairBatch(qs, ts, mode="NW", task="distance", k=k)
        try:
            b.run()
            first = {f: b.results_flat()[f].copy() for f in ("editDistance", "alphabetLength", "numLocations")}
            b.run()                                               # the resident layout is reused
            second = b.results_flat()
            st = b.stats()
        finally:
            b.close()
    finally:
        for key, v in old.items():
            if v is None: os.environ.pop(key, None)
            else: os.environ[key] = v
    assert np.array_equal(first["editDistance"], second["editDistance"])
    return first, st


def _compare(checker, qs, ts, k, got, idx):
    for i in idx:
        want = checker.align(qs[i], ts[i], "NW", "distance", k)
        assert got["editDistance"][i] == want["editDistance"], (k, i, got["editDistance"][i], want["editDistance"])
        assert got["alphabetLength"][i] == want["alphabetLength"], (k, i)
        assert got["numLocations"][i] == want["numLocations"], (k, i)


def test_lane_level_takes_a_big_batch_and_the_rings_take_its_tail(engine, checker):
    """9,000 pairs of ~2 kb at 4 % per edit class, every 40th at 20 % (beyond the level's threshold: climbs the rings), every
    97th unrelated; k = -1 and a fixed k inside / below the bulk"""
    rng = random.Random(6100 + SEED_SHIFT)
    rate = lambda i: None if i % 97 == 5 else (0.2 if i % 40 == 7 else 0.04)
    qs, ts = _batch(rng, 9000, 1900, 2100, rate)
    idx = list(range(0, 9000, 53)) + list(range(7, 9000, 40))[:40] + list(range(5, 9000, 97))[:20]
    for k in (-1, 260, 180):
        got, st = _run(engine, qs, ts, k)
        _compare(checker, qs, ts, k, got, idx)
    # and the same answers from the rings alone
    ring, _ = _run(engine, qs, ts, -1, {"EDLIB_AMD_LANEPAIR": "0"})
    lane, _ = _run(engine, qs, ts, -1)
    assert np.array_equal(ring["editDistance"], lane["editDistance"]) and np.array_equal(ring["alphabetLength"], lane["alphabetLength"])


def test_static_band_and_trimmed_band_agree(engine, checker):
    rng = random.Random(6200 + SEED_SHIFT)
    qs, ts = _batch(rng, 8500, 2800, 3300, lambda i: 0.03 + 0.02 * (i % 3))
    a, _ = _run(engine, qs, ts, -1)
    b, _ = _run(engine, qs, ts, -1, {"EDLIB_AMD_LANEPAIR": "static"})
    assert np.array_equal(a["editDistance"], b["editDistance"])
    _compare(checker, qs, ts, -1, a, range(0, 8500, 211))


def test_three_symbol_targets_and_foreign_query_bytes(engine, checker):
    """targets over three symbols: a fourth query symbol takes the free code; targets over four with an N in some queries:
    those units are flagged by the pack kernel and stay on the rings"""
    rng = random.Random(6300 + SEED_SHIFT)
    qs, ts = _batch(rng, 8200, 1500, 1700, lambda i: 0.05)
    three = bytes.maketrans(b"T", b"A")
    ts3 = [t.translate(three) for t in ts]
    got, _ = _run(engine, qs, ts3, -1)
    _compare(checker, qs, ts3, -1, got, range(0, 8200, 173))
    qsn = list(qs)
    for i in range(3, 8200, 29):
        b = bytearray(qsn[i]); b[len(b) // 2] = ord("N"); b[7] = ord("N"); qsn[i] = bytes(b)
    got, _ = _run(engine, qsn, ts, -1)
    _compare(checker, qsn, ts, -1, got, list(range(3, 8200, 29))[:60] + list(range(0, 8200, 401)))


def test_config4_shape_sample(engine, checker):
    """BASELINE config 4's recipe at a tenth of its size: 10,000 pairs of 10 kb at 4 / 4 / 4 %"""
    qs, ts = synth.mutated_pairs(10000, 10000, seed=12349, sub=0.04, ins=0.04, dele=0.04)
    qs = [q.tobytes() for q in qs]; ts = [t.tobytes() for t in ts]
    got, st = _run(engine, qs, ts, -1)
    _compare(checker, qs, ts, -1, got, range(0, 10000, 199))


def test_batches_over_more_than_four_symbols_code_every_unit_from_its_own_target(engine, checker):
    """a genome-like batch: most pairs ACGT, some soft-masked (acgt: four symbols of their own, nine in the batch), some with
    a run of N in the target (five symbols: those units stay on the rings), an N in a query whose target has none"""
    rng = random.Random(6400 + SEED_SHIFT)
    qs, ts = _batch(rng, 8300, 1400, 1600, lambda i: 0.04)
    for i in range(0, 8300, 70):
        qs[i] = qs[i].lower(); ts[i] = ts[i].lower()
    for i in range(11, 8300, 50):
        t = bytearray(ts[i]); t[300:340] = b"N" * 40; ts[i] = bytes(t)
    for i in range(23, 8300, 90):
        q = bytearray(qs[i]); q[100] = ord("N"); qs[i] = bytes(q)
    got, _ = _run(engine, qs, ts, -1)
    idx = list(range(0, 8300, 70))[:40] + list(range(11, 8300, 50))[:40] + list(range(23, 8300, 90))[:30] + list(range(1, 8300, 157))
    _compare(checker, qs, ts, -1, got, idx)
    ring, _ = _run(engine, qs, ts, -1, {"EDLIB_AMD_LANEPAIR": "0"})
    assert np.array_equal(ring["editDistance"], got["editDistance"]) and np.array_equal(ring["alphabetLength"], got["alphabetLength"])


def test_thresholds_at_the_edges_of_the_windows(engine, checker):
    """fixed k right at what a window holds (32 W - 32: 480 / 736 / 1312 / 1504) and one beyond, on pairs whose distances
    straddle it: the level scans with the cap, answers above it climb the rings or are final (> k)"""
    rng = random.Random(6500 + SEED_SHIFT)
    for length, rate, ks in ((4000, 0.04, (479, 480, 481)), (6000, 0.04, (735, 736, 737)), (11000, 0.04, (1311, 1312, 1313)), (12500, 0.04, (1503, 1504, 1505))):
        qs, ts = _batch(rng, 8192, length - 100, length + 100, lambda i: rate * (0.8 + 0.4 * ((i * 7919) % 100) / 100.0))
        idx = list(range(0, 8192, 431))
        for k in ks:
            got, _ = _run(engine, qs, ts, k)
            _compare(checker, qs, ts, k, got, idx)


def test_tiny_caps_identical_pairs_and_long_units(engine, checker):
    rng = random.Random(6600 + SEED_SHIFT)
    # (a) identical pairs and pairs one edit apart under k = 0 / 1 / 2
    qs, ts = _batch(rng, 8200, 1200, 1300, lambda i: 0.0)
    for i in range(0, 8200, 3):
        q = bytearray(qs[i]); q[len(q) // 2] = ord("A") if q[len(q) // 2] != ord("A") else ord("C"); qs[i] = bytes(q)
    for i in range(1, 8200, 3):
        qs[i] = qs[i][:600] + qs[i][601:]
    for k in (0, 1, 2, -1):
        got, _ = _run(engine, qs, ts, k)
        _compare(checker, qs, ts, k, got, range(0, 8200, 97))
    # (b) 40 kb pairs at 2 %: 1,250 blocks of 32 columns, thresholds around 900
    qs, ts = _batch(rng, 8192, 39000, 41000, lambda i: 0.007)
    got, _ = _run(engine, qs, ts, -1)
    _compare(checker, qs, ts, -1, got, range(0, 8192, 683))
    # (c) lengths that differ by up to the like-lengths limit, and queries much shorter than their targets (no band within the threshold)
    qs, ts = _batch(rng, 8300, 2000, 2400, lambda i: 0.03)
    for i in range(5, 8300, 40):
        qs[i] = qs[i][: len(qs[i]) - 350]
    got, _ = _run(engine, qs, ts, -1)
    _compare(checker, qs, ts, -1, got, list(range(5, 8300, 40))[:60] + list(range(0, 8300, 211)))
