"""-m gpu: batches of >= 1024 short independent pairs asked for distances take the flat pair path (engine.hip: descriptors
built once and resident, one ring scan per run, results left in HBM until results()); a batch with an end-location list
longer than 16 gets the exact second pass for that unit.  Every field against the reference (native pool), every mode."""
import numpy as np
import pytest

from edlib_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _pairs(n, seed, maxq, repeats=False):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for i in range(n):
        m = int(rng.choice([1, 5, 63, 64, 65, 100, 150, 255, 256, 257, maxq]))
        T = int(rng.choice([1, m, m + 37, 2 * m + 11, 400]))
        t = _ACGT[rng.integers(0, 4, T)]
        if T > m and rng.random() < 0.7:
            a = int(rng.integers(0, T - m + 1))
            q = t[a:a + m].copy()
            hit = rng.random(m) < 0.04
            q[hit] = _ACGT[rng.integers(0, 4, int(hit.sum()))]
        else:
            q = _ACGT[rng.integers(0, 4, m)]
        qs.append(q); ts.append(t)
    if repeats:                                           # more than 16 end locations: that unit gets the exact second pass
        qs[7] = np.frombuffer(b"ACAC", dtype=np.uint8); ts[7] = np.tile(np.frombuffer(b"AC", dtype=np.uint8), 60)
    return qs, ts


def _check(engine, qs, ts, mode, k=-1):
    b = engine.PairBatch(qs, ts, mode=mode, task="distance", k=k)
    try:
        b.run(); st = b.run()
        got = b.results_flat()
        again = b.results_flat()                          # (collected once, handed out twice)
    finally:
        b.close()
    qoff = np.zeros(len(qs) + 1, dtype=np.int64); qoff[1:] = np.cumsum([len(q) for q in qs])
    toff = np.zeros(len(ts) + 1, dtype=np.int64); toff[1:] = np.cumsum([len(t) for t in ts])
    ref = O.pool_align(np.concatenate(qs), qoff, np.concatenate(ts), toff, False, mode, "distance", k)
    for f in ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends"):
        assert np.array_equal(got[f], ref[f]), (mode, k, f)
        assert np.array_equal(again[f], ref[f]), (mode, k, f, "second collection")
    return st


@pytest.mark.parametrize("mode", ["NW", "SHW", "HW"])
@pytest.mark.parametrize("maxq", [256, 512, 1024])          # 4-, 8- and 16-lane rings
def test_flat_pairs_every_mode(engine, mode, maxq):
    qs, ts = _pairs(3000, 11 + maxq, maxq)
    _check(engine, qs, ts, mode)
    _check(engine, qs[:1500], ts[:1500], mode, k=6)


def test_overflowing_lists_get_the_exact_second_pass(engine):
    qs, ts = _pairs(2000, 5, 200, repeats=True)
    for mode in ("HW", "SHW"):
        _check(engine, qs, ts, mode)


def test_wide_alphabet_pairs(engine):
    rng = np.random.default_rng(3)
    ts = [rng.integers(0, 200, 300).astype(np.uint8) for _ in range(1500)]
    qs = [t[50:200].copy() for t in ts]
    for q in qs:
        q[::13] = 201
    _check(engine, qs, ts, "HW")
    _check(engine, qs, ts, "NW")


# ---------------------------------------------------------------- start locations and paths stay flat too (round 4)

_ALL = ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "starts", "alnOff", "alignment")


def _check_task(engine, qs, ts, mode, task, k=-1, what=""):
    b = engine.PairBatch(qs, ts, mode=mode, task=task, k=k)
    try:
        b.run(); b.run()
        got = b.results_flat()
        again = b.results_flat()
    finally:
        b.close()
    qoff = np.zeros(len(qs) + 1, dtype=np.int64); qoff[1:] = np.cumsum([len(q) for q in qs])
    toff = np.zeros(len(ts) + 1, dtype=np.int64); toff[1:] = np.cumsum([len(t) for t in ts])
    ref = O.pool_align(np.concatenate(qs), qoff, np.concatenate(ts), toff, False, mode, task, k)
    z = np.zeros(0, dtype=np.int32)
    for res_, tag in ((got, "first collection"), (again, "second collection")):
        for f in _ALL:
            a = res_[f] if res_[f] is not None else z
            r = ref[f] if ref[f] is not None else z
            assert np.array_equal(a, r), (mode, task, k, f, what, tag)


def _window_pairs(n, seed, m, win, sub=0.03, indel=0.01):
    """reads of m bases inside their own window (the verification step of a seed-and-extend mapper)"""
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for i in range(n):
        t = _ACGT[rng.integers(0, 4, win)]
        a = int(rng.integers(0, win - m + 1))
        q, _ = synth.mutate(t[a:a + m], int(rng.integers(1 << 30)), sub, indel, indel)
        if i % 97 == 0:
            q = _ACGT[rng.integers(0, 4, m)]                      # unrelated
        qs.append(np.ascontiguousarray(q)); ts.append(t)
    return qs, ts


@pytest.mark.parametrize("mode", ["HW", "SHW", "NW"])
@pytest.mark.parametrize("task", ["locations", "path"])
def test_flat_locations_and_paths(engine, mode, task):
    qs, ts = _window_pairs(2500, 21, 150, 400)
    _check_task(engine, qs, ts, mode, task, what="150 in 400")
    _check_task(engine, qs[:1200], ts[:1200], mode, task, k=4, what="fixed k")
    qs, ts = _pairs(2000, 31, 256)                                 # every length edge, T < m, T = 1
    _check_task(engine, qs, ts, mode, task, what="mixed lengths")


def test_flat_nw_paths_of_one_kb_pairs(engine):
    """BASELINE config 5's shape (1 kb NW pairs, PATH): the distance scan inside the first band level is the storing scan"""
    qs, ts = synth.mutated_pairs(1500, 1000, seed=44, sub=0.03, ins=0.01, dele=0.01)
    _check_task(engine, list(qs), list(ts), "NW", "path", what="config 5 shape")
    # a divergent unit among them fails the level: the run falls back to the general path, same answers
    qs = list(qs); ts = list(ts)
    qs[17] = synth.random_dna(5, 1000)
    _check_task(engine, qs, ts, "NW", "path", what="one divergent unit")


def test_flat_batches_fall_back_when_lists_overflow(engine):
    qs, ts = _pairs(2000, 5, 200, repeats=True)                    # unit 7: 59 end locations
    for mode in ("HW", "SHW"):
        for task in ("locations", "path"):
            _check_task(engine, qs, ts, mode, task, what="overflowing list")


def test_flat_paths_with_the_empty_prefix_first(engine):
    """a query that matches nothing: distance m, first location -1 (when 64 does not divide m), the path is m inserts"""
    rng = np.random.default_rng(9)
    qs = [np.frombuffer(b"A" * int(rng.choice([10, 63, 64, 65, 100])), dtype=np.uint8) for _ in range(1200)]
    ts = [np.frombuffer(b"C" * int(rng.choice([1, 30, 200])), dtype=np.uint8) for _ in range(1200)]
    for mode in ("HW", "SHW"):
        _check_task(engine, qs, ts, mode, "path", what="all mismatching")


# ---------------------------------------------------------------- rings of 32-row words, device-made views and CIGARs (round 5)

def _nw_pairs(n, seed, lengths, rates, gaps=False):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for i in range(n):
        T = int(rng.choice(lengths))
        t = _ACGT[rng.integers(0, 4, T)]
        sub, indel = rates[i % len(rates)]
        q, _ = synth.mutate(t, int(rng.integers(1 << 30)), sub, indel, indel)
        q = np.ascontiguousarray(q)
        if gaps and i % 5 == 0 and len(q) > 120:           # a long gap: runs of up / left moves across 32-row words
            a = int(rng.integers(0, len(q) - 100))
            g = int(rng.choice([33, 40, 70]))
            q = np.concatenate([q[:a], q[a + g:]]) if i % 10 == 0 else np.concatenate([q[:a], _ACGT[rng.integers(0, 4, g)], q[a:]])
        if len(q) > 1024:
            q = q[:1024]
        if len(q) == 0:
            q = _ACGT[:1].copy()
        qs.append(np.ascontiguousarray(q)); ts.append(t)
    return qs, ts


@pytest.mark.parametrize("lengths,rates", [
    ((1000,), ((0.03, 0.01),)),                                       # BASELINE config 5's shape: K = 128 on 8 words of 32 rows
    ((257, 300, 511, 512, 513, 777, 1023), ((0.01, 0.005), (0.05, 0.02))),   # every word edge, mixed T - m inside a wave
    ((1, 2, 31, 32, 33, 64, 100, 255, 256), ((0.1, 0.05), (0.0, 0.0))),      # queries that sit whole on their ring
])
def test_ring32_paths(engine, lengths, rates):
    qs, ts = _nw_pairs(2100, 7 + len(lengths), lengths, rates, gaps=True)
    _check_task(engine, qs, ts, "NW", "path", what="ring32 %r" % (lengths,))
    _check_task(engine, qs[:1100], ts[:1100], "NW", "path", k=60, what="ring32 fixed k")


def test_views_and_records_agree(engine):
    """edlibAmdBatchResultsView (device-made for a flat batch) against the per-unit records of edlibAmdBatchResults"""
    for mode, task in (("HW", "path"), ("NW", "path"), ("SHW", "locations"), ("HW", "distance")):
        qs, ts = _window_pairs(1300, 5, 150, 400)
        b = engine.PairBatch(qs, ts, mode=mode, task=task, k=-1)
        try:
            b.run()
            v = b.results_flat(copy=False)
            rec = b.results()
        finally:
            pass
        try:
            for u, r in enumerate(rec):
                lo, hi = int(v["locOff"][u]), int(v["locOff"][u + 1])
                assert r["editDistance"] == v["editDistance"][u] and r["numLocations"] == hi - lo and r["alphabetLength"] == v["alphabetLength"][u]
                assert (r["endLocations"] or []) == list(v["ends"][lo:hi])
                if r["startLocations"] is not None:
                    assert r["startLocations"] == list(v["starts"][lo:hi])
                if r["alignment"] is not None:
                    assert r["alignment"] == v["alignment"][int(v["alnOff"][u]):int(v["alnOff"][u + 1])].tobytes()
        finally:
            b.close()


@pytest.mark.parametrize("n", [1500, 40])       # a flat batch (CIGARs made on the device) and a general one (on the host)
def test_batch_cigars_equal_edlibAlignmentToCigar(engine, n):
    qs, ts = _nw_pairs(n, 91, (5, 64, 150, 700, 1000), ((0.03, 0.01), (0.2, 0.1)), gaps=True)
    qs[3] = np.frombuffer(b"A" * 300, dtype=np.uint8); ts[3] = np.frombuffer(b"A" * 300, dtype=np.uint8)     # one run of 300
    b = engine.PairBatch(qs, ts, mode="NW", task="path", k=-1)
    try:
        b.run()
        flat = b.results_flat()
        for extended in (True, False):
            got = b.cigar_list(extended)
            again = b.cigar_list(extended)
            assert got == again
            for u in range(n):
                ops = flat["alignment"][int(flat["alnOff"][u]):int(flat["alnOff"][u + 1])].tobytes()
                assert got[u] == engine.cigar_from_alignment(ops, extended), (u, extended)
        assert b.cigar_list(True)[3] == "300=" and b.cigar_list(False)[3] == "300M"
        # with a threshold some units have no alignment: their CIGAR is the empty string
        b2 = engine.PairBatch(qs, ts, mode="NW", task="path", k=3)
        try:
            b2.run()
            f2 = b2.results_flat()
            c2 = b2.cigar_list(True)
            for u in range(n):
                if f2["editDistance"][u] < 0:
                    assert c2[u] == ""
                else:
                    assert c2[u] == engine.cigar_from_alignment(f2["alignment"][int(f2["alnOff"][u]):int(f2["alnOff"][u + 1])].tobytes(), True)
        finally:
            b2.close()
    finally:
        b.close()
