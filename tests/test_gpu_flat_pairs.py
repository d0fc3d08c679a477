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
@pytest.mark.parametrize("maxq", [256, 1024])
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
