"""-m gpu: shared targets with more than four distinct bytes -- genomes with N runs, soft-masked lower case, IUPAC
codes (5 to 16 symbols) -- stay on the reads-per-lane kernels in HW mode (Peq rows [word][8 or 16 symbols][lane] in
LDS); reference semantics edlib.cpp:358-384 (buildPeq over any alphabet), 1417-1462.  Every field of every read is
compared with the oracle (native thread pool over the reference / the restatement) on the same bytes."""
import numpy as np
import pytest

from edlib_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _check(engine, reads, target, task, k=-1, eq=None):
    n, m = reads.shape
    b = engine.SharedBatch(reads, target, mode="HW", task=task, k=k, additionalEqualities=eq)
    try:
        st = b.run()
        got = b.results_flat()
    finally:
        b.close()
    ref = O.pool_align(reads.reshape(-1), np.arange(n + 1, dtype=np.int64) * m, target,
                       np.array([0, len(target)], dtype=np.int64), True, "HW", task, k, eq_pairs=eq)
    for f in ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "alnOff", "alignment"):
        assert np.array_equal(got[f], ref[f]), f
    if task != "distance":
        assert np.array_equal(got["starts"], ref["starts"])
    return st


@pytest.mark.parametrize("task", ["distance", "locations", "path"])
def test_genome_with_n_and_lower_case(engine, task):
    target = synth.masked_genome(41, 120_000)
    assert 5 <= len(set(target.tolist())) <= 16
    reads, _ = synth.window_reads(target, 1500 if task == "distance" else 400, 150, seed=42)
    st = _check(engine, reads, target, task)
    assert st["path"] & 1, "the batch left the reads-per-lane kernels"


def test_iupac_target_with_equalities(engine):
    target = synth.masked_genome(43, 60_000, iupac=True)
    assert 9 <= len(set(target.tolist())) <= 16
    reads, _ = synth.window_reads(target, 600, 100, seed=44, sub=0.02)
    eq = [("R", "A"), ("R", "G"), ("Y", "C"), ("Y", "T"), ("N", "A"), ("N", "C"), ("N", "G"), ("N", "T")]
    st = _check(engine, reads, target, "locations", eq=eq)
    assert st["path"] & 1


@pytest.mark.parametrize("m", [20, 33, 64, 97, 129, 200, 256])
def test_every_word_count(engine, m):
    target = synth.masked_genome(45 + m, 30_000, frac_n=0.02, frac_lower=0.2)
    reads, _ = synth.window_reads(target, 192, m, seed=46 + m, sub=0.03)
    _check(engine, reads, target, "distance")


def test_fixed_k_and_large_batch_through_both_passes(engine):
    """enough reads for the k-doubling probe (>= 16384 slots): pass 1 at a small threshold, leftovers in pass 2
    (banded kernel only: the plain kernel knows four symbols)"""
    target = synth.masked_genome(47, 40_000)
    reads, _ = synth.window_reads(target, 16500, 60, seed=48, sub=0.02)
    reads[::7] = synth.random_dna(49, len(reads[::7]) * 60).reshape(-1, 60)       # unrelated reads: pass 2
    _check(engine, reads, target, "distance")
    _check(engine, reads[:3000], target, "distance", k=2)


def test_many_unrelated_reads_take_the_full_height_kernel(engine):
    """>= 4096 leftovers whose band is the whole query: pass 2 runs on scan_reads_full_kernel (LDS rows, 8 / 16 symbols)"""
    for seed, kw in ((51, dict(frac_lower=0.0)), (52, dict())):          # 5 symbols (8 rows) and 9 symbols (16 rows)
        target = synth.masked_genome(seed, 30_000, **kw)
        reads, _ = synth.window_reads(target, 24000, 50, seed=seed + 10, sub=0.02)
        reads[::4] = synth.random_dna(seed + 20, len(reads[::4]) * 50).reshape(-1, 50)    # 6000 unrelated reads
        _check(engine, reads, target, "distance")


def test_seventeen_symbols_fall_back_to_pairs(engine):
    t = np.frombuffer(bytes(range(65, 82)) * 200, dtype=np.uint8)                  # 17 distinct bytes
    reads, _ = synth.window_reads(t, 64, 40, seed=50, sub=0.05)
    st = _check(engine, reads, t, "distance")
    assert st["path"] == 2
