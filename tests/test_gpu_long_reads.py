"""-m gpu: HW reads of 257..1024 bases against a shared target stay on the reads-per-lane kernels (groups of 12, 16, 24
and 32 words: the bottom row of a lane sits in any of the group's last four / eight words, band heights step 1, 2,
3, 4, 6, 8, 12, 16, 24, 32; above four target symbols the limit is 512 bases).  Reference semantics: edlib.cpp:550-704 (the semi-global scan and its band), 197-217 (k-doubling).  Every field
of every read is compared with the oracle (native thread pool over the reference / the restatement)."""
import numpy as np
import pytest

from edlib_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _reads(target, lengths, seed, unrelated_every=0, max_err=0.12):
    """one read per entry of `lengths`: a window of the target with substitutions, insertions and deletions at a
    per-read rate in [0, max_err), cut / extended to exactly that length; every `unrelated_every`-th read is random"""
    rng = np.random.default_rng(seed)
    out = []
    for i, m in enumerate(lengths):
        if unrelated_every and i % unrelated_every == unrelated_every - 1:
            out.append(_ACGT[rng.integers(0, 4, m)])
            continue
        s = int(rng.integers(0, len(target) - m - 64))
        w = target[s:s + m + 64].copy()
        rate = rng.random() * max_err
        nmut = int(rate * m)
        for _ in range(nmut):
            p = int(rng.integers(0, m))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                w[p] = _ACGT[rng.integers(0, 4)]
            elif kind == 1:
                w = np.delete(w, p)
            else:
                w = np.insert(w, p, _ACGT[rng.integers(0, 4)])
        w = w[:m]
        if len(w) < m:
            w = np.concatenate([w, _ACGT[rng.integers(0, 4, m - len(w))]])
        out.append(np.ascontiguousarray(w))
    return out


def _check(engine, reads, target, task, k=-1, eq=None):
    b = engine.SharedBatch(reads, target, mode="HW", task=task, k=k, additionalEqualities=eq)
    try:
        st = b.run()
        got = b.results_flat()
    finally:
        b.close()
    qoff = np.zeros(len(reads) + 1, dtype=np.int64)
    qoff[1:] = np.cumsum([len(r) for r in reads])
    ref = O.pool_align(np.concatenate(reads), qoff, target, np.array([0, len(target)], dtype=np.int64), True, "HW",
                       task, k, eq_pairs=eq)
    for f in ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "alnOff", "alignment"):
        assert np.array_equal(got[f], ref[f]), f
    if task != "distance":
        assert np.array_equal(got["starts"], ref["starts"])
    return st


_EDGES = [257, 258, 287, 288, 289, 319, 320, 321, 352, 353, 383, 384, 385, 386, 415, 416, 417, 448, 449, 479, 480, 481,
          511, 512]


@pytest.mark.parametrize("task", ["distance", "locations", "path"])
def test_mixed_lengths_257_to_512(engine, task):
    target = synth.random_dna(61, 80_000)
    rng = np.random.default_rng(62)
    n = 900 if task == "distance" else 300
    lengths = _EDGES + [int(x) for x in rng.integers(257, 513, n)]
    reads = _reads(target, lengths, 63, unrelated_every=9)
    st = _check(engine, reads, target, task)
    if task == "distance":
        assert st["path"] == 1, "a read of 257..512 bases left the reads-per-lane kernels"
    else:
        assert st["path"] & 1


@pytest.mark.parametrize("k", [0, 3, 25, 70, 600])
def test_fixed_k(engine, k):
    target = synth.random_dna(64, 50_000)
    rng = np.random.default_rng(65 + k)
    lengths = _EDGES + [int(x) for x in rng.integers(257, 513, 300)]
    reads = _reads(target, lengths, 66 + k, unrelated_every=7, max_err=0.2)
    _check(engine, reads, target, "distance", k=k)


def test_target_with_n_runs_eight_row_layout(engine):
    target = synth.masked_genome(67, 60_000, frac_lower=0.0)
    assert 5 <= len(set(target.tolist())) <= 8
    rng = np.random.default_rng(68)
    lengths = _EDGES + [int(x) for x in rng.integers(257, 513, 400)]
    reads = _reads(target, lengths, 69, unrelated_every=8)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 1
    _check(engine, reads[:200], target, "locations")


def test_nine_symbols_keep_long_reads_on_the_pair_path(engine):
    target = synth.masked_genome(70, 20_000)
    assert len(set(target.tolist())) > 8
    reads = _reads(target, [300, 400, 512, 257] * 8 + [100, 150] * 8, 71)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 3                       # short reads on the lane kernels, long ones on the rings


def test_neighbours_of_the_range_and_short_reads_in_one_batch(engine):
    target = synth.random_dna(72, 40_000)
    lengths = [1, 31, 32, 33, 150, 255, 256, 257, 512, 513, 514, 600, 1000] * 6
    reads = _reads(target, [max(m, 1) for m in lengths], 73, unrelated_every=5)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 1
    _check(engine, reads, target, "path")


def test_large_batch_through_probe_and_both_passes(engine):
    """>= 16384 slots per group: the k-doubling probe, pass 1 at a small threshold, and >= 4096 unrelated leftovers
    whose band is the whole query (scan_reads_full_kernel<12 / 16>)"""
    target = synth.random_dna(74, 12_000)
    rng = np.random.default_rng(75)
    lengths = [int(x) for x in rng.integers(257, 385, 16600)] + [int(x) for x in rng.integers(385, 513, 16600)]
    reads = _reads(target, lengths, 76, unrelated_every=3, max_err=0.03)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 1


def test_repeats_overflow_the_end_location_lists(engine):
    """a tandem repeat: every copy ends an alignment of the same score (more than the 16 positions a slot keeps)"""
    unit = synth.random_dna(77, 331)
    target = np.tile(unit, 60)
    rng = np.random.default_rng(78)
    reads = []
    for m in (300, 331, 400, 500):
        for _ in range(6):
            r = np.tile(unit, 3)[17:17 + m].copy()
            for p in rng.integers(0, m, 3):
                r[p] = _ACGT[(np.searchsorted(_ACGT, r[p]) + 1) % 4]
            reads.append(r)
    _check(engine, reads, target, "distance")
    _check(engine, reads, target, "locations")


_EDGES_1K = [513, 514, 543, 544, 545, 576, 577, 640, 641, 704, 705, 736, 737, 767, 768, 769, 770, 800, 801, 832, 833, 896,
             897, 960, 961, 992, 993, 1023, 1024]


@pytest.mark.parametrize("task", ["distance", "locations", "path"])
def test_mixed_lengths_513_to_1024(engine, task):
    target = synth.random_dna(81, 60_000)
    rng = np.random.default_rng(82)
    n = 500 if task == "distance" else 150
    lengths = _EDGES_1K + [int(x) for x in rng.integers(513, 1025, n)]
    reads = _reads(target, lengths, 83, unrelated_every=9)
    st = _check(engine, reads, target, task)
    if task == "distance":
        assert st["path"] == 1, "a read of 513..1024 bases left the reads-per-lane kernels"


@pytest.mark.parametrize("k", [0, 5, 60, 150, 2000])
def test_fixed_k_up_to_1024(engine, k):
    target = synth.random_dna(84, 40_000)
    rng = np.random.default_rng(85 + k)
    lengths = _EDGES_1K + [int(x) for x in rng.integers(257, 1025, 200)]
    reads = _reads(target, lengths, 86 + k, unrelated_every=7, max_err=0.2)
    _check(engine, reads, target, "distance", k=k)


def test_every_group_in_one_batch_and_1025_on_the_pair_path(engine):
    target = synth.random_dna(87, 30_000)
    lengths = [20, 150, 256, 257, 384, 385, 512, 513, 768, 769, 1024, 1025, 1500] * 5
    reads = _reads(target, lengths, 88, unrelated_every=6)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 3
    _check(engine, reads, target, "locations")


def test_five_symbols_stop_at_512(engine):
    target = synth.masked_genome(89, 30_000, frac_lower=0.0)
    reads = _reads(target, [300, 512, 513, 700, 1024] * 8, 90)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 3


def test_large_long_batch_through_probe_and_both_passes(engine):
    """>= 16384 slots in the 24- and 32-word groups: probe, pass 1, and unrelated leftovers on scan_reads_full_kernel"""
    target = synth.random_dna(91, 8_000)
    rng = np.random.default_rng(92)
    lengths = [int(x) for x in rng.integers(513, 769, 16500)] + [int(x) for x in rng.integers(769, 1025, 16500)]
    reads = _reads(target, lengths, 93, unrelated_every=3, max_err=0.02)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 1


def test_equalities_with_long_reads(engine):
    target = synth.random_dna(94, 30_000)
    reads = _reads(target, [300, 500, 700, 1000] * 12, 95)
    for r in reads[::3]:
        r[::17] = ord("N")
    _check(engine, reads, target, "locations", eq=[("N", "A"), ("N", "C"), ("N", "G"), ("N", "T")])
