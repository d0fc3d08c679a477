"""-m gpu: the results view of a TASK_DISTANCE batch of reads of one word count is made on the device (engine_flat.hip:
buildReadsView -- the per-slot results of the scans laid out by flat_results.hip, no per-read host record).  Against the
reference: every field of the flat view; against the per-unit records of the same run (edlibAmdBatchResults builds them on
demand): the same answers by the other route.  Repeats (lists beyond the 16 kept per read: the exact pass), unrelated reads,
fixed k, the three modes, the empty-prefix rule of lengths that are not a multiple of 64."""
import numpy as np
import pytest

from edlib_amd import synth
from oracle import oracle as O
from test_gpu_many_segments import _target_with_repeats
from test_gpu_long_reads import _reads

pytestmark = pytest.mark.gpu


def _compare(engine, reads, target, mode, k):
    R = np.ascontiguousarray(np.stack(reads))
    n, m = R.shape
    b = engine.SharedBatch(R, target, mode=mode, task="distance", k=k)
    try:
        b.run()
        got = b.results_flat()                     # the device-made view (nothing collected yet)
        rec = b.results(raw=True)                  # ... and the per-unit malloc'd results, copied from that view
        b.run()
        rec2 = b.results(raw=True)                 # the other order: per-unit results first (copied from the same view), then the view
        got2 = b.results_flat()
    finally:
        b.close()
    ref = O.pool_align(R.reshape(-1), np.arange(n + 1, dtype=np.int64) * m, target, np.array([0, len(target)], dtype=np.int64),
                       True, mode, "distance", k)
    for g in (got, got2):
        for f in ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "alnOff"):
            assert np.array_equal(g[f], ref[f]), (mode, k, f)
        assert g["starts"] is None or len(g["starts"]) == 0 or np.all(np.asarray(g["starts"]) == -1)
    lo = got["locOff"]
    for records in (rec, rec2):
        for u in range(n):
            assert records[u]["editDistance"] == got["editDistance"][u]
            assert list(records[u]["endLocations"] or []) == list(got["ends"][lo[u]:lo[u + 1]]), (mode, k, u)


@pytest.mark.parametrize("mode,k", [("HW", -1), ("HW", 3), ("HW", 40), ("SHW", -1), ("SHW", 5), ("NW", -1), ("NW", 30)])
def test_view_of_a_distance_batch_of_reads(engine, mode, k):
    T = 300_000 if mode == "HW" else 140                    # (SHW / NW: reads against a target of their own size)
    target, motifs = _target_with_repeats(401, T, [40, 6]) if mode == "HW" else (synth.random_dna(401, T), [])
    m = 150 if mode == "HW" else 128                        # 150: the empty prefix takes part (W = 42); 128: it does not
    reads = _reads(target, [m] * 1400, 402, unrelated_every=9, max_err=0.05) if mode == "HW" else \
        [synth.mutate(target[:m + 8], 403, 0.03, 0.01, 0.01, stream=i)[0][:m] for i in range(1400)]
    reads = [r if len(r) == m else np.resize(r, m) for r in reads]
    for motif in motifs:
        for s in (0, 100, 250):
            reads.append(np.ascontiguousarray(motif[s:s + m]))
    _compare(engine, reads, target, mode, k)


@pytest.mark.parametrize("mode,k", [("HW", -1), ("HW", 4), ("SHW", -1), ("NW", -1)])
def test_several_word_count_groups_are_gathered_into_unit_order(engine, mode, k):
    """reads of 40..160 bases in arbitrary order: five groups (one of them a handful of reads whose results sit in pinned host
    memory), slot != unit; repeats give some of them lists of the exact pass"""
    T = 200_000 if mode == "HW" else 150
    target, motifs = _target_with_repeats(406, T, [30, 5]) if mode == "HW" else (synth.random_dna(406, T), [])
    rng = np.random.default_rng(407)
    lengths = [int(x) for x in rng.integers(40, 161, 1500)] + [20, 25, 31]            # (the last three: a group of their own)
    if mode == "HW":
        reads = _reads(target, lengths, 408, unrelated_every=8, max_err=0.05)
        for motif in motifs:
            for s0, m in ((0, 70), (50, 100), (200, 150), (10, 129)):
                reads.append(np.ascontiguousarray(motif[s0:s0 + m]))
    else:
        reads = [np.resize(synth.mutate(target[:m + 8], 409, 0.03, 0.01, 0.01, stream=i)[0], m) for i, m in enumerate(lengths)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    b = engine.SharedBatch(reads, target, mode=mode, task="distance", k=k)
    try:
        b.run(); got = b.results_flat(); rec = b.results(raw=True)
    finally:
        b.close()
    qoff = np.zeros(len(reads) + 1, dtype=np.int64); qoff[1:] = np.cumsum([len(r) for r in reads])
    ref = O.pool_align(np.concatenate(reads), qoff, target, np.array([0, len(target)], dtype=np.int64), True, mode, "distance", k)
    for f in ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "alnOff"):
        assert np.array_equal(got[f], ref[f]), (mode, k, f)
    lo = got["locOff"]
    for u in range(len(reads)):
        assert rec[u]["editDistance"] == got["editDistance"][u]
        assert list(rec[u]["endLocations"] or []) == list(got["ends"][lo[u]:lo[u + 1]]), (mode, k, u)
