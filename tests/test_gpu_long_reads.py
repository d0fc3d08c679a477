"""-m gpu: HW reads longer than 256 bases against a shared target: piece filter on the reads-per-lane kernel + window
verification on kernel W (edlib_amd/csrc/long_reads.hip; stats path bit 2), any length, targets of up to 16 symbols; what the
filter cannot narrow (unrelated reads, low complexity) is handed back to kernel W over the whole target.
Reference semantics: edlib.cpp:550-704 (the semi-global scan and its band), 197-217 (k-doubling).  Every field
of every read is compared with the oracle (native thread pool over the reference / the restatement).
EDLIB_AMD_FILTER=0 restores round 2's groups of 12 / 16 / 24 / 32 words (test_round2_word_groups_still_agree)."""
import os
import subprocess
import sys
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
    if task != "distance":          # (no unit with a result: the flat form has no starts array at all)
        assert np.array_equal(got["starts"], ref["starts"]) or (got["starts"] is None and len(ref["starts"]) == 0)
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
    assert st["path"] & 4, "reads of 257..512 bases did not take the piece filter"


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
    assert st["path"] & 4
    _check(engine, reads[:200], target, "locations")


def test_nine_symbols_sixteen_row_layout(engine):
    target = synth.masked_genome(70, 20_000)
    assert len(set(target.tolist())) > 8
    reads = _reads(target, [300, 400, 512, 257, 1500] * 8 + [100, 150] * 8, 71)
    st = _check(engine, reads, target, "distance")
    assert st["path"] & 1 and st["path"] & 4     # short reads on the lane kernels, long ones through the filter


def test_neighbours_of_the_range_and_short_reads_in_one_batch(engine):
    target = synth.random_dna(72, 40_000)
    lengths = [1, 31, 32, 33, 150, 255, 256, 257, 512, 513, 514, 600, 1000] * 6
    reads = _reads(target, [max(m, 1) for m in lengths], 73, unrelated_every=5)
    st = _check(engine, reads, target, "distance")
    assert st["path"] & 1 and st["path"] & 4
    _check(engine, reads, target, "path")


def test_large_batch_through_probe_and_both_passes(engine):
    """33,200 reads of 257..512 bases, a third of them unrelated (handed back to kernel W over the whole target)"""
    target = synth.random_dna(74, 12_000)
    rng = np.random.default_rng(75)
    lengths = [int(x) for x in rng.integers(257, 385, 16600)] + [int(x) for x in rng.integers(385, 513, 16600)]
    reads = _reads(target, lengths, 76, unrelated_every=3, max_err=0.03)
    st = _check(engine, reads, target, "distance")
    assert st["path"] & 4


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
    assert st["path"] & 4


@pytest.mark.parametrize("k", [0, 5, 60, 150, 2000])
def test_fixed_k_up_to_1024(engine, k):
    target = synth.random_dna(84, 40_000)
    rng = np.random.default_rng(85 + k)
    lengths = _EDGES_1K + [int(x) for x in rng.integers(257, 1025, 200)]
    reads = _reads(target, lengths, 86 + k, unrelated_every=7, max_err=0.2)
    _check(engine, reads, target, "distance", k=k)


def test_every_length_class_in_one_batch(engine):
    target = synth.random_dna(87, 30_000)
    lengths = [20, 150, 256, 257, 384, 385, 512, 513, 768, 769, 1024, 1025, 1500] * 5
    reads = _reads(target, lengths, 88, unrelated_every=6)
    st = _check(engine, reads, target, "distance")
    assert st["path"] == 7
    _check(engine, reads, target, "locations")


def test_five_symbols_any_length(engine):
    target = synth.masked_genome(89, 30_000, frac_lower=0.0)
    reads = _reads(target, [300, 512, 513, 700, 1024, 2100] * 8, 90)
    st = _check(engine, reads, target, "distance")
    assert st["path"] & 4


def test_large_long_batch_through_probe_and_both_passes(engine):
    """33,000 reads of 513..1024 bases, a third of them unrelated"""
    target = synth.random_dna(91, 8_000)
    rng = np.random.default_rng(92)
    lengths = [int(x) for x in rng.integers(513, 769, 16500)] + [int(x) for x in rng.integers(769, 1025, 16500)]
    reads = _reads(target, lengths, 93, unrelated_every=3, max_err=0.02)
    st = _check(engine, reads, target, "distance")
    assert st["path"] & 4


def test_equalities_with_long_reads(engine):
    target = synth.random_dna(94, 30_000)
    reads = _reads(target, [300, 500, 700, 1000] * 12, 95)
    for r in reads[::3]:
        r[::17] = ord("N")
    _check(engine, reads, target, "locations", eq=[("N", "A"), ("N", "C"), ("N", "G"), ("N", "T")])


# ------------------------------------------------------------------------------------------ beyond 1024 bases

_EDGES_LONG = [1025, 1026, 1279, 1280, 1281, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 5000, 8191, 8192, 8193, 10000]


@pytest.mark.parametrize("task", ["distance", "locations", "path"])
def test_lengths_1025_to_10000(engine, task):
    """every length class of the filter's plan (1 .. 40 parts), Illumina-like to ONT-like divergence, a few unrelated"""
    target = synth.random_dna(101, 120_000)
    rng = np.random.default_rng(102)
    n = 120 if task == "distance" else 40
    lengths = (_EDGES_LONG if task == "distance" else _EDGES_LONG[::3]) + [int(x) for x in rng.integers(1025, 10001, n)]
    reads = _reads(target, lengths, 103, unrelated_every=11, max_err=0.15)
    st = _check(engine, reads, target, task)
    assert st["path"] & 4


@pytest.mark.parametrize("k", [0, 7, 64, 300, 1500, 20000])
def test_fixed_k_long(engine, k):
    """the ladder is capped by the caller's k: -1 / NULL / 0 above it, the same answers below"""
    target = synth.random_dna(104, 60_000)
    rng = np.random.default_rng(105 + k)
    lengths = [1025, 2048, 4096, 6000] + [int(x) for x in rng.integers(300, 6000, 60)]
    reads = _reads(target, lengths, 106 + k, unrelated_every=9, max_err=0.12)
    _check(engine, reads, target, "distance", k=k)
    _check(engine, reads[:20], target, "locations", k=k)


def test_low_complexity_is_handed_back(engine):
    """homopolymers, short tandem repeats, a read that occurs hundreds of times: the candidate lists overflow or the windows
    cover the target, the queries go back to kernel W over the whole target -- same answers"""
    unit = synth.random_dna(107, 40)
    target = np.concatenate([synth.random_dna(108, 5000), np.tile(unit, 700), np.full(3000, ord("A"), dtype=np.uint8),
                             synth.random_dna(109, 5000)])
    reads = [np.full(400, ord("A"), dtype=np.uint8), np.tile(unit, 20)[:777].copy(), np.tile(unit, 40)[3:1503].copy(),
             np.tile(np.frombuffer(b"AC", dtype=np.uint8), 300), target[4000:5500].copy(), target[32000:33200].copy()]
    _check(engine, reads, target, "distance")
    _check(engine, reads, target, "locations")


def test_many_copies_many_windows(engine):
    """a 600-base element planted 50 times with one substitution each: 50 windows per query, all end locations reported"""
    el = synth.random_dna(110, 600)
    parts = []
    for c in range(50):
        e = el.copy(); e[(37 * c) % 600] = _ACGT[(np.searchsorted(_ACGT, e[(37 * c) % 600]) + 1) % 4]
        parts += [synth.random_dna(111 + c, 900), e]
    target = np.concatenate(parts)
    reads = [el.copy(), el[:500].copy(), el[50:].copy()]
    _check(engine, reads, target, "locations")


def test_match_at_the_target_edges(engine):
    """windows clipped at column 0 and at the last column; a query longer than the target"""
    target = synth.random_dna(112, 9000)
    reads = [target[:1500].copy(), target[-1500:].copy(), target[10:2000].copy(), target[-3000:-5].copy(),
             np.concatenate([synth.random_dna(113, 300), target[:1200]]), np.concatenate([target[-1200:], synth.random_dna(114, 300)]),
             np.concatenate([target, synth.random_dna(115, 500)])]
    for task in ("distance", "locations", "path"):
        _check(engine, reads, target, task)


def test_large_batch_of_long_reads(engine):
    """enough pieces for several segments per lane and every word-count group of the filter at once"""
    target = synth.random_dna(116, 200_000)
    rng = np.random.default_rng(117)
    lengths = [int(x) for x in rng.integers(257, 3000, 6000)]
    reads = _reads(target, lengths, 118, unrelated_every=40, max_err=0.08)
    _check(engine, reads, target, "distance")


def test_round2_word_groups_still_agree():
    """EDLIB_AMD_FILTER=0 (read when the library loads: a fresh interpreter) restores the groups of 12 / 16 / 24 / 32 words"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, edlib_amd\nfrom edlib_amd import synth\nfrom test_gpu_long_reads import _reads, _check\n"
            "t = synth.random_dna(119, 30000)\n"
            "r = _reads(t, [257, 300, 384, 385, 512, 513, 768, 769, 1024] * 8, 120, unrelated_every=9)\n"
            "st = _check(edlib_amd, r, t, 'distance'); assert st['path'] == 1, st\nprint('ok')\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, EDLIB_AMD_FILTER="0"))
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-800:] + p.stderr[-2000:]


def test_filter_from_257_bases_still_agrees():
    """EDLIB_AMD_FILTER=9 moves the switch back to where rounds 3-5 had it: reads of 257..384 bases through the piece filter
    (the default keeps them on kernel A's 12-word group: test_reads_of_257_to_384_bases_stay_on_the_lanes)"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, edlib_amd\nfrom edlib_amd import synth\nfrom test_gpu_long_reads import _reads, _check\n"
            "t = synth.random_dna(219, 30000)\n"
            "r = _reads(t, [257, 288, 300, 320, 321, 352, 383, 384] * 12, 220, unrelated_every=9)\n"
            "for task in ('distance', 'path'):\n"
            "    st = _check(edlib_amd, r, t, task); assert st['path'] & 4, st\nprint('ok')\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, EDLIB_AMD_FILTER="9"))
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-800:] + p.stderr[-2000:]


@pytest.mark.parametrize("task", ["distance", "locations", "path"])
def test_reads_of_257_to_384_bases_stay_on_the_lanes(engine, task):
    """the 12-word group of kernel A (4- and 5-symbol targets) takes them; 385 bases and more go through the filter"""
    for target in (synth.random_dna(221, 60_000), synth.masked_genome(222, 60_000, frac_lower=0.0)):
        rng = np.random.default_rng(223)
        lengths = [257, 258, 288, 289, 320, 321, 352, 353, 383, 384] * 3 + [int(x) for x in rng.integers(257, 385, 200)]
        reads = _reads(target, lengths, 224, unrelated_every=8, max_err=0.06)
        st = _check(engine, reads, target, task)
        assert st["path"] & 1 and not st["path"] & 4, st          # (bit 1: the pair kernels of start locations / paths)
        st = _check(engine, reads + _reads(target, [385, 386, 400], 225), target, task)
        assert st["path"] & 4 and st["path"] & 1, st


def test_tall_unrelated_queries_on_chained_strips(engine):
    """queries the filter hands back that are taller than the lane kernel's 32 words run as strips of 1024 rows chained
    through HBM (long_reads.hip: solveTallFull); EDLIB_AMD_TALL_MIN_WAVES=1 lets a small batch take that path.  Unrelated
    queries, low complexity, exact multiples of the strip height, one row more / less, related ones mixed in."""
    os.environ["EDLIB_AMD_TALL_MIN_WAVES"] = "1"
    try:
        target = synth.random_dna(121, 70_000)
        rng = np.random.default_rng(122)
        lengths = [1025, 1026, 1055, 1056, 1057, 2047, 2048, 2049, 2050, 3071, 3072, 3073, 3100, 4096, 4097, 5000] * 2
        reads = [_ACGT[rng.integers(0, 4, m)] for m in lengths]                        # unrelated: distance ~0.45 m
        reads += _reads(target, [1500, 2048, 2500, 4100], 123, max_err=0.45)           # too divergent for the filter
        reads += _reads(target, [1100, 2300], 124, max_err=0.05)                       # resolved by the filter
        reads.append(np.tile(np.frombuffer(b"ACGGT", dtype=np.uint8), 300))            # low complexity
        for task in ("distance", "locations"):
            _check(engine, reads, target, task)
        _check(engine, reads[:12], target, "distance", k=700)
        # five target symbols: strips of 16 words
        t5 = synth.masked_genome(125, 40_000, frac_lower=0.0)
        r5 = [_ACGT[rng.integers(0, 4, m)] for m in (513, 1024, 1025, 1600)]
        _check(engine, r5, t5, "distance")
    finally:
        os.environ.pop("EDLIB_AMD_TALL_MIN_WAVES", None)


def test_tall_unrelated_queries_default_plan(engine):
    """the same path as planned by default (no EDLIB_AMD_TALL_MIN_WAVES): enough unrelated two-strip queries for one round of
    chained-strip waves -- nine read blocks x 97 target segments of a little more than one warm-up each (round 4's plan;
    rounds 2-3: four warm-ups per segment, which a target of this length cannot give 512 waves)"""
    saved = {k: os.environ.pop(k) for k in ("EDLIB_AMD_TALL_MIN_WAVES",) if k in os.environ}
    try:
        target = synth.random_dna(131, 400_000)
        rng = np.random.default_rng(132)
        lengths = [int(x) for x in rng.integers(1100, 1301, 520)]
        reads = [_ACGT[rng.integers(0, 4, m)] for m in lengths]
        st = _check(engine, reads, target, "distance")
        assert st["path"] & 4        # (the filter ran; EDLIB_AMD_DEBUG=1 times the strips as "handed back (full height)": 19 ms here.
        #                               Bit 2 is set as well: the few queries whose end locations tie beyond a segment's list go on to kernel W)
        reads += _reads(target, [1150, 1290, 2000], 133, max_err=0.04)              # resolved by the filter, among them
        _check(engine, reads, target, "locations")
    finally:
        os.environ.update(saved)
