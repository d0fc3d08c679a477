"""-m gpu: one unit on many waves (scan_pairs_wide_kernel, edlib_amd/csrc/wide_kernels.hip) against the compiled
reference: NW distances beyond the band of every lane ring (K > 3968), the K ladder from a small start, fewer pipeline
slots than strips alive (hand-off buffers reused), Hirschberg halves on the wide band (TASK_PATH of long divergent
pairs), long SHW / HW queries as pipelined strips, fixed k, mixed batches."""
import os
import random

import numpy as np
import pytest

from edlib_amd import synth

pytestmark = pytest.mark.gpu
SEED_SHIFT = int(os.environ.get("EDLIB_FUZZ_SEED", "0"))
FIELDS = ("status", "editDistance", "endLocations", "startLocations", "numLocations",
          "alignment", "alignmentLength", "alphabetLength")


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


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


def _mut(rng, n, sub, ins, dele):
    t = synth.random_dna(rng.randrange(1 << 30), n)
    q, _ = synth.mutate(t, rng.randrange(1 << 30), sub, ins, dele)
    return q.tobytes(), t.tobytes()


def test_nw_distances_beyond_the_rings(engine, checker):
    """distances 4,000 ... 25,000 on queries of 5 ... 10 strips: first pass from the 4 kb prefix estimate"""
    rng = random.Random(9001 + SEED_SHIFT)
    qs, ts = [], []
    for n, rate in ((33000, 0.06), (40000, 0.10), (36000, 0.25), (50000, 0.04), (34000, 0.5)):
        q, t = _mut(rng, n, rate / 2, rate / 4, rate / 4)
        qs.append(q); ts.append(t)
    # unrelated, and very different lengths (|T - m| alone is beyond the rings)
    qs.append(synth.random_dna(5, 33000).tobytes()); ts.append(synth.random_dna(6, 36000).tobytes())
    qs.append(synth.random_dna(7, 9000).tobytes()); ts.append(synth.random_dna(8, 45000).tobytes())
    qs.append(synth.random_dna(9, 47000).tobytes()); ts.append(synth.random_dna(10, 5000).tobytes())
    for i in range(len(qs)):                           # one at a time (single call) ...
        _check(engine, checker, qs[i:i + 1], ts[i:i + 1], "NW", "distance", -1, "wide single %d" % i)
    _check(engine, checker, qs, ts, "NW", "distance", -1, "wide batch")       # ... and as one batch


@pytest.mark.parametrize("slots", [None, 1, 2, 3])
def test_k_ladder_from_a_small_start(engine, checker, slots):
    """every pair on the wide kernel (rings off), K doubling from 64; with 1 / 2 / 3 slots a wave runs several strips of
    its unit one after the other and the hand-off buffers of a slot are reused"""
    rng = random.Random(9002 + SEED_SHIFT)
    qs, ts = [], []
    for n in (1, 63, 64, 65, 700, 4095, 4096, 4097, 8192, 8200, 12289, 17000, 21000):
        for rate in (0.01, 0.2):
            q, t = _mut(rng, n, rate / 2, rate / 4, rate / 4)
            qs.append(q); ts.append(t)
    qs.append(synth.random_dna(11, 9000).tobytes()); ts.append(synth.random_dna(12, 14000).tobytes())
    with _env(EDLIB_AMD_NWBAND="0", EDLIB_AMD_WIDE_K0="64", EDLIB_AMD_WIDE_SLOTS=slots):
        _check(engine, checker, qs, ts, "NW", "distance", -1, "ladder slots=%s" % slots)
        _check(engine, checker, qs[:10], ts[:10], "NW", "distance", 40, "ladder fixed k")


def test_fixed_k_around_a_wide_distance(engine, checker):
    rng = random.Random(9003 + SEED_SHIFT)
    q, t = _mut(rng, 30000, 0.1, 0.05, 0.05)
    d = checker.align(q, t, "NW", "distance", -1)["editDistance"]
    assert d > 3968
    for k in (d - 1, d, d + 1, 3968, 4000, 2 * d):
        _check(engine, checker, [q], [t], "NW", "distance", k, "fixed k %d (d %d)" % (k, d))


def test_paths_of_long_divergent_pairs(engine, ref):
    """TASK_PATH: Hirschberg levels whose halves run inside bands beyond the rings (edlib.cpp:1231-1396)"""
    if ref is None:
        pytest.skip("the oracle restatement has no Hirschberg regime")
    rng = random.Random(9004 + SEED_SHIFT)
    qs, ts = [], []
    for n, rate in ((30000, 0.2), (45000, 0.12), (20000, 0.5)):
        q, t = _mut(rng, n, rate / 2, rate / 4, rate / 4)
        qs.append(q); ts.append(t)
    _check(engine, ref, qs, ts, "NW", "path", -1, "wide paths")
    _check(engine, ref, qs[:1], ts[:1], "NW", "path", -1, "wide path single")


@pytest.mark.parametrize("mode", ["SHW", "HW"])
def test_long_semi_global_queries_as_pipelined_strips(engine, checker, mode):
    rng = random.Random(9005 + SEED_SHIFT + len(mode))
    qs, ts = [], []
    for n, tn in ((4097, 9000), (9000, 30000), (13000, 13500), (20000, 5000)):
        t = synth.random_dna(rng.randrange(1 << 30), tn)
        a = 0 if mode == "SHW" else rng.randrange(0, max(1, tn - n))
        src = t[a:a + n] if tn >= n else synth.random_dna(rng.randrange(1 << 30), n)
        q, _ = synth.mutate(src, rng.randrange(1 << 30), 0.03, 0.01, 0.01)
        qs.append(q.tobytes()); ts.append(t.tobytes())
    qs.append(synth.random_dna(21, 5000).tobytes()); ts.append(synth.random_dna(22, 7000).tobytes())   # unrelated
    for task in ("distance", "locations"):
        _check(engine, checker, qs, ts, mode, task, -1, "strips %s" % task)
    with _env(EDLIB_AMD_WIDE_SLOTS=1):
        _check(engine, checker, qs, ts, mode, "distance", -1, "strips, one slot")
    _check(engine, checker, qs[:2], ts[:2], mode, "distance", 300, "strips fixed k")


def test_shw_inside_the_band_of_a_threshold(engine, checker):
    """SHW pairs on the smallest ring that holds the BAND [-K, K] (4 / 16 lanes, 16 lanes of 2 / 4 blocks) or on the wide
    kernel inside that band: fixed k below / at / above the distance, k = -1 (levels 256, 1024, ...), the reverse scans
    of HW start locations (k = the distance) behind HW locations / path on long queries, targets shorter than m - k."""
    rng = random.Random(9006 + SEED_SHIFT)
    qs, ts = [], []
    for m in (300, 1100, 2500, 5000, 12000):
        for rate in (0.01, 0.08, 0.3):
            t = synth.random_dna(rng.randrange(1 << 30), m + rng.randrange(0, 3000))
            q, _ = synth.mutate(t[:m], rng.randrange(1 << 30), rate / 2, rate / 4, rate / 4)
            qs.append(q.tobytes()); ts.append(t.tobytes())
    qs.append(synth.random_dna(31, 6000).tobytes()); ts.append(synth.random_dna(32, 9000).tobytes())   # unrelated
    qs.append(synth.random_dna(33, 6000).tobytes()); ts.append(synth.random_dna(34, 2000).tobytes())   # target shorter than the query
    for task in ("distance", "locations"):
        _check(engine, checker, qs, ts, "SHW", task, -1, "shw band, open")
    ds = [checker.align(q, t, "SHW", "distance", -1)["editDistance"] for q, t in zip(qs, ts)]
    for k in (5, 40, 130, 500, 2000):
        _check(engine, checker, qs, ts, "SHW", "locations", k, "shw band, fixed k")
    for i in (1, 4, 7, 10, 13):                          # one unit at a time right around its own distance
        for k in (ds[i] - 1, ds[i], ds[i] + 1):
            if k >= 0:
                _check(engine, checker, qs[i:i + 1], ts[i:i + 1], "SHW", "locations", k, "shw band at its distance")
    # HW on long queries: every end location gets a reverse SHW scan with k = the distance (edlib.cpp:253-257)
    hq, ht = [], []
    for m, rate in ((1100, 0.02), (3000, 0.05), (7000, 0.01), (9000, 0.1)):
        t = synth.random_dna(rng.randrange(1 << 30), 3 * m)
        a = rng.randrange(0, 2 * m)
        q, _ = synth.mutate(t[a:a + m], rng.randrange(1 << 30), rate / 2, rate / 4, rate / 4)
        hq.append(q.tobytes()); ht.append(t.tobytes())
    for task in ("locations", "path"):
        _check(engine, checker, hq, ht, "HW", task, -1, "hw starts behind banded reverse scans")
    with _env(EDLIB_AMD_SHWBAND="0"):
        _check(engine, checker, qs[:6], ts[:6], "SHW", "locations", 40, "shw band off")


def test_concurrent_callers_take_turns_on_the_wide_kernel(engine, checker):
    """edlibAlign() on long pairs from several host threads at once: the strip pipelines spin on each other, so two wide
    launches must never share the device's resident-wave budget (Batch::launchWide: a per-device gate)"""
    import threading
    rng = random.Random(9007 + SEED_SHIFT)
    pairs = [_mut(rng, n, 0.06, 0.03, 0.03) for n in (30000, 42000, 36000, 25000)]
    want = [checker.align(q, t, "NW", "distance", -1)["editDistance"] for q, t in pairs]
    got = [[None] * 3 for _ in pairs]

    def work(i):
        for r in range(3):
            got[i][r] = engine.align_raw(pairs[i][0], pairs[i][1], "NW", "distance", -1)["editDistance"]
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(pairs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert got == [[w] * 3 for w in want]


@pytest.mark.parametrize("where", ["process", "stream"])
def test_wide_launch_survives_a_cu_hog(engine, checker, where):
    """Something else holds all but 12 wave slots of the device (tools/cu_hog.hip) while a 30 k x 30 k NW call wants 30
    workgroups that wait for each other: a second PROCESS, or a second stream of THIS process.  Whatever the device's
    scheduler makes of it -- on this pool the call waits its turn in both cases; a launch that fitted only in part
    would notice at entry (wide_kernels.hip: wide_all_resident) and run again with one slot per unit -- the call
    returns the right distance, not EDLIB_STATUS_ERROR, and does not hang (the reference always returns:
    edlib.cpp:197-217).  The rerun itself is exercised by the next test."""
    import ctypes
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, so = os.path.join(root, "build", "cu_hog"), os.path.join(root, "build", "libcu_hog.so")
    if not os.path.exists(exe if where == "process" else so):
        pytest.skip("build/cu_hog was not built")
    rng = random.Random(9100 + SEED_SHIFT)
    q, t = _mut(rng, 30000, 0.05, 0.025, 0.025)
    want = checker.align(q, t, "NW", "distance", -1)
    b = engine.PairBatch([q], [t], mode="NW", task="distance", k=-1)
    try:
        st = b.run()                                  # undisturbed: pipelined strips, no retry
        assert st["wide_retries"] == 0 and b.results()[0]["editDistance"] == want["editDistance"]
        hog = lib = None
        if where == "process":
            hog = subprocess.Popen([exe, "--leave", "12", "--seconds", "4"], stdout=subprocess.PIPE, text=True)
            line = hog.stdout.readline().split()      # "resident <n> of <m>"
            assert line and line[0] == "resident", line
            resident, total = int(line[1]), int(line[3])
        else:
            lib = ctypes.CDLL(so)
            lib.cu_hog_start.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int)]
            tot = ctypes.c_int(0)
            resident = lib.cu_hog_start(12, 4.0, ctypes.byref(tot))
            total = tot.value
            assert resident >= 0
        try:
            t0 = time.time()
            st = b.run()
            wall = time.time() - t0
            got = b.results()[0]
        finally:
            if hog is not None:
                hog.wait(timeout=60)
            else:
                assert lib.cu_hog_wait() == 0
        assert got["status"] == 0 and got["editDistance"] == want["editDistance"] and got["endLocations"] == want["endLocations"]
        assert resident <= total and wall < 20, (resident, total, wall, st["wide_retries"])
        st = b.run()                                  # the hog is gone: pipelined again
        assert st["wide_retries"] == 0 and b.results()[0]["editDistance"] == want["editDistance"]
    finally:
        b.close()


def test_wide_rerun_with_one_slot_after_a_launch_that_is_not_resident(engine, checker, ref):
    """EDLIB_AMD_WIDE_TEST_NOT_RESIDENT makes the residency check of every pipelined launch wait for one workgroup more
    than the launch has: it gives up after 0.2 s exactly as if part of the launch had not fitted the device, the launch's
    waves leave, and the same units run again with ONE slot each (a wave then only reads granules it wrote itself) --
    right answers for distances (two half scans), Hirschberg paths on the wide band, long SHW / HW queries; the next run
    without the hook is pipelined again."""
    rng = random.Random(9200 + SEED_SHIFT)
    q, t = _mut(rng, 30000, 0.05, 0.025, 0.025)
    want = checker.align(q, t, "NW", "distance", -1)
    b = engine.PairBatch([q], [t], mode="NW", task="distance", k=-1)
    try:
        with _env(EDLIB_AMD_WIDE_TEST_NOT_RESIDENT="1"):
            st = b.run()
        assert st["wide_retries"] >= 1 and b.results()[0]["editDistance"] == want["editDistance"]
        st = b.run()
        assert st["wide_retries"] == 0 and b.results()[0]["editDistance"] == want["editDistance"]
    finally:
        b.close()
    with _env(EDLIB_AMD_WIDE_TEST_NOT_RESIDENT="1"):
        qs, ts = [], []
        for n, rate in ((33000, 0.06), (36000, 0.25)):
            a, c = _mut(rng, n, rate / 2, rate / 4, rate / 4)
            qs.append(a); ts.append(c)
        _check(engine, checker, qs, ts, "NW", "distance", -1, "rerun: distances")
        if ref is not None:
            _check(engine, ref, qs[:1], ts[:1], "NW", "path", -1, "rerun: Hirschberg on the wide band")
        tq = synth.random_dna(31, 9000).tobytes()
        tt = synth.random_dna(32, 20000).tobytes()
        for mode in ("SHW", "HW"):
            _check(engine, checker, [tq], [tt], mode, "locations", -1, "rerun: long %s query" % mode)


def test_hw_inside_the_band_of_a_threshold(engine, checker):
    """HW pair units whose window is not much longer than the query, inside the static band [-K, (T - m) + 2 K]
    (Batch::solveHwBanded): fixed k below / at / above the distance, open units (levels 64, 256, ...), queries from
    5 blocks (an 8-lane ring holds them whole) to beyond 64 blocks (the wide kernel inside the band), equal hits at
    both ends of the window, targets shorter than m - k, unrelated pairs; the same answers with EDLIB_AMD_HWBAND=0;
    and the band is what runs: fewer word-steps than the whole matrix."""
    rng = random.Random(9007 + SEED_SHIFT)
    qs, ts = [], []
    for m in (320, 500, 1000, 1100, 2500, 5000, 9000):
        for rate, slack in ((0.01, 0.1), (0.04, 0.2), (0.15, 0.05)):
            lead = rng.randrange(0, int(m * slack) + 1)
            t = synth.random_dna(rng.randrange(1 << 30), m + int(m * slack))
            q, _ = synth.mutate(t[lead:lead + m], rng.randrange(1 << 30), rate / 2, rate / 4, rate / 4)
            qs.append(q.tobytes()); ts.append(t.tobytes())
    # the window holds the query twice (overlapping copies end at both sides), an unrelated pair, a target shorter than the query
    rep = synth.random_dna(77, 900).tobytes()
    qs.append(rep); ts.append(rep + rep[-150:] + rep[:60])
    qs.append(synth.random_dna(31, 3000).tobytes()); ts.append(synth.random_dna(32, 3300).tobytes())
    qs.append(synth.random_dna(33, 2000).tobytes()); ts.append(synth.random_dna(34, 1500).tobytes())
    for task in ("distance", "locations", "path"):
        _check(engine, checker, qs, ts, "HW", task, -1, "hw band, open")
    ds = [checker.align(q, t, "HW", "distance", -1)["editDistance"] for q, t in zip(qs, ts)]
    for k in (0, 5, 20, 60, 130, 500):
        _check(engine, checker, qs, ts, "HW", "locations", k, "hw band, fixed k")
    for i in (0, 4, 8, 13, 17, 20):                      # one unit at a time right around its own distance
        for k in (ds[i] - 1, ds[i], ds[i] + 1):
            if k >= 0:
                _check(engine, checker, qs[i:i + 1], ts[i:i + 1], "HW", "locations", k, "hw band at its distance")
    with _env(EDLIB_AMD_HWBAND="0"):
        _check(engine, checker, qs, ts, "HW", "locations", 20, "band off")
    # 1 kb queries in 1.2 kb windows at k = 20: 261 rows per column on 8-lane rings instead of 16 blocks
    q1, t1 = [], []
    for i in range(600):
        t = synth.random_dna(rng.randrange(1 << 30), 1200)
        lead = rng.randrange(0, 200)
        q, _ = synth.mutate(t[lead:lead + 1000], rng.randrange(1 << 30), 0.005, 0.002, 0.002)
        q1.append(q.tobytes()); t1.append(t.tobytes())
    steps = {}
    for band in ("1", "0"):
        with _env(EDLIB_AMD_HWBAND=band):
            b = engine.PairBatch(q1, t1, mode="HW", task="distance", k=20)
            try:
                steps[band] = b.run()["word_steps"]
                got = b.results()
            finally:
                b.close()
        for i in range(0, 600, 37):
            want = checker.align(q1[i], t1[i], "HW", "distance", 20)
            assert got[i]["editDistance"] == want["editDistance"] and got[i]["endLocations"] == want["endLocations"], (band, i)
    assert steps["1"] * 3 < steps["0"] * 2, steps         # (at least a third fewer executed word-steps; the geometry says ~2.9x)


def test_banded_units_with_overflowing_end_location_lists(engine, checker):
    """a banded semi-global unit of more than 64 blocks on a ring of 4-block lanes, in a periodic target: more end locations
    than the 16 a unit's list keeps -- its exact second pass runs on the strips and needs their hand-off buffer (a device
    fault in the soak of round 5 until it got one)"""
    unit = synth.random_dna(41, 331).tobytes()
    t = (unit * 40)[:9000]
    for m in (4500, 6000):
        q = (unit * 40)[100:100 + m]
        for mode in ("HW", "SHW"):
            for k in (-1, 5):
                _check(engine, checker, [q, q[:3000]], [t, t], mode, "locations", k, "periodic target m=%d" % m)
