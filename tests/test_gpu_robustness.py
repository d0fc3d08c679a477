"""-m gpu: shapes the fixtures do not reach -- concurrent callers, mixed query lengths in one batch
(every word-count group of the reads kernel + the pair-kernel fallback + empty queries), noisy
batches that make the k-doubling probe fall back to a single full pass, wider alphabets."""
import threading

import numpy as np
import pytest

from edlib_amd import synth

pytestmark = pytest.mark.gpu
FIELDS = ("status", "editDistance", "endLocations", "startLocations", "numLocations",
          "alignment", "alignmentLength", "alphabetLength")


def same(got, want):
    return all(got[f] == want[f] for f in FIELDS)


def test_concurrent_callers(engine, oracle):
    """edlibAlign is re-entrant in the reference (bindings/python/edlib.pyx:128 calls it nogil)."""
    target = synth.random_dna(41, 6000).tobytes()
    reads = synth.illumina_reads(np.frombuffer(target, dtype=np.uint8), 64, m=120, seed=42)["reads"]
    want = [oracle.align(r.tobytes(), target, "HW", "path", -1) for r in reads]
    bad = []

    def work(tid):
        for i in range(tid, len(reads), 8):
            got = engine.align_raw(reads[i].tobytes(), target, "HW", "path", -1)
            if not same(got, want[i]):
                bad.append(i)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not bad, bad


@pytest.mark.parametrize("mode,task", [("HW", "distance"), ("HW", "path"), ("SHW", "locations"), ("NW", "distance")])
def test_mixed_lengths_in_one_batch(engine, oracle, mode, task):
    target = synth.random_dna(43, 3000)
    qs = [b""]
    for i, m in enumerate([1, 2, 31, 32, 33, 63, 64, 65, 96, 97, 128, 150, 160, 161, 200, 255, 256, 257, 300, 420]):
        a = (i * 131) % (len(target) - m)
        q, _ = synth.mutate(target[a:a + m], 44 + i, 0.04, 0.01, 0.01)
        qs.append(q.tobytes() or b"A")
    got = engine.align_batch(qs, target.tobytes(), mode=mode, task=task, raw=True)
    for q, g in zip(qs, got):
        want = oracle.align(q, target.tobytes(), mode, task, -1)
        assert same(g, want), (mode, task, len(q), g, want)


def test_noisy_batch_takes_the_full_threshold_pass(engine, oracle):
    """>= 16384 reads with ~12 % errors: almost nothing resolves at the first threshold of the
    k-doubling, so the probe must switch to one full pass -- same answers either way."""
    target = synth.random_dna(45, 120000)
    r = synth.illumina_reads(target, 16500, m=100, seed=46, sub=0.08, ins=0.02, dele=0.02, frac_random=0.2)
    got = engine.align_batch(r["reads"], target.tobytes(), mode="HW", task="distance", raw=True)
    tb = target.tobytes()
    for i in range(0, len(got), 97):
        want = oracle.align(r["reads"][i].tobytes(), tb, "HW", "distance", -1)
        assert same(got[i], want), i


def test_fixed_k_shared_batch(engine, oracle):
    target = synth.random_dna(47, 40000)
    r = synth.illumina_reads(target, 300, m=150, seed=48)
    for k in (0, 2, 7, 8, 9, 40, 200):
        got = engine.align_batch(r["reads"], target.tobytes(), mode="HW", task="locations", k=k, raw=True)
        for i in range(0, 300, 7):
            want = oracle.align(r["reads"][i].tobytes(), target.tobytes(), "HW", "locations", k)
            assert same(got[i], want), (k, i)


def test_wider_alphabets_go_through_the_pair_kernel(engine, oracle):
    for sigma in (5, 20, 33, 90):
        t = synth.random_symbols(50 + sigma, 2500, sigma, base=33)
        qs = []
        for i in range(12):
            a = 100 * i
            q = t[a:a + 180].copy()
            q[::17] = 33 + (q[::17] - 33 + 1) % sigma
            qs.append(q.tobytes())
        got = engine.align_batch(qs, t.tobytes(), mode="HW", task="path", raw=True)
        for q, g in zip(qs, got):
            assert same(g, oracle.align(q, t.tobytes(), "HW", "path", -1)), sigma


def test_exact_pass_at_full_target_length(engine, ref, oracle):
    """4096 reads vs the 5 Mb target: a handful of reads have more end locations than the first pass
    keeps; their lists come from the exact pass, whose launch has a ragged last wave (this once read its
    offset tables out of bounds).  Every overflowing read and a sample of the rest must match."""
    impl = ref or oracle
    target = synth.random_dna(12345, 5_000_000)
    r = synth.illumina_reads(target, 4096, m=150, seed=77)
    b = engine.SharedBatch(r["reads"], target, mode="HW", task="distance")
    st = b.run()
    got = b.results(raw=True)
    b.close()
    assert st["overflow_units"] > 0
    # a unit overflows when one target segment holds > 8 of its end locations or the whole list > 16
    many = [i for i, g in enumerate(got) if g["numLocations"] > 8]
    assert len(many) >= st["overflow_units"] >= sum(1 for g in got if g["numLocations"] > 16)
    tb = target.tobytes()
    for i in many + list(range(0, 4096, 512)):
        want = impl.align(r["reads"][i].tobytes(), tb, "HW", "distance", -1)
        assert same(got[i], want), i


def test_oneshot_entry_points_shard_over_devices(engine, oracle, monkeypatch):
    """edlibAlignBatchSharedTarget / edlibAlignBatchPairs split the units into contiguous shards, one host
    thread + stream per listed device.  On a 1-GPU box the same device is listed 3 times."""
    target = synth.random_dna(61, 30000)
    reads = synth.illumina_reads(target, 50, m=150, seed=62)["reads"]
    qs = [r.tobytes() for r in reads] + [b""]
    for devs in ("0", "0,0,0"):
        monkeypatch.setenv("EDLIB_AMD_DEVICES", devs)
        got = engine.align_batch_oneshot(qs, target.tobytes(), mode="HW", task="path")
        for q, g in zip(qs, got):
            assert same(g, oracle.align(q, target.tobytes(), "HW", "path", -1)), devs
        ts = [target[i * 100:i * 100 + 170].tobytes() for i in range(len(qs))]
        got = engine.align_batch_oneshot(qs, None, targets=ts, mode="NW", task="distance")
        for q, t, g in zip(qs, ts, got):
            assert same(g, oracle.align(q, t, "NW", "distance", -1)), devs


def test_pair_batches_of_unequal_work_are_pulled_from_a_chunk_queue(engine, checker, monkeypatch):
    """edlibAlignBatchPairs over several devices: pairs differ in work (query x target cells), so the device threads pull
    chunks of about equal work from a shared queue instead of taking static slices.  Two "devices" on the
    one GPU of the test box; a few long pairs among many short ones, in both orders; EDLIB_AMD_SHARD=static: the slices."""
    rng = np.random.default_rng(77)
    qs, ts = [], []
    for i in range(60):
        n = int(rng.choice([40, 150, 700, 5000])) if i % 13 else 20000
        t = synth.random_dna(500 + i, n)
        q, _ = synth.mutate(t, 900 + i, 0.03, 0.01, 0.01)
        qs.append(q.tobytes()); ts.append(t.tobytes())
    for order in (1, -1):
        for how in (None, "static"):
            monkeypatch.setenv("EDLIB_AMD_DEVICES", "0,0")
            if how:
                monkeypatch.setenv("EDLIB_AMD_SHARD", how)
            else:
                monkeypatch.delenv("EDLIB_AMD_SHARD", raising=False)
            a, b = qs[::order], ts[::order]
            for mode, task in (("NW", "distance"), ("HW", "locations")):
                got = engine.align_batch_oneshot(a, None, targets=b, mode=mode, task=task)
                for q, t, g in zip(a, b, got):
                    assert same(g, checker.align(q, t, mode, task, -1)), (order, how, mode, len(q))


def test_large_batch_takes_probe_and_leftover_paths(engine, ref, oracle):
    """A batch big enough for everything the k-doubling does above 16,384 reads: the 2048-read probe with its
    adaptive first threshold, a second pass with more than 4096 leftovers (band sample, plain full-height kernel),
    and leftovers of a middling distance.  A strided sample of 1,500 reads is checked against the reference on
    all host cores; whole-batch invariants cover the rest."""
    import os
    impl = ref if ref is not None else oracle
    target = synth.random_dna(77, 120000)
    n = 24000
    rd = synth.illumina_reads(target, n, m=150, seed=78, frac_random=0.22)
    reads = rd["reads"].copy()
    rng = np.random.default_rng(79)
    noisy = np.arange(0, n, 9)                      # ~11 %: 6 % substitutions on top -> distances around 8..14
    for i in noisy:
        pos = rng.choice(150, size=9, replace=False)
        reads[i, pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=9)
    tbytes = target.tobytes()
    got = engine.align_batch([r.tobytes() for r in reads], tbytes, mode="HW", task="distance", k=-1, raw=True)
    ed = np.array([g["editDistance"] for g in got])
    assert ((ed >= 0) & (ed <= 150)).all()
    planted = ~rd["random"]
    clean = planted.copy(); clean[noisy] = False
    assert (ed[clean] <= rd["edits"][clean]).all()
    idx = list(range(0, n, 16))
    want = [None] * len(idx)
    cores = min(os.cpu_count() or 1, 64)

    def work(k):
        for j in range(k, len(idx), cores):
            want[j] = impl.align(reads[idx[j]].tobytes(), tbytes, "HW", "distance", -1)

    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    bad = [i for j, i in enumerate(idx) if not same(got[i], want[j])]
    assert not bad, (len(bad), bad[:5])


def test_callers_device_and_env_device(engine, oracle, monkeypatch):
    """no entry point leaves the calling thread on another HIP device than it found it (RAII guard), and
    edlibAlign() takes EDLIB_AMD_DEVICE (an ordinal out of range falls back to the current device)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")                     # the runtime libedlib.so itself is linked against

    def current():
        d = C.c_int(-1)
        assert hip.hipGetDevice(C.byref(d)) == 0
        return d.value
    assert hip.hipSetDevice(0) == 0
    q, t = b"ACGTTGCAAC", b"TTACGTAGCAACGG"
    want = oracle.align(q, t, "HW", "path", -1)
    for env in (None, "0", "63"):
        if env is None:
            monkeypatch.delenv("EDLIB_AMD_DEVICE", raising=False)
        else:
            monkeypatch.setenv("EDLIB_AMD_DEVICE", env)
        got = engine.align_raw(q, t, "HW", "path", -1)
        assert got == want
        assert current() == 0
    b = engine.SharedBatch([q, q], t, mode="HW", task="distance")
    b.run(); b.results(); b.close()
    assert current() == 0
    engine.lib().edlibAmdTrim()
    assert engine.align_raw(q, t, "HW", "path", -1) == want          # the cache refills after a trim


def test_shw_against_long_targets_stops_at_2m(engine, checker):
    """prefix mode: D[m][j] >= j - m, so nothing beyond column 2m can tie the best score -- the scans stop there (reads
    against a long shared target, pairs with long targets, the flat pair path); every field still equals the reference"""
    import numpy as np
    from edlib_amd import synth
    target = synth.random_dna(301, 300_000)
    rng = np.random.default_rng(302)
    reads = []
    for m in (1, 31, 64, 150, 255, 256, 300, 700):
        reads.append(target[:m].copy())                                   # prefix itself: distance 0, end m - 1
        r = target[3:3 + m].copy(); r[::7] = ord("A"); reads.append(r)
        reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, m)])
    for task in ("distance", "locations", "path"):
        got = engine.align_batch([r.tobytes() for r in reads], target.tobytes(), mode="SHW", task=task, raw=True)
        for r, g in zip(reads, got):
            want = checker.align(r.tobytes(), target.tobytes(), "SHW", task, -1)
            assert all(g[f] == want[f] for f in ("status", "editDistance", "endLocations", "startLocations", "numLocations", "alignment", "alphabetLength")), (task, len(r))
    qs = [r.tobytes() for r in reads] * 150                                # 3600 pairs: the flat pair path
    ts = [target[:5000 + 13 * i].tobytes() for i in range(len(qs))]
    got = engine.align_pairs(qs, ts, mode="SHW", task="distance", raw=True)
    for i in range(0, len(qs), 37):
        want = checker.align(qs[i], ts[i], "SHW", "distance", -1)
        assert got[i]["editDistance"] == want["editDistance"] and got[i]["endLocations"] == want["endLocations"], i


def test_invalid_mode_values(engine, checker):
    """EdlibAlignConfig.mode outside {0, 1, 2} through the C ABI (edlib.cpp:205-225, 177-179): the
    distance as NW, no locations; ERROR with an empty input.  Every field against the reference."""
    fields = ("status", "editDistance", "endLocations", "startLocations", "numLocations", "alignment", "alignmentLength", "alphabetLength")
    cases = [(b"ACGTACGT", b"ACGTTCGT"), (b"AAAA", b"TTTTTT"), (b"ACGT" * 40, b"ACGA" * 41), (b"", b"ACGT"), (b"ACGT", b""), (b"", b""),
             (synth.random_dna(1, 3000).tobytes(), synth.random_dna(2, 3100).tobytes())]
    for mode in (3, 7, -1, 100):
        for task in ("distance", "locations"):
            for k in (-1, 0, 3):
                for q, t in cases:
                    got = engine.align_raw(q, t, mode, task, k)
                    want = checker.align(q, t, mode, task, k)
                    for f in fields:
                        assert got[f] == want[f], (mode, task, k, len(q), len(t), f, got[f], want[f])
        # TASK_PATH with such a mode: the reference dereferences its NULL endLocations (a crash, no answer to match);
        # here it has to come back with a status and no stray pointers
        for q, t in cases:
            got = engine.align_raw(q, t, mode, "path", -1)
            assert got["status"] in (0, 1) and got["alignment"] is None and got["endLocations"] is None
    # and in a batch (the batch entry points take the same config)
    qs = [c[0] for c in cases[:3]]; ts = [c[1] for c in cases[:3]]
    got = engine.align_pairs(qs, ts, mode=7, task="locations", raw=True)
    for g, q, t in zip(got, qs, ts):
        want = checker.align(q, t, 7, "locations", -1)
        for f in fields:
            assert g[f] == want[f], (f, g[f], want[f])
