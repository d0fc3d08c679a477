#!/usr/bin/env python3
"""Differential soak of the batch entry points against the oracle (test infrastructure) for a time budget: random
shared-target batches over every read-length group of the lane-per-read kernels and the piece filter / chained strips
above them (1 .. 6000 bases), random pair batches over the ring sizes and the flat pair path, single edlibAlign() calls
(fused one-pair kernel); every field of every unit is compared.  usage: soak.py [seconds] [seed] [max cases]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import edlib_amd
from edlib_amd import synth
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
rng = np.random.default_rng(seed)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
FIELDS = ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "alnOff", "alignment")
LENS = [1, 5, 31, 32, 33, 64, 65, 100, 150, 160, 161, 255, 256, 257, 258, 300, 383, 384, 385, 400, 511, 512, 513, 600, 767, 768,
        769, 900, 1023, 1024, 1025, 1100, 1500, 2047, 2048, 2049, 3000, 4097, 6000]
os.environ.setdefault("EDLIB_AMD_TALL_MIN_WAVES", "1")        # small batches also take the chained strips


TRACE = os.environ.get("SOAK_TRACE") == "1"


def trace(*a):
    if TRACE:
        print("[soak]", *a, file=sys.stderr, flush=True)


def mutate(w, m, rate):
    n = int(rate * m)
    for _ in range(n):
        p = int(rng.integers(0, max(1, min(m, len(w)))))
        kind = int(rng.integers(0, 3))
        if kind == 0: w[p] = ACGT[rng.integers(0, 4)]
        elif kind == 1 and len(w) > 1: w = np.delete(w, p)
        else: w = np.insert(w, p, ACGT[rng.integers(0, 4)])
    w = w[:m]
    if len(w) < m: w = np.concatenate([w, ACGT[rng.integers(0, 4, m - len(w))]])
    return np.ascontiguousarray(w)


def compare(got, ref, task, what):
    for f in FIELDS:
        if not np.array_equal(got[f], ref[f]):
            bad = np.nonzero(got["editDistance"] != ref["editDistance"])[0][:5] if f == "editDistance" else []
            return "%s: field %s differs %s" % (what, f, list(bad))
    if task != "distance":
        # (no unit with a solution: the flat call returns no start array at all)
        gs = got["starts"] if got["starts"] is not None else np.zeros(0, dtype=np.int32)
        rs = ref["starts"] if ref["starts"] is not None else np.zeros(0, dtype=np.int32)
        if len(gs) or len(rs) or len(got["ends"]):
            if not np.array_equal(gs, rs): return "%s: starts differ" % what
    return None


def shared_case():
    kind = rng.random()
    tn = int(rng.choice([2000, 9000, 40000, 150000]))
    if kind < 0.15:
        unit = ACGT[rng.integers(0, 4, int(rng.choice([1, 2, 7, 31, 331])))]
        target = np.tile(unit, tn // len(unit) + 1)[:tn].copy()
    elif kind < 0.3:
        target = synth.masked_genome(int(rng.integers(1 << 30)), tn, frac_lower=0.0)      # ACGT + N
    else:
        target = synth.random_dna(int(rng.integers(1 << 30)), tn)
    nq = int(rng.choice([1, 7, 64, 65, 300, 1500]))
    if rng.random() < 0.4:
        lens = [int(rng.choice(LENS))] * nq
    else:
        lens = [int(rng.choice(LENS)) for _ in range(nq)]
    if max(lens) > 1100: lens = lens[:65]                 # (the reference needs ~0.1 s per such read)
    reads = []
    for m in lens:
        if rng.random() < 0.8 and tn > m + 80:
            s = int(rng.integers(0, tn - m - 70))
            reads.append(mutate(target[s:s + m + 64].copy(), m, float(rng.choice([0.0, 0.01, 0.04, 0.1, 0.25]))))
        else:
            reads.append(ACGT[rng.integers(0, 4, m)])
    mode = str(rng.choice(["HW", "HW", "HW", "HW", "SHW", "NW"]))
    if mode != "HW" and tn > 9000: target = target[:9000]
    task = str(rng.choice(["distance", "distance", "locations", "path"]))
    k = int(rng.choice([-1, -1, -1, 0, 4, 8, 9, 30, 100, 700]))
    trace("shared mode=%s task=%s k=%d tn=%d nq=%d lens=%s" % (mode, task, k, len(target), len(lens), sorted(set(lens))[:8]))
    b = edlib_amd.SharedBatch(reads, target, mode=mode, task=task, k=k)
    try:
        b.run(); got = b.results_flat()
    finally:
        b.close()
    qoff = np.zeros(len(reads) + 1, dtype=np.int64); qoff[1:] = np.cumsum([len(r) for r in reads])
    ref = O.pool_align(np.concatenate(reads), qoff, target, np.array([0, len(target)], dtype=np.int64), True, mode, task, k)
    return len(reads), compare(got, ref, task, "shared mode=%s task=%s k=%d tn=%d nq=%d lens=%s" % (mode, task, k, len(target), len(lens), sorted(set(lens))[:8]))


def pair_case():
    nq = int(rng.choice([1, 5, 17, 300, 3000]))
    base = int(rng.choice([20, 150, 400, 1000, 3000]))
    if nq == 3000 and rng.random() < 0.5: base = int(rng.choice([20, 150, 400]))      # the flat pair path (>= 1024 short pairs, distance)
    qs, ts = [], []
    sig = int(rng.choice([2, 4, 4, 4, 20]))
    alpha = np.frombuffer(b"ACGTDEFHIKLMNPQRSVWY", dtype=np.uint8)[:sig]
    for _ in range(nq):
        tn = max(1, int(base * (0.5 + rng.random())))
        t = alpha[rng.integers(0, sig, tn)]
        if rng.random() < 0.8:
            m = max(1, int(tn * (0.6 + 0.6 * rng.random())))
            q = t[:m].copy() if m <= tn else np.concatenate([t, alpha[rng.integers(0, sig, m - tn)]])
            nm = int(float(rng.choice([0.0, 0.02, 0.1, 0.3])) * m)
            for p in rng.integers(0, m, nm): q[p] = alpha[rng.integers(0, sig)]
        else:
            q = alpha[rng.integers(0, sig, max(1, int(base * (0.5 + rng.random()))))]
        qs.append(np.ascontiguousarray(q)); ts.append(np.ascontiguousarray(t))
    mode = str(rng.choice(["NW", "NW", "HW", "SHW"]))
    task = str(rng.choice(["distance", "locations", "path"]))
    k = int(rng.choice([-1, -1, 0, 5, 50, 1000]))
    trace("pairs mode=%s task=%s k=%d nq=%d base=%d sigma=%d" % (mode, task, k, nq, base, sig))
    b = edlib_amd.PairBatch(qs, ts, mode=mode, task=task, k=k)
    try:
        b.run(); got = b.results_flat()
    finally:
        b.close()
    qoff = np.zeros(nq + 1, dtype=np.int64); qoff[1:] = np.cumsum([len(r) for r in qs])
    toff = np.zeros(nq + 1, dtype=np.int64); toff[1:] = np.cumsum([len(r) for r in ts])
    ref = O.pool_align(np.concatenate(qs), qoff, np.concatenate(ts), toff, False, mode, task, k)
    return nq, compare(got, ref, task, "pairs mode=%s task=%s k=%d nq=%d base=%d sigma=%d" % (mode, task, k, nq, base, sig))


def lane_case():
    """a big NW distance batch of pairs of like lengths above 16 blocks over at most four symbols: the lane-per-pair level
     -- its pack kernel, per-unit thresholds from the probe, wave-uniform trims over 64 different lanes, units it
    leaves open (a divergent tail, unrelated pairs, foreign query bytes) and the rings that take them"""
    nq = int(rng.choice([8192, 8200, 9000, 12000]))
    base = int(rng.choice([1100, 1400, 2000, 3000]))
    sig = int(rng.choice([2, 3, 4, 4, 4]))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)[:sig]
    bulk = float(rng.choice([0.003, 0.02, 0.05, 0.1]))
    foreign = rng.random() < 0.3
    qs, ts = [], []
    for i in range(nq):
        tn = int(base * (1.0 + 0.08 * rng.random()))
        t = alpha[rng.integers(0, sig, tn)]
        x = rng.random()
        if x < 0.01:
            q = alpha[rng.integers(0, sig, int(base * (1.0 + 0.08 * rng.random())))]
        else:
            rate = 0.3 if x < 0.03 else bulk * (0.5 + rng.random())
            q = mutate(np.concatenate([t, alpha[rng.integers(0, sig, 64)]]), int(tn * (0.97 + 0.06 * rng.random())), rate)
        if foreign and rng.random() < 0.02:
            q = q.copy(); q[int(rng.integers(0, len(q)))] = ord("N")
        qs.append(np.ascontiguousarray(q)); ts.append(np.ascontiguousarray(t))
    k = int(rng.choice([-1, -1, -1, int(bulk * base), int(3 * bulk * base) + 20, 5000]))
    what = "lane level: nq=%d base=%d sigma=%d bulk=%.3f k=%d foreign=%d" % (nq, base, sig, bulk, k, foreign)
    trace(what)
    b = edlib_amd.PairBatch(qs, ts, mode="NW", task="distance", k=k)
    try:
        b.run(); got = b.results_flat()
    finally:
        b.close()
    qoff = np.zeros(nq + 1, dtype=np.int64); qoff[1:] = np.cumsum([len(r) for r in qs])
    toff = np.zeros(nq + 1, dtype=np.int64); toff[1:] = np.cumsum([len(r) for r in ts])
    ref = O.pool_align(np.concatenate(qs), qoff, np.concatenate(ts), toff, False, "NW", "distance", k)
    return nq, compare(got, ref, "distance", what)


def long_pair_case():
    """a few long pairs: the wide kernel (NW bands beyond the rings, two half scans, Hirschberg halves), SHW / HW queries
    of many strips, SHW inside the band of a threshold"""
    from oracle.oracle import load_ref, load_oracle
    chk = load_ref() or load_oracle()
    nq = int(rng.choice([1, 2, 5]))
    qs, ts = [], []
    mode = str(rng.choice(["NW", "NW", "SHW", "HW"]))
    for _ in range(nq):
        tn = int(rng.choice([3000, 5000, 9000, 20000, 40000]))
        t = synth.random_dna(int(rng.integers(1 << 30)), tn)
        rate = float(rng.choice([0.005, 0.03, 0.12, 0.3]))
        if mode == "NW":
            q, _ = synth.mutate(t, int(rng.integers(1 << 30)), rate / 2, rate / 4, rate / 4)
        else:
            m = int(tn * (0.2 + 0.7 * rng.random()))
            a = 0 if mode == "SHW" else int(rng.integers(0, tn - m + 1))
            q, _ = synth.mutate(t[a:a + m], int(rng.integers(1 << 30)), rate / 2, rate / 4, rate / 4)
        if rng.random() < 0.15: q = synth.random_dna(int(rng.integers(1 << 30)), max(1, len(q)))
        qs.append(q.tobytes()); ts.append(t.tobytes())
    task = str(rng.choice(["distance", "distance", "locations", "path"]))
    if task == "path" and chk.name != "reference": task = "locations"      # (the restatement has no Hirschberg regime)
    k = int(rng.choice([-1, -1, -1, 30, 400, 5000]))
    trace("long pairs mode=%s task=%s k=%d" % (mode, task, k), [(len(q), len(t)) for q, t in zip(qs, ts)])
    got = edlib_amd.align_pairs(qs, ts, mode=mode, task=task, k=k, raw=True)
    bad = None
    for q, t, g in zip(qs, ts, got):
        w = chk.align(q, t, mode, task, k)
        if any(g[f] != w[f] for f in ("status", "editDistance", "endLocations", "startLocations", "numLocations", "alignment", "alphabetLength")):
            bad = "long pairs mode=%s task=%s k=%d m=%d T=%d" % (mode, task, k, len(q), len(t))
    return nq, bad


def single_case():
    """edlibAlign() one pair at a time: the fused one-pair kernel and its hand-over to the general path"""
    from oracle.oracle import load_ref, load_oracle
    chk = load_ref() or load_oracle()
    bad = None
    for _ in range(40):
        T = int(rng.choice([1, 30, 64, 100, 300, 1000, 4096, 5000])); m = int(rng.choice([1, 20, 64, 65, 100, 150, 500, 1024, 1030]))
        t = ACGT[rng.integers(0, 4, T)]
        q = mutate(t[:m + 64].copy(), m, float(rng.choice([0.0, 0.03, 0.2]))) if rng.random() < 0.7 and T >= m else ACGT[rng.integers(0, 4, m)]
        mode = str(rng.choice(["NW", "HW", "SHW"])); task = str(rng.choice(["distance", "locations", "path"])); k = int(rng.choice([-1, -1, 3, 60]))
        trace("single mode=%s task=%s k=%d m=%d T=%d" % (mode, task, k, m, T))
        g = edlib_amd.align_raw(q.tobytes(), t.tobytes(), mode, task, k); w = chk.align(q.tobytes(), t.tobytes(), mode, task, k)
        if w["status"] != 2 and any(g[f] != w[f] for f in ("status", "editDistance", "endLocations", "startLocations", "numLocations", "alignment", "alphabetLength")):
            bad = "single mode=%s task=%s k=%d m=%d T=%d" % (mode, task, k, m, T)
    return 40, bad


t0 = time.time(); cases = units = 0; failures = []
while time.time() - t0 < budget and cases < max_cases:
    x = rng.random()
    if os.environ.get("SOAK_ONLY") == "lane" or x > 0.97:
        n, err = lane_case()
    else:
        n, err = shared_case() if x < 0.5 else (pair_case() if x < 0.8 else (long_pair_case() if x < 0.9 else single_case()))
    cases += 1; units += n
    if err:
        failures.append(err); print("MISMATCH", err, file=sys.stderr)
        if len(failures) >= 5: break
print(json.dumps({"seconds": round(time.time() - t0, 1), "seed": seed, "cases": cases, "units": units, "failures": failures}))
sys.exit(1 if failures else 0)
