#!/usr/bin/env python3
"""Batches of SHORT independent pairs (the verification step of a seed-and-extend mapper): n reads of 150 bases, each
against its own window -- HW against a 400-base window around its origin, NW against its 150-base mutated mate, and
100 x 100 NW; distances, start locations, paths.  Resident run time (`run_ms`: everything up to the results sitting in
HBM), the collection behind it (`results_ms`: edlibAmdBatchResultsView -- the caller-facing arrays laid out by device
kernels, one block to pinned host memory; `results_copy_ms`: the same as owned numpy copies), both per 1,000 pairs too,
GCUPS, and a strided sample checked against the oracle."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import edlib_amd
from edlib_amd import synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
T = synth.random_dna(12345, 5_000_000)
R = synth.illumina_reads(T, n)
reads, pos = R["reads"], R["start"]
out = []


def run(name, q, t, mode, task="distance"):
    b = edlib_amd.PairBatch(q, t, mode=mode, task=task)
    b.run()
    t0 = time.perf_counter(); b.results_flat(copy=False); first_ms = (time.perf_counter() - t0) * 1e3   # (pays the pinned blocks of the session)
    st = b.run()
    t0 = time.perf_counter(); got = b.results_flat(copy=False); results_ms = (time.perf_counter() - t0) * 1e3
    b.run()
    t0 = time.perf_counter(); got = b.results_flat(); copy_ms = (time.perf_counter() - t0) * 1e3
    b.close()
    nq, mq = q.shape; nt, mt = t.shape
    sel = np.arange(0, nq, max(1, nq // 256), dtype=np.int32)
    ref = O.pool_align(q.reshape(-1), np.arange(nq + 1, dtype=np.int64) * mq, t.reshape(-1),
                       np.arange(nt + 1, dtype=np.int64) * mt, False, mode, task, -1, select=sel)
    ok = bool(np.array_equal(got["editDistance"][sel], ref["editDistance"]))
    if task != "distance" and ok:
        first = lambda x, off, idx: np.array([x[off[i]] if off[i + 1] > off[i] else -9 for i in idx])
        ok = bool(np.array_equal(first(got["starts"], got["locOff"], sel), first(ref["starts"], ref["locOff"], range(len(sel)))))
    if task == "path" and ok:
        ao, ro = got["alnOff"], ref["alnOff"]
        ok = all(np.array_equal(got["alignment"][ao[i]:ao[i + 1]], ref["alignment"][ro[j]:ro[j + 1]]) for j, i in enumerate(sel))
    out.append({"case": name, "pairs": nq, "run_ms": round(st["run_ms"], 2), "results_ms": round(results_ms, 2),
                "results_copy_ms": round(copy_ms, 2), "results_first_call_ms": round(first_ms, 2),
                "results_over_run": round(results_ms / st["run_ms"], 2),
                "gcups": round(st["cells"] / st["run_ms"] / 1e6, 1),
                "us_per_1k_pairs": round(st["run_ms"] * 1e3 / (nq / 1000.0), 1),
                "us_per_1k_pairs_results": round(results_ms * 1e3 / (nq / 1000.0), 1), "sample_ok": ok})


start = np.clip(np.asarray(pos, dtype=np.int64) - 125, 0, len(T) - 400)
win = T[start[:, None] + np.arange(400)[None, :]]
run("150 bp HW in its 400 bp window", reads, np.ascontiguousarray(win), "HW")
run("150 bp HW in its 400 bp window, locations", reads, np.ascontiguousarray(win), "HW", "locations")
run("150 bp HW in its 400 bp window, path", reads, np.ascontiguousarray(win), "HW", "path")
mates = T[np.clip(np.asarray(pos, dtype=np.int64), 0, len(T) - 150)[:, None] + np.arange(150)[None, :]]
run("150 bp NW vs the 150 bp it came from", reads, np.ascontiguousarray(mates), "NW")
run("100 x 100 NW", np.ascontiguousarray(reads[:, :100]), np.ascontiguousarray(mates[:, :100]), "NW")
run("150 bp NW path", reads, np.ascontiguousarray(mates), "NW", "path")
print(json.dumps(out, indent=1))
