#!/usr/bin/env python3
"""Batches of SHORT independent pairs (the verification step of a seed-and-extend mapper): n reads of 150 bases, each
against its own window -- HW against a 400-base window around its origin, NW against its 150-base mutated mate, and
100 x 100 NW.  Resident run time, GCUPS, and a strided sample checked against the oracle."""
import sys, os, json
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
    b.run(); st = b.run(); got = b.results_flat(); b.close()
    nq, mq = q.shape; nt, mt = t.shape
    sel = np.arange(0, nq, max(1, nq // 256), dtype=np.int32)
    ref = O.pool_align(q.reshape(-1), np.arange(nq + 1, dtype=np.int64) * mq, t.reshape(-1),
                       np.arange(nt + 1, dtype=np.int64) * mt, False, mode, task, -1, select=sel)
    ok = bool(np.array_equal(got["editDistance"][sel], ref["editDistance"]))
    out.append({"case": name, "pairs": nq, "run_ms": round(st["run_ms"], 2), "gcups": round(st["cells"] / st["run_ms"] / 1e6, 1),
                "us_per_1k_pairs": round(st["run_ms"] * 1e3 / (nq / 1000.0), 1), "sample_ok": ok})


start = np.clip(np.asarray(pos, dtype=np.int64) - 125, 0, len(T) - 400)
win = T[start[:, None] + np.arange(400)[None, :]]
run("150 bp HW in its 400 bp window", reads, np.ascontiguousarray(win), "HW")
run("150 bp HW in its 400 bp window, locations", reads[: n // 4], np.ascontiguousarray(win[: n // 4]), "HW", "locations")
mates = T[np.clip(np.asarray(pos, dtype=np.int64), 0, len(T) - 150)[:, None] + np.arange(150)[None, :]]
run("150 bp NW vs the 150 bp it came from", reads, np.ascontiguousarray(mates), "NW")
run("100 x 100 NW", np.ascontiguousarray(reads[:, :100]), np.ascontiguousarray(mates[:, :100]), "NW")
run("150 bp NW path", reads[: n // 4], np.ascontiguousarray(mates[: n // 4]), "NW", "path")
print(json.dumps(out, indent=1))
