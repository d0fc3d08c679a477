#!/usr/bin/env python3
"""A handful of UNRELATED long HW queries against a 5 Mb target (nothing for the piece filter to find: every row of every
column is needed, and too few queries for the chained strips): kernel W with the target cut into segments, queries above 64
blocks as pipelined strips on the wide kernel.  One JSON object: ms per batch call, the reference on one core per query."""
import sys, time, json, os
sys.path.insert(0, os.getcwd())
import numpy as np, edlib_amd
from edlib_amd import synth
from oracle.oracle import load_ref, load_oracle
t = synth.random_dna(77, 5_000_000)
ref = load_ref() or load_oracle()
out = {}
for m, n in ((1000, 1), (3000, 4), (10000, 1), (10000, 8)):
    qs = [synth.random_dna(900 + 7 * i + m, m) for i in range(n)]
    edlib_amd.align_batch(qs, t, mode="HW", task="distance", raw=True)
    t0 = time.perf_counter(); got = edlib_amd.align_batch(qs, t, mode="HW", task="distance", raw=True); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); want = ref.align(qs[0].tobytes(), t.tobytes(), "HW", "distance", -1); dr = time.perf_counter() - t0
    out["%d x %d bp unrelated" % (n, m)] = {"gpu_ms": round(dt * 1e3, 1), "ok": got[0]["editDistance"] == want["editDistance"] and got[0]["endLocations"] == want["endLocations"], "reference_one_core_ms_per_query": round(dr * 1e3, 1)}
print(json.dumps(out))
