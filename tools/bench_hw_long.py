#!/usr/bin/env python3
"""A few long HW queries against a long target (the reference CLI's shape: `edlib-aligner -m HW reads.fa chr.fa`):
the engine (piece filter, window verification, target segments for what the filter hands back), the reference on one
core beside it."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, edlib_amd
    from edlib_amd import synth
    from oracle.oracle import load_ref, load_oracle
    t = synth.random_dna(77, 5_000_000)
    out = {}
    for m, n in ((1000, 1), (1000, 16), (3000, 4)):
        qs = []
        for i in range(n):
            at = 100_000 + i * 300_000
            q, _ = synth.mutate(t[at:at + m], 78 + i, 0.03, 0.01, 0.01)
            qs.append(q)
        edlib_amd.align_batch(qs, t, mode="HW", task="locations", raw=True)
        t0 = time.perf_counter(); got = edlib_amd.align_batch(qs, t, mode="HW", task="locations", raw=True); dt = time.perf_counter() - t0
        ref = load_ref() or load_oracle()
        t0 = time.perf_counter(); want = ref.align(qs[0].tobytes(), t.tobytes(), "HW", "locations", -1); dr = time.perf_counter() - t0
        assert got[0] == want
        out["%d x %d bp" % (n, m)] = {"gpu_ms": round(dt * 1e3, 1), "reference_one_core_ms_per_query": round(dr * 1e3, 1)}
    print(json.dumps(out))
else:
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, timeout=1200)
    print(p.stdout.strip().splitlines()[-1] if p.returncode == 0 else json.dumps({"error": p.stderr[-400:]}))
