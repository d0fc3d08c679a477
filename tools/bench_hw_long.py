#!/usr/bin/env python3
"""A few long HW queries against a long target (the reference CLI's shape: `edlib-aligner -m HW reads.fa chr.fa`):
kernel W with and without the target segmentation (EDLIB_AMD_HWSEG=0), the reference on one core beside it."""
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
    res = {}
    for seg in ("1", "0"):
        env = dict(os.environ, EDLIB_AMD_HWSEG=seg)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=env, timeout=1200)
        res["segmented" if seg == "1" else "one wave per query (EDLIB_AMD_HWSEG=0)"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else p.stderr[-400:]
    print(json.dumps(res))
