"""HW reads of 150 .. 1200 bases against 5 Mb: where the reads-per-lane kernels end (1024 bases) and what the step
costs.  16,384 Illumina-like reads per length; one JSON object.  A strided sample of every batch is checked against
the oracle (test infrastructure) so that the rates are rates of correct results."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import edlib_amd
from edlib_amd import synth
from oracle import oracle as O

T = synth.random_dna(12345, 5_000_000)
out = {}
n = 16384
for m in (150, 256, 257, 300, 384, 385, 450, 512, 513, 600, 768, 769, 1024, 1025, 1200):
    R = synth.illumina_reads(T, n, m=m)["reads"]
    b = edlib_amd.SharedBatch(R, T, mode="HW", task="distance")
    b.run(); st = b.run(); got = b.results_flat(); b.close()
    sel = np.arange(0, n, 512, dtype=np.int32)
    ref = O.pool_align(R.reshape(-1), np.arange(n + 1, dtype=np.int64) * m, T, np.array([0, len(T)], dtype=np.int64),
                       True, "HW", "distance", -1, select=sel)
    ok = bool(np.array_equal(got["editDistance"][sel], ref["editDistance"]))
    out["%d bp" % m] = {"run_ms": round(st["run_ms"], 1), "gcups": round(st["cells"] / st["run_ms"] / 1e6, 1),
                        "path": st["path"], "sample_ok": ok}
print(json.dumps(out))
