import sys, numpy as np
sys.path.insert(0, ".")
import edlib_amd
from edlib_amd import synth
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
rng = np.random.default_rng(21)
qs, ts = [], []
for i in range(2500):
    t = _ACGT[rng.integers(0, 4, 400)]
    a = int(rng.integers(0, 400 - 150 + 1))
    q, _ = synth.mutate(t[a:a + 150], int(rng.integers(1 << 30)), 0.03, 0.01, 0.01)
    qs.append(np.ascontiguousarray(q)); ts.append(t)
mode, task = sys.argv[1], sys.argv[2]
b = edlib_amd.PairBatch(qs, ts, mode=mode, task=task)
print("created", flush=True)
b.run(); print("run ok", flush=True)
r = b.results_flat(); print("results ok", r["editDistance"][:5], flush=True)
