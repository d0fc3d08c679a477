import sys, os, time, json
sys.path.insert(0, os.getcwd())
import edlib_amd
from edlib_amd import synth
T = synth.random_dna(12345, 5_000_000)
out = {}
for m, n in ((256, 16384), (300, 16384)):
    R = synth.illumina_reads(T, n, m=m)["reads"]
    b = edlib_amd.SharedBatch(R, T, mode="HW", task="distance")
    b.run(); st = b.run(); b.close()
    out["%d x %d bp" % (n, m)] = {"run_ms": round(st["run_ms"], 1), "gcups": round(st["cells"] / st["run_ms"] / 1e6, 1), "path": st["path"]}
print(json.dumps(out))
