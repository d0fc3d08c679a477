"""What the k ladder of kernel A decides for 150-base reads at a given error rate (EDLIB_AMD_DEBUG lines `ladder` /
`level` on stderr) and what the run costs: python tools/ladder_probe.py [sub ins del] [n]"""
import os, sys, json
os.environ["EDLIB_AMD_DEBUG"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np
import edlib_amd
from edlib_amd import synth
sub, ins, dele = [float(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (0.045, 0.0025, 0.0025)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
T = synth.random_dna(12345, 5_000_000)
R = synth.illumina_reads(T, n, m=150, sub=sub, ins=ins, dele=dele)["reads"]
b = edlib_amd.SharedBatch(R, T, mode="HW", task="distance")
b.run(); st = b.run(); f = b.results_flat(); b.close()
ed = f["editDistance"]
print(json.dumps({"n": n, "run_ms": round(st["run_ms"], 2), "scan_ms": round(st["scan_ms"], 2), "word_steps": st["word_steps"],
                  "words_per_read_column": round(st["word_steps"] / (n * 5e6), 3),
                  "ed_quantiles": [int(x) for x in np.quantile(ed, [0.5, 0.9, 0.95, 0.99])]}))
