#!/usr/bin/env python3
"""Config-2-shaped batch over a target with MORE than four symbols (N runs, soft-masked lower case) next to the
same batch over plain ACGT: what leaving the 4-symbol layout costs the reads-per-lane kernels (8 Peq rows per
word in LDS: 4 waves per SIMD instead of 8; no plain kernel for pass 2)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edlib_amd
from edlib_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=262144)
ap.add_argument("--target-len", type=int, default=5_000_000)
a = ap.parse_args()
out = {}
for name, target in (("ACGT", synth.random_dna(12345, a.target_len)),
                     ("ACGT+N", synth.masked_genome(12345, a.target_len, frac_n=0.005, frac_lower=0.0)),
                     ("ACGT+N+lower", synth.masked_genome(12345, a.target_len, frac_n=0.005, frac_lower=0.05))):
    rd = synth.illumina_reads(target, a.reads, m=150, seed=12346)["reads"]        # reads drawn from THAT target
    b = edlib_amd.SharedBatch(rd, target, mode="HW", task="distance", k=-1)
    b.run(); st = b.run()
    ed = b.results_flat()["editDistance"]
    b.close()
    out[name] = {"symbols": int(len(set(target.tolist()))), "run_ms": round(st["run_ms"], 1), "scan_ms": round(st["scan_ms"], 1),
                 "gcups": round(st["cells"] / st["run_ms"] / 1e6, 1), "mean_ed": round(float(ed.mean()), 2)}
out["slowdown_8_rows"] = round(out["ACGT+N"]["run_ms"] / out["ACGT"]["run_ms"], 3)
out["slowdown_16_rows"] = round(out["ACGT+N+lower"]["run_ms"] / out["ACGT"]["run_ms"], 3)
print(json.dumps(out))
