#!/usr/bin/env python3
"""How much do LOC / PATH cost on top of DISTANCE for the config-2 batch? (phases 2 and 3 run on the pair kernel)"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import edlib_amd
from edlib_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = synth.random_dna(12345, 5_000_000)
R = synth.illumina_reads(T, n)
for task in ("distance", "locations", "path"):
    b = edlib_amd.SharedBatch(R["reads"], T, mode="HW", task=task)
    b.run(); t0 = time.perf_counter(); st = b.run(); dt = time.perf_counter() - t0
    print(json.dumps({"task": task, "reads": n, "wall_ms": round(dt * 1e3, 1), "run_ms": round(st["run_ms"], 1),
                      "scan_ms": round(st["scan_ms"], 1), "launches": st["scan_launches"]}))
    b.close()
