#!/usr/bin/env python3
"""Config 5's batch (10,000 1 kb NW pairs) with TASK_DISTANCE and TASK_PATH: what the column store costs the scan."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import edlib_amd
from edlib_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Q, T = synth.mutated_pairs(n, 1000, 12345 + 5, 0.03, 0.01, 0.01, workers=8)
for task in ("distance", "path"):
    b = edlib_amd.PairBatch(Q, T, mode="NW", task=task)
    b.run(); b.run(); t0 = time.perf_counter(); st = b.run(); dt = time.perf_counter() - t0
    print(json.dumps({"task": task, "pairs": n, "wall_ms": round(dt * 1e3, 3), "run_ms": round(st["run_ms"], 3),
                      "scan_ms": round(st["scan_ms"], 3), "launches": st["scan_launches"]}))
    b.close()
