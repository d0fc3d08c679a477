#!/usr/bin/env python3
"""Config 2's batch (N x 150 bp HW reads vs the 5 Mb target) with TASK_DISTANCE, TASK_LOC and TASK_PATH: what start locations
and paths of a shared-target read batch cost beside the scan.  EDLIB_AMD_DEBUG=1 prints the laps of each run."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import edlib_amd
from edlib_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
target = synth.random_dna(12345, 5_000_000)
w = {"reads": np.ascontiguousarray(synth.illumina_reads(target, n, m=150, seed=12346)["reads"]), "target": target}
base = None
for task in ("distance", "locations", "path"):
    b = edlib_amd.SharedBatch(w["reads"], w["target"], mode="HW", task=task, k=-1)
    b.run(); b.run()            # (the records of a batch are recycled from its third run on)
    if os.environ.get("EDLIB_AMD_DEBUG"): sys.stderr.write("---- %s\n" % task)
    t0 = time.perf_counter(); st = b.run(); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); v = b.results_flat() if hasattr(b, "results_flat") else None; dv = time.perf_counter() - t1
    if base is None: base = dt
    print(json.dumps({"task": task, "reads": n, "wall_ms": round(dt * 1e3, 2), "over_distance": round(dt / base, 4), "run_ms": round(st["run_ms"], 2),
                      "scan_ms": round(st["scan_ms"], 2), "launches": st["scan_launches"], "view_ms": round(dv * 1e3, 2)}))
    b.close()
