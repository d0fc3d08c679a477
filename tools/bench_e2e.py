#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the headline workload -- SURVEY.md 8(d) metric (i).

Times ONE call of the one-shot C entry point edlibAlignBatchSharedTarget(): inputs are plain host
buffers (a pointer per read), outputs are malloc'd EdlibAlignResult members on the host, so the timed
region holds upload, target packing, buildPeq, every scan pass, download and result marshalling.
Building the pointer array and reading the results back into numpy happen outside the timed region
(a C caller already has the former and does not need the latter).  Prints one JSON line.

The numbers are cross-checked against the resident-session path (bench.py's path) on the same reads.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edlib_amd                      # noqa: E402
from edlib_amd import synth           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--target-len", type=int, default=5000000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--task", default="distance", choices=["distance", "locations", "path"])
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()

    L = edlib_amd.lib()
    target = synth.random_dna(12345, args.target_len)
    rd = synth.illumina_reads(target, args.reads, m=args.read_len, seed=12346)
    reads = np.ascontiguousarray(rd["reads"])
    n, m = reads.shape
    base = reads.ctypes.data
    ptrs = (base + np.arange(n, dtype=np.uint64) * np.uint64(m)).astype(np.uint64)
    qarr = ptrs.ctypes.data_as(C.POINTER(C.c_char_p))
    qlen = np.full(n, m, dtype=np.int32)
    tbytes = target.tobytes()
    cfg, keep = edlib_amd._make_config("HW", args.task, -1, None)
    res = (edlib_amd.AlignResult * n)()

    fn = L.edlibAlignBatchSharedTarget
    times = []
    ed = None
    for rep in range(args.repeat + 1):                 # first call is the warm-up (pool, code objects)
        t0 = time.perf_counter()
        rc = fn(qarr, qlen.ctypes.data_as(C.POINTER(C.c_int)), n, tbytes, len(tbytes), cfg, res)
        t1 = time.perf_counter()
        if rc != 0:
            raise SystemExit("one-shot batch failed: " + edlib_amd.last_error())
        if rep:
            times.append(t1 - t0)
        view = np.frombuffer(res, dtype=np.uint8).reshape(n, C.sizeof(edlib_amd.AlignResult))
        ed = view[:, 4:8].copy().view(np.int32).ravel()
        t2 = time.perf_counter()
        for i in range(n):
            L.edlibFreeAlignResult(res[i])
        t_free = time.perf_counter() - t2

    # same reads through the resident session (what bench.py times)
    b = edlib_amd.SharedBatch(reads, target, mode="HW", task=args.task, k=-1)
    b.run(); b.run()
    st = b.stats()
    arr = (edlib_amd.AlignResult * n)()
    t3 = time.perf_counter()
    if L.edlibAmdBatchResults(b._h, arr) != 0:
        raise SystemExit("results failed: " + edlib_amd.last_error())
    t_results = time.perf_counter() - t3
    ed2 = np.frombuffer(arr, dtype=np.uint8).reshape(n, C.sizeof(edlib_amd.AlignResult))[:, 4:8].copy().view(np.int32).ravel()
    for i in range(n):
        L.edlibFreeAlignResult(arr[i])
    b.close()

    cells = float(n) * m * len(tbytes)
    best = min(times)
    print(json.dumps({
        "metric": "end-to-end GCUPS, host buffers in -> EdlibAlignResult[] out (one-shot C entry point)",
        "reads": n, "read_len": m, "target_len": len(tbytes), "task": args.task,
        "seconds": [round(t, 4) for t in times],
        "end_to_end_gcups": round(cells / best / 1e9, 1),
        "resident_run_ms": round(st["run_ms"], 2),
        "resident_gcups": round(cells / (st["run_ms"] * 1e-3) / 1e9, 1),
        "overhead_ms": round(best * 1e3 - st["run_ms"], 2),
        "resident_results_ms": round(t_results * 1e3, 2),
        "free_results_python_loop_s": round(t_free, 3),
        "distances_equal": bool(np.array_equal(ed, ed2)),
    }))


if __name__ == "__main__":
    main()
