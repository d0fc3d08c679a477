#!/usr/bin/env python3
"""PATH in the Hirschberg regime (column store >= 1 MiB, edlib.cpp:1188-1211): N x 10 kb ONT-like NW pairs
and a few 94 kb pairs.  One JSON line each: run_ms on the GPU (resident batch), the reference on the host
cores on a sample, bit-exact count of the sample."""
import argparse, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import edlib_amd
from edlib_amd import synth
from oracle.oracle import load_ref, load_oracle


def run(name, qs, ts, sample):
    impl = load_ref() or load_oracle()
    b = edlib_amd.PairBatch(qs, ts, mode="NW", task="path")
    b.run()
    st = b.run()
    res = b.results(raw=True)
    b.close()
    idx = np.linspace(0, len(qs) - 1, sample).astype(int)
    cores = min(os.cpu_count() or 1, len(idx))
    out = [None] * len(idx)
    def work(k):
        for j in range(k, len(idx), cores):
            out[j] = impl.align(qs[idx[j]].tobytes(), ts[idx[j]].tobytes(), "NW", "path", -1)
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    ok = sum(1 for j, i in enumerate(idx) if out[j]["status"] == 2 or all(res[i][f] == out[j][f] for f in
             ("editDistance", "endLocations", "startLocations", "alignment", "alphabetLength")))
    print(json.dumps({"case": name, "units": len(qs), "run_ms": round(st["run_ms"], 1), "scan_ms": round(st["scan_ms"], 1),
                      "scan_launches": st["scan_launches"], "ms_per_unit": round(st["run_ms"] / len(qs), 4),
                      "cpu_reference_ms_per_unit_per_thread": round(dt * 1e3 * cores / len(idx), 2), "cpu_threads": cores,
                      "parity_sample": {"checked": len(idx), "bit_exact": ok}}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n10k", type=int, default=2000)
    ap.add_argument("--n94k", type=int, default=8)
    a = ap.parse_args()
    if a.n10k:
        qs, ts = synth.mutated_pairs(a.n10k, 10000, seed=12351, sub=0.04, ins=0.04, dele=0.04)
        run("%d x 10kb NW path (12%% edits)" % a.n10k, qs, ts, 64)
    if a.n94k:
        qs, ts = synth.mutated_pairs(a.n94k, 94481, seed=12352, sub=0.01, ins=0.005, dele=0.005)
        run("%d x 94kb NW path (2%% edits)" % a.n94k, qs, ts, min(8, a.n94k))
