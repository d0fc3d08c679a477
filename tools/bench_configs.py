#!/usr/bin/env python3
"""Timing + parity of BASELINE.json configs 4 and 5 (parity-test cases, not the bench line):
   config 4: N x 10 kb ONT-like NW pairs (4/4/4 %), TASK_DISTANCE        (default N = 10000)
   config 5: N x 1 kb NW pairs (3/1/1 %), TASK_PATH + CIGAR               (default N = 10000)
Prints one JSON line per config: GCUPS on the GPU (resident batch, HIP-event time), the reference on
the host cores on a bounded sample, and the number of sampled units that are bit-identical."""
import argparse, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import edlib_amd
from edlib_amd import synth
from oracle.oracle import load_ref, load_oracle


def cpu_sample(impl, qs, ts, mode, task, idx):
    cores = os.cpu_count() or 1
    out = [None] * len(idx)
    def work(k):
        for j in range(k, len(idx), cores):
            out[j] = impl.align(qs[idx[j]].tobytes(), ts[idx[j]].tobytes(), mode, task, -1)
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
    return out, time.perf_counter() - t0, cores


def run(name, qs, ts, mode, task, sample):
    impl = load_ref() or load_oracle()
    b = edlib_amd.PairBatch(qs, ts, mode=mode, task=task)
    b.run()
    t0 = time.perf_counter(); st = b.run(); wall = time.perf_counter() - t0
    res = b.results(raw=True)
    b.close()
    idx = np.linspace(0, len(qs) - 1, sample).astype(int)
    ref, dt, cores = cpu_sample(impl, qs, ts, mode, task, idx)
    ok = sum(1 for j, i in enumerate(idx) if all(res[i][f] == ref[j][f] for f in
             ("editDistance", "endLocations", "startLocations", "alignment", "alphabetLength")))
    cells_s = sum(len(qs[i]) * len(ts[i]) for i in idx)
    print(json.dumps({"config": name, "units": len(qs), "gpu_gcups_wall": round(st["cells"] / wall / 1e9, 1),
                      "gpu_gcups_scan": round(st["cells"] / st["scan_ms"] / 1e6, 1), "run_ms": round(st["run_ms"], 1),
                      "scan_ms": round(st["scan_ms"], 1), "scan_launches": st["scan_launches"],
                      "cpu_reference_gcups": round(cells_s / dt / 1e9, 1), "cpu_threads": cores,
                      "parity_sample": {"checked": len(idx), "bit_exact": ok}}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n4", type=int, default=10000)
    ap.add_argument("--n5", type=int, default=10000)
    a = ap.parse_args()
    if a.n4:
        qs, ts = synth.mutated_pairs(a.n4, 10000, seed=12349, sub=0.04, ins=0.04, dele=0.04)
        run("4: %d x 10kb NW distance" % a.n4, qs, ts, "NW", "distance", 256)
    if a.n5:
        qs, ts = synth.mutated_pairs(a.n5, 1000, seed=12350, sub=0.03, ins=0.01, dele=0.01)
        run("5: %d x 1kb NW path" % a.n5, qs, ts, "NW", "path", 512)
