#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

Workload (BASELINE.json configs[1]): per GPU, 1,000,000
synthetic 150 bp reads (1 % sub, 0.05 % ins, 0.05 % del, 5 % unrelated reads) in
HW (infix) mode, k = -1, TASK_DISTANCE, against one 5,000,000-base uniform ACGT
target.  A "step" is one pass of the device path over the whole resident batch:
target encoding, buildPeq, the scan kernel, the segment merge and (when a read
has more end locations than the first pass keeps) the exact second pass.
Inputs are resident in HBM before the timed region; results stay on the device.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  metric = GCUPS = sum(queryLen*targetLen)/s/1e9 over
all ranks (weak scaling: every rank owns its own 1M reads, no collective on the
data path).  `roofline` prices the dominant kernel (scan_reads_banded_kernel<5>, all its launches of a step)
against HBM with the ALGORITHMIC bytes of SURVEY.md §8d (each pair counted as if it
streamed its own target); `valu_roofline` is the bound that actually binds this
integer kernel.  `cpu_baseline` times the unmodified reference
(oracle/_ref) on the host cores on a bounded sample and doubles as a parity check.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TARGET_LEN = 5_000_000
READ_LEN = 150
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes/clk x Hz (MI355X_MICROARCH.md)
VALU_OPS_PER_WORD_STEP = 10.0  # DP ops per 32-row word-column (7 logic + add + 2 shift), counted from gfx950 ISA


def cpu_baseline(reads, target, gpu_results, seconds_budget=20.0):
    """Reference edlib (oracle/_ref) on the host cores, one thread per core, on a
    bounded sample of the same reads; also the bit-exact parity check of that sample."""
    from oracle.oracle import load_oracle, load_ref
    impl = load_ref()
    kind = "reference"
    if impl is None:
        impl = load_oracle()
        kind = "port"
    cores = os.cpu_count() or 1
    tbytes = target.tobytes()
    # calibrate on one read, then size the sample to ~seconds_budget of total CPU work
    t0 = time.perf_counter()
    impl.align(reads[0].tobytes(), tbytes, "HW", "distance", -1)
    per_read = max(time.perf_counter() - t0, 1e-4)
    n = int(max(cores, min(len(reads), seconds_budget / per_read)))
    n = (n // cores) * cores or cores
    idx = np.linspace(0, len(reads) - 1, n).astype(np.int64)
    out = [None] * n

    def work(k):
        for j in range(k, n, cores):
            out[j] = impl.align(reads[idx[j]].tobytes(), tbytes, "HW", "distance", -1)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    ok = 0
    for j in range(n):
        i = idx[j]
        w = out[j]
        if (gpu_results["editDistance"][i] == w["editDistance"]
                and gpu_results["ends"][i].tolist() == (w["endLocations"] or [])
                and gpu_results["numLocations"][i] == w["numLocations"]
                and gpu_results["alphabetLength"][i] == w["alphabetLength"]):
            ok += 1
    gcups = n * READ_LEN * len(target) / dt / 1e9
    return ({"value": round(gcups, 2), "unit": "GCUPS", "cores": cores, "kind": kind,
             "sample": "%d of the batch's reads (evenly strided), full %d-base target, %d threads, %.1f s wall"
                       % (n, len(target), cores, dt)},
            {"checked": n, "bit_exact": ok})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (default: the BASELINE config)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (BASELINE config 3): --reads is the TOTAL, split evenly over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a dry run)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank uses device 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if args.share_gpu:
        local_rank = 0
    if world > 1 or "RANK" in os.environ:        # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    import edlib_amd
    from edlib_amd import synth

    # synthetic inputs: same target everywhere, each rank its own reads (weak scaling: --reads per rank;
    # --strong: the same global batch cut into contiguous shards, edlib_amd.parallel.shard_range)
    target = synth.random_dna(12345, TARGET_LEN)
    if args.strong and world > 1:
        from edlib_amd.parallel import shard_range
        lo, hi = shard_range(args.reads, rank, world)
        rd = synth.illumina_reads(target, args.reads, m=READ_LEN, seed=12346)
        rd = {k: v[lo:hi] for k, v in rd.items()}
        args.reads = hi - lo
    else:
        rd = synth.illumina_reads(target, args.reads, m=READ_LEN, seed=12346 + rank)
    batch = edlib_amd.SharedBatch(rd["reads"], target, mode="HW", task="distance", k=-1, device=local_rank)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    sync()
    t0 = time.perf_counter()
    scan_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        st = batch.run()
        scan_ms += st["scan_ms"]
        launches += st["scan_launches"]
    sync()
    dt = time.perf_counter() - t0
    from edlib_amd.parallel import aggregate_throughput
    # whole-job cells (SUM over ranks) and the slowest rank's time (MAX over ranks)
    cells_all, dt = aggregate_throughput(st["cells"] * args.steps, dt, dist,
                                         ("cuda" if args.backend == "nccl" else "cpu") if dist is not None else None)
    value = cells_all / dt / 1e9
    out = None
    if rank == 0:
        # dominant kernel: the first scan launch of a step covers the whole batch
        main_scan_ms = scan_ms / args.steps          # all scan launches of a step (incl. exact pass)
        algo_bytes = st["algo_bytes"]
        achieved = algo_bytes / (main_scan_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        lane_ops = st["word_steps"] * VALU_OPS_PER_WORD_STEP
        valu_achieved = lane_ops / (main_scan_ms * 1e-3)
        out = {
            "metric": "GCUPS (cell updates/s), 1M x 150bp HW reads vs 5Mb target",
            "value": round(value, 1), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "per GPU: %d x %dbp reads (1%% sub, 0.05%% ins/del, 5%% unrelated), "
                                   "EDLIB_MODE_HW, k=-1, EDLIB_TASK_DISTANCE, vs one %d-base uniform ACGT target"
                                   % (args.reads, READ_LEN, TARGET_LEN),
                       "reads_per_gpu": args.reads, "read_len": READ_LEN, "target_len": TARGET_LEN,
                       "parallelism": "reads sharded over %d GPU(s), target replicated, no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "scan_reads_banded_kernel<5>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_step": algo_bytes,
                         "scan_ms_per_step": round(main_scan_ms, 2),
                         "scan_launches_per_step": launches / args.steps},
            "valu_roofline": {"bound": "valu-int32", "achieved": round(valu_achieved / 1e12, 2),
                              "peak": round(VALU_LANE_OPS / 1e12, 2), "unit": "T lane-ops/s",
                              "frac": round(valu_achieved / VALU_LANE_OPS, 4),
                              "word_steps_per_step": st["word_steps"],
                              "valu_ops_per_word_step": VALU_OPS_PER_WORD_STEP},
            "overflow_units": st["overflow_units"],
        }
        if world == 1 and not args.no_cpu_baseline:
            res = batch.results_arrays()
            base, parity = cpu_baseline(rd["reads"], target, res)
            out["cpu_baseline"] = base
            out["parity_sample"] = parity
            # whole-batch invariants: ed <= planted edits, ed <= read length
            ed = res["editDistance"]
            planted = ~rd["random"]
            out["invariants"] = {
                "ed_le_planted_edits": bool(np.all(ed[planted] <= rd["edits"][planted])),
                "ed_le_read_len": bool(np.all((ed >= 0) & (ed <= READ_LEN))),
            }
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
