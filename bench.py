#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|4|5] [--strong] ...
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`):
  --config 2 (default, the headline; configs[1] at N=1, configs[2] sharded at N>1): per GPU 1,000,000 synthetic
      150 bp reads (1 % sub, 0.05 % ins, 0.05 % del, 5 % unrelated) in HW mode, k = -1, TASK_DISTANCE, against
      one 5,000,000-base uniform ACGT target.
  --config 4 (configs[3]): per GPU 100,000 ONT-like 10 kb pairs (4 % sub / ins / del), NW, TASK_DISTANCE.
  --config 5 (configs[4]): per GPU 10,000 1 kb pairs (3 % sub, 1 % ins, 1 % del), NW, TASK_PATH; both CIGAR
      formats of every op string are part of the parity check.
A "step" is one pass of the device path over the whole resident batch (edlibAmdBatchRun: target encoding /
buildPeq, every scan pass, merges; for PATH the storing scan + traceback).  Inputs are resident in HBM before the
timed region.  Config 5 is a PATH workload -- its product is the op strings and their CIGARs ON THE HOST -- so its
step is run + collection: edlibAmdBatchRun, then edlibAmdBatchResultsView (the results laid out on the device, one
block copied to pinned host memory) and edlibAmdBatchCigarView for both formats, all inside the timed region; `value`
and `ms_per_step` are that whole.  For the distance workloads (configs 2, 4) a step ends with the results resident
in HBM and `collection` reports what bringing them over costs, once, beside it.

--gpus N: one process per GPU.  Under torch.distributed.run (RANK set) this process is one rank; without it and
with N > 1 bench.py re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.  It refuses to
run when N differs from the world size, and `n_gpus` in the line is the number of ranks that ran (each on its
own device unless --share-gpu, the 1-GPU dry run).  Weak scaling by default (the per-GPU batch is fixed);
--strong cuts ONE global batch into contiguous shards (edlib_amd.parallel.shard_range).  No collective on the
data path; RCCL carries the barrier and the MAX / SUM of time and cells.

Rank 0 prints ONE JSON line: metric = GCUPS = sum(queryLen * targetLen) / s / 1e9 over all ranks.
`roofline` prices the dominant scan kernel against HBM with the ALGORITHMIC bytes of SURVEY.md §8d (each pair
as if it streamed its own target); `valu_roofline` is the bound that actually binds this integer path.
`cpu_baseline` is the unmodified reference (oracle/_ref) driven by a native std::thread pool (oracle/ref_pool.cpp)
on all host cores over a bounded sample of the same batch, which doubles as the bit-exact parity check;
`e2e` is one call of the one-shot C entry point with host buffers in and malloc'd EdlibAlignResults out.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9        # CUs x SIMDs x lanes/clk x Hz (MI355X_MICROARCH.md)
VALU_OPS_PER_WORD_STEP = 10.0               # DP ops per 32-row word-column (7 logic + add + 2 shift), from the gfx950 ISA
VALU_OPS_PER_WORD_STEP_OF = {4: 12.0}       # config 4's lane-per-pair column: the same 10 + 2 that synthesise Eq from the query's bit planes
CONFIGS = {
    2: dict(units=1_000_000, name="1M x 150bp HW reads vs 5Mb target", mode="HW", task="distance",
            kernel="scan_reads_banded_kernel<5> (+ scan_reads_kernel<5,2> for pass 2)", dtype="u32"),
    4: dict(units=100_000, name="100k x 10kb NW pairs, distance", mode="NW", task="distance",
            kernel="lanepair_scan_kernel<42> (a lane per pair, the band narrows with the scores; scan_pairs_ring_kernel for units it leaves open)", dtype="u32"),
    5: dict(units=10_000, name="10k x 1kb NW pairs, path + CIGAR", mode="NW", task="path",
            kernel="scan_pairs_ring32_kernel<8,true> + traceback32_kernel<32> (+ the collection: flat_write_kernel, cigar_kernel)", dtype="u32"),
    # config 5 sixteen times over: the same recipe with enough pairs to fill the chip (10,000 pairs are 0.6 waves per SIMD:
    # that line measures launches and the link, not the kernels) -- VERDICT r5 item 6
    6: dict(units=160_000, name="160k x 1kb NW pairs, path + CIGAR (config 5 x 16: a chip-filling PATH batch)", mode="NW", task="path",
            kernel="scan_pairs_ring32_kernel<8,true> + traceback32_kernel<32> (+ the collection: flat_write_kernel, cigar_kernel)", dtype="u32"),
}
PATH_CONFIGS = (5, 6)
TARGET_LEN, READ_LEN = 5_000_000, 150


# ------------------------------------------------------------------------------------------ launch

def reexec_under_torchrun(n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------ workloads

def make_workload(cfg_id, units, rank, world, strong):
    """(batch factory, packed host arrays for the checker, bookkeeping)"""
    from edlib_amd import synth
    from edlib_amd.parallel import shard_range
    w = {"config": cfg_id}
    from edlib_amd.parallel import host_cpus
    workers = max(1, min(host_cpus(), 64) // max(1, world))     # the cgroup quota, not os.cpu_count(): 8 ranks share 16 CPUs
    if cfg_id == 2:
        target = synth.random_dna(12345, TARGET_LEN)
        if strong and world > 1:
            lo, hi = shard_range(units, rank, world)
            rd = synth.illumina_reads(target, units, m=READ_LEN, seed=12346)
            rd = {k: v[lo:hi] for k, v in rd.items()}
        else:
            rd = synth.illumina_reads(target, units, m=READ_LEN, seed=12346 + rank)
        reads = np.ascontiguousarray(rd["reads"])
        n = len(reads)
        w.update(reads=reads, rd=rd, target=target, n=n, shared=True,
                 qpool=reads.reshape(-1), qoff=np.arange(n + 1, dtype=np.int64) * READ_LEN,
                 tpool=target, toff=np.array([0, len(target)], dtype=np.int64))
        w["describe"] = ("per GPU: %d x %dbp reads (1%% sub, 0.05%% ins/del, 5%% unrelated), EDLIB_MODE_HW, k=-1, "
                         "EDLIB_TASK_DISTANCE, vs one %d-base uniform ACGT target" % (n, READ_LEN, TARGET_LEN))
    else:
        length, seed, rates = (10000, 12349, (0.04, 0.04, 0.04)) if cfg_id == 4 else (1000, 12350, (0.03, 0.01, 0.01))      # (5 and 6)
        if strong and world > 1:
            lo, hi = shard_range(units, rank, world)
        else:
            lo, hi = rank * units, (rank + 1) * units
        qs, ts = synth.mutated_pairs(hi - lo, length, seed, *rates, workers=workers, first=lo)
        n = len(qs)
        qoff = np.zeros(n + 1, dtype=np.int64); qoff[1:] = np.cumsum([len(q) for q in qs])
        toff = np.arange(n + 1, dtype=np.int64) * length
        w.update(qs=qs, ts=ts, n=n, shared=False, qpool=np.concatenate(qs), qoff=qoff,
                 tpool=np.concatenate(ts), toff=toff)
        w["describe"] = ("per GPU: %d pairs, %d-base uniform ACGT targets, queries = target with %g%% sub / %g%% ins / "
                         "%g%% del, EDLIB_MODE_NW, k=-1, EDLIB_TASK_%s" % (n, length, rates[0] * 100, rates[1] * 100,
                                                                         rates[2] * 100, CONFIGS[cfg_id]["task"].upper()))
    return w


def make_batch(w, device):
    import edlib_amd
    c = CONFIGS[w["config"]]
    if w["config"] == 2:
        return edlib_amd.SharedBatch(w["reads"], w["target"], mode=c["mode"], task=c["task"], k=-1, device=device)
    return edlib_amd.PairBatch(w["qs"], w["ts"], mode=c["mode"], task=c["task"], k=-1, device=device)


# ------------------------------------------------------------------------------------------ checker

def segment_sum(values, starts, counts):
    """sum of values[starts[i] : starts[i] + counts[i]] for every i (empty segments give 0)"""
    cs = np.concatenate([[0], np.cumsum(values.astype(np.int64))])
    return cs[starts + counts] - cs[starts]


def cigars_of(flat):
    """both CIGAR strings of every op string through the product's edlibAlignmentToCigar, concatenated"""
    import edlib_amd
    L = edlib_amd.lib()
    ops, off = flat["alignment"], flat["alnOff"]
    ext, std = [], []
    base = ops.ctypes.data if ops is not None and len(ops) else 0
    for i in range(len(off) - 1):
        n = int(off[i + 1] - off[i])
        for fmt, out in ((1, ext), (0, std)):
            p = L.edlibAlignmentToCigar(C.cast(base + int(off[i]), C.c_char_p), n, fmt)
            out.append(C.string_at(p) if p else b"")
            if p:
                L.libc.free(p)
    return b"".join(ext), b"".join(std)


def cpu_baseline_and_parity(w, flat, sample_target, batch=None):
    """The reference on all host cores over a bounded sample of the batch (native thread pool), compared
    field by field with the GPU results of the same units."""
    from oracle import oracle as O
    c = CONFIGS[w["config"]]
    lib, kind = O.checker_library()
    threads, quota = O.cpu_quota()               # not os.cpu_count(): the box caps the container with a cgroup quota
    n = w["n"]
    want_cigar = c["task"] == "path"

    def run(sel):
        return O.pool_align(w["qpool"], w["qoff"], w["tpool"], w["toff"], w["shared"], c["mode"], c["task"], -1,
                            select=sel, threads=threads, want_cigar=want_cigar, libpath=lib)
    if sample_target >= n:
        sel = np.arange(n, dtype=np.int32)
    else:
        # calibrate on two units per thread, then let the sample take at most ~60 s of wall time
        cal = run(np.linspace(0, n - 1, min(n, 2 * threads)).astype(np.int32))
        per_unit = max(cal["wall_seconds"], 1e-3) / max(1, cal["n"]) * min(cal["n"], threads)   # thread-seconds per unit
        fit = int(60.0 * threads / per_unit)
        cnt = max(min(2000, n), min(sample_target, fit))
        half = cnt // 2                                   # first half + evenly strided half
        sel = np.unique(np.concatenate([np.arange(half), np.linspace(0, n - 1, cnt - half).astype(np.int64)])).astype(np.int32)
    ref = run(sel)
    # ---- parity: every field of EdlibAlignResult
    bad = np.zeros(len(sel), dtype=bool)
    for f in ("status", "editDistance", "numLocations", "alphabetLength"):
        bad |= flat[f][sel] != ref[f]
    gl = flat["locOff"]
    cnts = (gl[1:] - gl[:-1])[sel]
    bad |= cnts != (ref["locOff"][1:] - ref["locOff"][:-1])
    if not bad.any():
        first = np.cumsum(cnts) - cnts
        idx = np.repeat(gl[:-1][sel], cnts) + (np.arange(int(cnts.sum())) - np.repeat(first, cnts))
        same_ends = flat["ends"][idx] == ref["ends"]
        if flat["starts"] is not None:
            same_ends &= flat["starts"][idx] == ref["starts"]
        elif ref["hasStarts"].any():
            same_ends[:] = False
        bad |= segment_sum(~same_ends, first, cnts) > 0
    detail = {}
    if want_cigar:
        ga, ra = flat["alnOff"], ref["alnOff"]
        alen = (ga[1:] - ga[:-1])[sel]
        bad |= alen != (ra[1:] - ra[:-1])
        if len(sel) < n and not bad.any():          # a sample: the op bytes of every sampled unit
            fa, rr = flat["alignment"], ref["alignment"]
            for j, i in enumerate(sel):
                if not np.array_equal(fa[ga[i]:ga[i + 1]], rr[ra[j]:ra[j + 1]]):
                    bad[j] = True
            detail = {"op_bytes_compared": int(ra[-1])}
        if len(sel) == n and not bad.any():
            same = np.array_equal(flat["alignment"], ref["alignment"])
            ext, std = cigars_of(flat)
            detail = {"op_bytes_equal": bool(same), "cigar_extended_equal": ext == ref["cigExt"],
                      "cigar_standard_equal": std == ref["cigStd"], "op_bytes": int(ga[-1])}
            if batch is not None:                   # the batch-wide CIGARs (made on the device) against the reference's as well
                dext, dstd = device_cigars(batch)
                detail["batch_cigar_extended_equal"] = dext == ref["cigExt"]
                detail["batch_cigar_standard_equal"] = dstd == ref["cigStd"]
                detail["cigar_bytes"] = [len(dext), len(dstd)]
                same = same and detail["batch_cigar_extended_equal"] and detail["batch_cigar_standard_equal"]
            if not (same and detail["cigar_extended_equal"] and detail["cigar_standard_equal"]):
                bad[:] = True
    cells = float(np.sum((w["qoff"][1:] - w["qoff"][:-1])[sel].astype(np.float64) *
                         (float(w["toff"][1] - w["toff"][0]) if w["shared"] else (w["toff"][1:] - w["toff"][:-1])[sel])))
    gcups = cells / ref["wall_seconds"] / 1e9
    phys = O.physical_cores()
    base = {"value": round(gcups, 2), "unit": "GCUPS", "cores": ref["threads"], "kind": kind,
            "per_thread": round(gcups / ref["threads"], 3),
            "host": {"logical_cpus": os.cpu_count(), "physical_cores": phys,
                     "cgroup_cpu_quota": quota, "threads_used": ref["threads"]},
            "wall_seconds": round(ref["wall_seconds"], 2),
            "sample": "%d of the batch's %d units (%s), native std::thread pool (oracle/ref_pool.cpp), %d threads "
                      "(= the container's CPU quota), one edlibAlign() per unit, glibc mmap threshold pinned" % (len(sel), n, "the whole batch" if len(sel) == n else
                                                      "first half + evenly strided half", ref["threads"])}
    parity = {"checked": int(len(sel)), "bit_exact": int(len(sel) - bad.sum()),
              "fields": "status, editDistance, numLocations, endLocations, startLocations, alphabetLength" +
                        (", alignment, both CIGAR strings" if want_cigar else "")}
    parity.update(detail)
    # The reference pool pins glibc's mmap / trim thresholds for its own sake (oracle/ref_pool.cpp) and leaves gigabytes of freed
    # heap behind: the harness cleans up after its own checker.  (Config 5's 1.85 ms steps after config 4's leg, first blamed on
    # this heap, were the device's clocks coming back from idle: measure_secondary.)
    try:
        C.CDLL(None).malloc_trim(0)
    except Exception:
        pass
    return base, parity


def invariants_config2(w, flat):
    """whole-batch invariants of SURVEY.md §8d"""
    rd = w["rd"]
    ed = flat["editDistance"]
    planted = ~rd["random"]
    out = {"ed_le_planted_edits": bool(np.all(ed[planted] <= rd["edits"][planted])),
           "ed_le_read_len": bool(np.all((ed >= 0) & (ed <= READ_LEN)))}
    # a read without planted indels is its genome window with `edits` substitutions: when the distance equals
    # that count, the window's last column start + m - 1 must be among the end locations
    who = np.nonzero(planted & (rd["indels"] == 0) & (ed == rd["edits"]))[0]
    loc = flat["locOff"]
    cnts = (loc[1:] - loc[:-1])[who]
    first = np.cumsum(cnts) - cnts
    idx = np.repeat(loc[:-1][who], cnts) + (np.arange(int(cnts.sum())) - np.repeat(first, cnts))
    hit = flat["ends"][idx] == np.repeat(rd["start"][who] + READ_LEN - 1, cnts)
    out["planted_position_in_end_locations"] = bool((segment_sum(hit, first, cnts) > 0).all())
    out["planted_position_reads_checked"] = int(len(who))
    return out


def e2e_config2(w):
    """SURVEY.md §8d metric (i): host buffers in -> EdlibAlignResult[] out through ONE call of the one-shot C
    entry point (upload, target packing, buildPeq, every scan pass, download, one malloc per result array)."""
    import edlib_amd
    L = edlib_amd.lib()
    reads = w["reads"]
    n, m = reads.shape
    ptrs = (reads.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(m)).astype(np.uint64)
    qlen = np.full(n, m, dtype=np.int32)
    tbytes = w["target"].tobytes()
    cfg, keep = edlib_amd._make_config("HW", "distance", -1, None)
    res = (edlib_amd.AlignResult * n)()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        rc = L.edlibAlignBatchSharedTarget(ptrs.ctypes.data_as(C.POINTER(C.c_char_p)), qlen.ctypes.data_as(C.POINTER(C.c_int)),
                                           n, tbytes, len(tbytes), cfg, res)
        dt = time.perf_counter() - t0
        if rc != 0:
            return {"error": edlib_amd.last_error()}
        ed = np.frombuffer(res, dtype=np.uint8).reshape(n, C.sizeof(edlib_amd.AlignResult))[:, 4:8].copy().view(np.int32).ravel()
        L.edlibAmdFreeResults(res, n)
        best = dt if best is None else min(best, dt)
    e2e = {"value": round(n * m * len(tbytes) / best / 1e9, 1), "unit": "GCUPS", "seconds": round(best, 4),
           "what": "edlibAlignBatchSharedTarget(): %d host query pointers + host target in, %d malloc'd EdlibAlignResult "
                   "out; best of 2 calls" % (n, n)}
    # the in-library sharding of the same entry point (one host thread + stream per listed device, contiguous
    # slices, target replicated).  A 1-GPU box lists its device twice: the figure prices the host side of the fan-out
    # (two packs / uploads / marshalling threads), not a second GPU.
    old = os.environ.get("EDLIB_AMD_DEVICES")
    os.environ["EDLIB_AMD_DEVICES"] = "0,0" if edlib_amd.device_count() < 2 else "all"
    try:
        t0 = time.perf_counter()
        rc = L.edlibAlignBatchSharedTarget(ptrs.ctypes.data_as(C.POINTER(C.c_char_p)), qlen.ctypes.data_as(C.POINTER(C.c_int)),
                                           n, tbytes, len(tbytes), cfg, res)
        dts = time.perf_counter() - t0
        if rc == 0:
            ed2 = np.frombuffer(res, dtype=np.uint8).reshape(n, C.sizeof(edlib_amd.AlignResult))[:, 4:8].copy().view(np.int32).ravel()
            L.edlibAmdFreeResults(res, n)
            sharded = {"value": round(n * m * len(tbytes) / dts / 1e9, 1), "unit": "GCUPS", "seconds": round(dts, 4),
                       "EDLIB_AMD_DEVICES": os.environ["EDLIB_AMD_DEVICES"], "devices_visible": edlib_amd.device_count(),
                       "distances_equal_unsharded": bool(np.array_equal(ed, ed2))}
        else:
            sharded = {"error": edlib_amd.last_error()}
    finally:
        if old is None:
            os.environ.pop("EDLIB_AMD_DEVICES", None)
        else:
            os.environ["EDLIB_AMD_DEVICES"] = old
    return e2e, ed, sharded



def device_clocks(device):
    """sclk / mclk / fclk, power and temperature of the device as rocm-smi reports them right now (a 27 %
    box-to-box swing on config 4 could not be attributed because the line carried no clocks); {} when the tool is not there"""
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "power (w)", "socket power", "temperature (sensor junction)", "temperature (sensor memory)", "performance level")):
                keep[k] = v
        return keep
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def traffic_of(cfg_id, units):
    """HBM bytes per step from the rocprofv3 --pmc passes of tools/gpu_visit.sh (profiles/hbm_traffic.json records the
    commit they were taken at); None when there is no entry for this batch size"""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        ent = json.load(open(tpath)).get("configs", {}).get(str(cfg_id))
        if ent and ent.get("units") == units:
            head = None
            try:                                  # (.git does not travel to the GPU box; tools/gpu_visit.sh leaves the commit of the visit)
                head = open(os.path.join(ROOT, ".visit_commit")).read().strip() or None
            except Exception:
                pass
            from edlib_amd.parallel import sources_sha
            here = sources_sha(ROOT)
            return ent.get("bytes_per_step"), {"file": "profiles/hbm_traffic.json", "commit": ent.get("commit"), "head": head,
                                               "taken_at_head": (head is not None and str(ent.get("commit", "")).startswith(head[:7])) if head else None,
                                               "sources_sha": {"measured_on": ent.get("sources_sha"), "running": here},
                                               "sources_identical": (ent.get("sources_sha") == here) if ent.get("sources_sha") else None,
                                               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_visit.sh traffic) at "
                                                       "that commit, not measured inside this run"}
    except Exception:
        pass
    return None, None


def report(cfg_id, w, st, scan_ms, launches, steps, warmup, dt, value, world, scaling):
    """the JSON line of one config from the stats of its timed steps"""
    c = CONFIGS[cfg_id]
    main_scan_ms = scan_ms / steps               # all scan launches of a step (HIP events on the batch's stream)
    algo_bytes = st["algo_bytes"]
    achieved = algo_bytes / (main_scan_ms * 1e-3) / 1e9 if main_scan_ms > 0 else 0.0
    traffic, traffic_src = traffic_of(cfg_id, w["n"])
    ops_per_word = VALU_OPS_PER_WORD_STEP_OF.get(cfg_id, VALU_OPS_PER_WORD_STEP)
    lane_ops = st["word_steps"] * ops_per_word
    valu_achieved = lane_ops / (main_scan_ms * 1e-3) if main_scan_ms > 0 else 0.0
    return {
        "metric": "GCUPS (cell updates/s), %s" % c["name"],
        "value": round(value, 1), "unit": "GCUPS", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": c["dtype"], "data": "synthetic",
        "config": {"workload": w["describe"], "baseline_config": cfg_id, "units_per_gpu": w["n"],
                   "parallelism": "units sharded over %d rank(s), one per GPU, target replicated, no collective"
                                  % world},
        "roofline": {"bound": "hbm", "kernel": c["kernel"],
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_step": algo_bytes,
                     "scan_ms_per_step": round(main_scan_ms, 3),
                     "scan_launches_per_step": launches / steps},
        "valu_roofline": {"bound": "valu-int32", "achieved": round(valu_achieved / 1e12, 2),
                          "peak": round(VALU_LANE_OPS / 1e12, 2), "unit": "T lane-ops/s",
                          "frac": round(valu_achieved / VALU_LANE_OPS, 4),
                          "word_steps_per_step": st["word_steps"],
                          "valu_ops_per_word_step": ops_per_word},
        "overflow_units": st["overflow_units"],
    }


def measure_chromosome(repeat=2):
    """The reference's own long-pair shape (test_data/perf_tests.sh:180-191 "Chromosome, NW": 1 Mb x 1 Mb, seven
    divergences), one edlibAlign() call per pair, distance and path; answers against tests/golden/realdata/expected.json
    (made by the compiled reference: score, location, md5 of the op bytes).  Seconds per call here, the reference's on one
    core of this box for the 99 % pair (the rest of its column: the fixture's, taken in the build container)."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edlib_amd
    from test_realdata import EXP, REAL, chromosome, read_fasta
    from oracle.oracle import load_ref
    t = chromosome()
    pairs = []
    for c in EXP["chromosome"]:
        q = read_fasta(os.path.join(REAL, "chromosome", c["query"]))
        row = {"percent": c["percent"], "editDistance": c["editDistance"]}
        best = None
        for _ in range(repeat):
            t0 = time.perf_counter(); g = edlib_amd.align_raw(q, t, "NW", "distance", -1); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row["gpu_s_distance"] = round(best, 4)
        ok = g["status"] == 0 and g["editDistance"] == c["editDistance"] and g["endLocations"] == c["endLocations"]
        best = None
        for _ in range(repeat):
            t0 = time.perf_counter(); g = edlib_amd.align_raw(q, t, "NW", "path", -1); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row["gpu_s_path"] = round(best, 4)
        ok = ok and g["alignment"] is not None and hashlib.md5(g["alignment"]).hexdigest() == c["ops_md5"] \
            and g["alignmentLength"] == c["alignmentLength"] and g["startLocations"] == c["startLocations"]
        row["bit_exact"] = bool(ok)
        row["reference_s_fixture"] = [c["ref_seconds_distance"], c["ref_seconds_path"]]
        if c["percent"] == 99:
            ref = load_ref()
            if ref is not None:
                t0 = time.perf_counter(); r = ref.align(q, t, "NW", "distance", -1); t1 = time.perf_counter()
                ref.align(q, t, "NW", "path", -1); t2 = time.perf_counter()
                row["reference_s_here"] = [round(t1 - t0, 3), round(t2 - t1, 3)]
                row["bit_exact"] = bool(row["bit_exact"] and r["editDistance"] == c["editDistance"])
        row["gcups_distance"] = round(len(q) * len(t) / row["gpu_s_distance"] / 1e9, 1)
        pairs.append(row)
    return {"workload": "7 x (1 Mb x 1 Mb) NW, one edlibAlign() per pair, distance and path", "pairs": pairs,
            "bit_exact": sum(1 for r in pairs if r["bit_exact"]), "checked": len(pairs)}


def measure_collection(cfg_id, batch, ms_per_step):
    """the collection on its own clock: the results view (+ both CIGAR views for the PATH workload, after one more run --
    the timed steps have already collected theirs)"""
    if cfg_id in PATH_CONFIGS:
        batch.run()
    tc = time.perf_counter()
    flat = batch.results_flat(copy=False)
    t1 = time.perf_counter()
    if cfg_id in PATH_CONFIGS:
        batch.cigars(True, copy=False); batch.cigars(False, copy=False)
    t2 = time.perf_counter()
    note = collection_note((t2 - tc) * 1e3, ms_per_step)
    if cfg_id in PATH_CONFIGS:
        # the link: what one collection brings over PCIe (the dense op bytes dominate) against the time the view took
        nbytes = int(sum(v.nbytes for v in flat.values() if isinstance(v, np.ndarray)))
        note["link"] = {"bytes_per_collection": nbytes, "view_ms": round((t1 - tc) * 1e3, 3),
                        "GBps": round(nbytes / max(t1 - tc, 1e-9) / 1e9, 2),
                        "note": "results view only (device-made arrays in one block, D2H into pinned memory); op bytes travel at one byte per op"}
        note["what"] = ("edlibAmdBatchResultsView (device-made arrays, one block D2H: %.3f ms) + edlibAmdBatchCigarView x 2 (%.3f ms); "
                        "ALREADY INSIDE ms_per_step for this config" % ((t1 - tc) * 1e3, (t2 - t1) * 1e3))
        note["ms_per_step_plus_one_collection"] = ms_per_step
        note["inside_step"] = True
    return note, flat


def collection_note(collect_ms, ms_per_step):
    """A step (`Batch.run()`) ends with the results resident in HBM; bringing them to the host happens in `results_flat()`, once
    per call whatever the number of steps (DISTANCE batches of reads and flat pair batches: the caller-facing arrays are laid
    out by device kernels and come over as one block; other batches: download, per-unit records, flat arrays).  Reported beside
    the step so that the lazily collected paths do not hide that work."""
    return {"results_flat_ms": round(collect_ms, 3), "ms_per_step_plus_one_collection": round(ms_per_step + collect_ms, 3),
            "what": "results_flat(copy=False) after the timed steps: the view of what run() left in HBM (read batches with "
                    "TASK_DISTANCE and flat pair batches: arrays made on the device, one block D2H into pinned memory)"}


def run_step(cfg_id, batch):
    """one timed step: the device pass; for the PATH workload also the collection of everything a caller gets"""
    st = batch.run()
    if cfg_id in PATH_CONFIGS:
        batch.results_flat(copy=False)
        batch.cigars(True, copy=False)
        batch.cigars(False, copy=False)
    return st


def device_cigars(batch):
    """both CIGAR formats of the batch as the concatenation of the strings (no terminators), made by the library over the
    whole batch (edlibAmdBatchCigarView)"""
    out = []
    for extended in (True, False):
        chars, _ = batch.cigars(extended, copy=False)
        out.append(chars[chars != 0].tobytes())
    return out


def measure_secondary(cfg_id, device, torch, steps=5, warmup=2):
    """One more BASELINE config on this rank's device: resident batch, `warmup` + `steps` timed steps between two
    device synchronisations, then the reference over the WHOLE batch (parity of every field + the CPU baseline)."""
    w = make_workload(cfg_id, CONFIGS[cfg_id]["units"], 0, 1, False)
    batch = make_batch(w, device)
    try:
        # The device has idled through this config's workload generation and the leg of the reference before it, and its clocks
        # take tens of milliseconds of work to come back: config 5's steps (0.7 ms) were timed at 0.7 or at 1.8 ms depending on
        # what had run how long before (tools probe of round 6: 13 slow steps, then 0.70 ms for good).  Warm-up and timed region
        # are therefore at least `warmup` / `steps` steps AND at least 0.25 s / 0.1 s long; both counts are in the line.
        tw, nw = time.perf_counter(), 0
        while nw < warmup or (time.perf_counter() - tw < 0.25 and nw < 5000):
            run_step(cfg_id, batch); nw += 1
        warmup = nw
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scan_ms, launches, st, ns = 0.0, 0, None, 0
        while ns < steps or (time.perf_counter() - t0 < 0.1 and ns < 5000):
            st = run_step(cfg_id, batch); ns += 1
            scan_ms += st["scan_ms"]; launches += st["scan_launches"]
        steps = ns
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = report(cfg_id, w, st, scan_ms, launches, steps, warmup, dt, st["cells"] * steps / dt / 1e9, 1, "weak")
        out["collection"], flat = measure_collection(cfg_id, batch, out["ms_per_step"])
        out["cpu_baseline"], out["parity_sample"] = cpu_baseline_and_parity(w, flat, w["n"] if cfg_id != 6 else 10000, batch if cfg_id == 5 else None)
    finally:
        batch.close()
    return out

# ------------------------------------------------------------------------------------------ main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--reads", "--units", dest="units", type=int, default=None,
                    help="units per GPU (default: the BASELINE size of the config); with --strong the TOTAL")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (BASELINE config 3): --units is the TOTAL, split evenly over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs 4 / 5 lines that a default (config 2, full size, 1 GPU) run appends")
    ap.add_argument("--parity-sample", type=int, default=None,
                    help="units checked against the reference (default: 20000 for config 2, the whole batch for 4 / 5)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run on a 1-GPU box: every rank uses device 0")
    ap.add_argument("--dump", default=None, help="rank 0 writes the gathered editDistance of all ranks to this .npy")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(reexec_under_torchrun(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): refusing to report a wrong n_gpus"
                         % (args.gpus, world))
    import torch
    from edlib_amd.parallel import host_cpus
    # the library's host fan-outs (marshalling, packing) share the container's CPU quota with the other ranks
    os.environ.setdefault("EDLIB_AMD_HOST_THREADS", str(max(1, min(6, host_cpus() // max(1, world)))))
    import edlib_amd
    ndev = edlib_amd.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device (%s); there is no CPU path" % edlib_amd.last_error())
    device = 0 if args.share_gpu else local_rank
    if device >= ndev:
        raise SystemExit("bench.py: rank %d needs device %d but only %d device(s) are visible (use --share-gpu for a dry run)"
                         % (rank, device, ndev))
    backend = args.backend or ("gloo" if args.share_gpu else "nccl")
    dist = None
    torch.cuda.set_device(device)
    if "RANK" in os.environ:                     # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    coll_dev = ("cuda" if backend == "nccl" else "cpu") if dist is not None else None

    c = CONFIGS[args.config]
    units = args.units if args.units is not None else c["units"]
    w = make_workload(args.config, units, rank, world, args.strong)
    batch = make_batch(w, device)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step(args.config, batch)
    sync()
    t0 = time.perf_counter()
    scan_ms = 0.0
    launches = 0
    st = batch.stats()
    for _ in range(args.steps):
        st = run_step(args.config, batch)
        scan_ms += st["scan_ms"]
        launches += st["scan_launches"]
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    sync()
    dt_sync = time.perf_counter() - t0
    clocks_after = device_clocks(device) if rank == 0 else None          # (outside the timed region)
    from edlib_amd.parallel import aggregate_throughput, gather_int_results
    # whole-job cells (SUM over ranks) and the slowest rank's time (MAX over ranks, barrier included)
    cells_all, dt = aggregate_throughput(st["cells"] * args.steps, dt_sync, dist, coll_dev)
    per_rank_ms = [round(dt_local / max(1, args.steps) * 1e3, 2)]
    devices = ["%s:%d" % (socket.gethostname(), device)]
    if dist is not None and world > 1:
        tl = [None] * world
        dist.all_gather_object(tl, (per_rank_ms[0], devices[0]))
        per_rank_ms = [x[0] for x in tl]
        devices = [x[1] for x in tl]
    flat = None
    coll_note = None
    # (rank 0 also at world > 1: the line of a multi-rank run carries the reference baseline and the parity check of rank 0's
    # OWN shard -- taken after the timed region, while the other ranks wait at the closing barrier)
    if args.dump or (rank == 0 and not args.no_cpu_baseline):
        coll_note, flat = measure_collection(args.config, batch, round(dt_sync / max(1, args.steps) * 1e3, 2))
    if args.dump:
        total = units if args.strong else units * world
        full = gather_int_results(flat["editDistance"], total, dist, coll_dev)    # shard order = rank order
        if rank == 0:
            np.save(args.dump, full)
    value = cells_all / dt / 1e9
    out = None
    if rank == 0:
        out = report(args.config, w, st, scan_ms, launches, max(1, args.steps), args.warmup, dt, value, world,
                     "strong" if args.strong else "weak")
        if coll_note is not None:
            if not coll_note.get("inside_step"):
                coll_note["ms_per_step_plus_one_collection"] = round(out["ms_per_step"] + coll_note["results_flat_ms"], 3)
            out["collection"] = coll_note
        out["clocks"] = {"right_after_the_timed_steps": clocks_after, "after_the_run": device_clocks(device),
                         "source": "rocm-smi --showclocks --showpower --showtemp (rank 0's device)"}
        out["per_rank_ms_per_step"] = per_rank_ms
        out["devices"] = devices
        out["devices_distinct"] = len(set(devices))
        if args.share_gpu:
            out["dry_run_shared_gpu"] = True
        if not args.no_cpu_baseline:
            sample = args.parity_sample if args.parity_sample is not None else (20000 if args.config == 2 else (10000 if args.config == 6 else w["n"]))
            base, parity = cpu_baseline_and_parity(w, flat, sample, batch if args.config == 5 else None)
            if world > 1:
                base["sample"] += "; rank 0's shard of a %d-rank run (the other ranks idle at the barrier meanwhile)" % world
                parity["shard"] = "rank 0 of %d" % world
            out["cpu_baseline"] = base
            out["parity_sample"] = parity
            if args.config == 2:
                out["invariants"] = invariants_config2(w, flat)
                # the WHOLE batch against the reference's committed answers (tests/golden/c2_full_ref.npz), when this is the
                # batch they were made for (1,000,000 reads, rank 0, weak scaling)
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import full_parity_c2
                    pf = full_parity_c2.compare_with_fixture(flat, w["target"], w["reads"])
                    if pf is not None:
                        out["parity_full"] = pf
                except Exception as e:                                 # the headline line must survive
                    out["parity_full"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and args.config == 2 and not args.no_e2e:
            r = e2e_config2(w)
            if isinstance(r, tuple):
                out["e2e"] = r[0]
                if flat is not None:
                    out["e2e"]["distances_equal_resident"] = bool(np.array_equal(r[1], flat["editDistance"]))
                if len(r) > 2:
                    out["e2e_sharded"] = r[2]
            else:
                out["e2e"] = r
    batch.close()
    del batch, w, flat
    # BASELINE configs 4 and 5 at their full sizes, AFTER the headline's timed region and on the same device: every
    # default run carries a driver-visible line for them (value, rooflines, reference baseline, whole-batch parity)
    if rank == 0 and world == 1 and args.config == 2 and args.units is None and not args.no_secondary and not args.no_cpu_baseline:
        out["secondary"] = {}
        for cid in (4, 5, 6):
            try:
                out["secondary"]["config%s" % ("5x" if cid == 6 else cid)] = measure_secondary(cid, device, torch)
            except Exception as e:                                    # the headline line must survive
                out["secondary"]["config%s" % ("5x" if cid == 6 else cid)] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            out["secondary"]["chromosome"] = measure_chromosome()
        except Exception as e:
            out["secondary"]["chromosome"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
