"""Multi-GPU sharding of a batch: the units of a batch are independent,
so each rank (one process per GPU) owns a contiguous slice, the shared target is
replicated, and there is NO collective on the data path.  The only communication is
bookkeeping: a barrier around the timed region, MAX over ranks of the elapsed time,
SUM of the work, and (optionally) a gather of per-unit results to rank 0.  The
backend is whatever torch.distributed was initialised with: "nccl" (= RCCL over xGMI)
on the GPU box, "gloo" in the CPU tests.
"""
import os

import numpy as np


def host_cpus():
    """CPUs this process may really use: the cgroup CPU quota (cpu.max, or cfs_quota_us on cgroup v1) capped by the
    affinity mask.  The GPU boxes show 256 logical CPUs behind a 16-CPU quota; worker pools sized from
    os.cpu_count() there only measure the throttle, and 8 ranks doing so start 64 threads each on 16 CPUs."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def shard_range(n_units, rank, world):
    """Contiguous slice [lo, hi) of rank `rank`: ceil(n/world) units each, last ranks may get fewer."""
    per = (n_units + world - 1) // world
    lo = min(n_units, rank * per)
    return lo, min(n_units, lo + per)


def aggregate_throughput(cells_local, seconds_local, dist=None, device=None):
    """(sum of cells over ranks, max of seconds over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(cells_local), float(seconds_local)
    import torch
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    c = torch.tensor([float(cells_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return int(c.item()), float(t.item())


def gather_int_results(local, n_units, dist=None, device=None):
    """Concatenate per-rank int32 result arrays (in shard order) on every rank."""
    local = np.asarray(local, dtype=np.int32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    world = dist.get_world_size()
    per = (n_units + world - 1) // world
    buf = torch.full((per,), -2, dtype=torch.int32, device=device)
    buf[:len(local)] = torch.from_numpy(local).to(buf.device)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    full = torch.cat(outs).cpu().numpy()
    keep = []
    for r in range(world):
        lo, hi = shard_range(n_units, r, world)
        keep.append(full[r * per: r * per + (hi - lo)])
    return np.concatenate(keep)


def sources_sha(root):
    """sha256 over the product's sources and bench.py (names and contents, sorted): profiles/hbm_traffic.json records it with
    the counters, bench.py recomputes it at run time -- `sources_identical` says whether the counters were taken on the very
    code that is running, whatever commits of documents and profiles lie in between."""
    import glob
    import hashlib
    import os
    h = hashlib.sha256()
    files = [os.path.join(root, "bench.py"), os.path.join(root, "Makefile")]
    for pat in ("edlib_amd/csrc/*", "edlib_amd/*.py", "include/*.h"):
        files += glob.glob(os.path.join(root, pat))
    for f in sorted(set(files)):
        if not os.path.isfile(f):
            continue
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
