import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|MHz' | head -8")
import bench
from oracle import oracle as O
w = bench.make_workload(2, 4096, 0, 1, False)
for thr in (1, 8, 16, 32, 64, 128, 256):
    n = max(8, thr * 2)
    sel = np.arange(n, dtype=np.int32)
    r = O.pool_align(w["qpool"], w["qoff"], w["tpool"], w["toff"], True, "HW", "distance", select=sel, threads=thr)
    g = n * 150 * 5e6 / r["wall_seconds"] / 1e9
    print("threads %3d: %6.2f s  %8.1f GCUPS  %6.2f per thread" % (thr, r["wall_seconds"], g, g / thr), flush=True)
