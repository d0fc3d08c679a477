import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, edlib_amd
from edlib_amd import synth
t = synth.random_dna(1, 5_000_000); rd = synth.illumina_reads(t, 8, m=150, seed=2, sub=0.03)["reads"]
tb = t.tobytes()
for i in range(3): edlib_amd.align_raw(rd[i].tobytes(), tb, "HW", "distance", -1)
os.environ["X"]="1"
t0=time.perf_counter(); r = edlib_amd.align_raw(rd[4].tobytes(), tb, "HW", "distance", -1); print("call ms", (time.perf_counter()-t0)*1e3, r["editDistance"])
