import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import edlib_amd
from edlib_amd import synth
n = 262144
T = synth.random_dna(12345, 5_000_000)
R = synth.illumina_reads(T, n)
reads, pos = R["reads"], R["start"]
start = np.clip(np.asarray(pos, dtype=np.int64) - 125, 0, len(T) - 400)
win = np.ascontiguousarray(T[start[:, None] + np.arange(400)[None, :]])
mates = np.ascontiguousarray(T[np.clip(np.asarray(pos, dtype=np.int64), 0, len(T) - 150)[:, None] + np.arange(150)[None, :]])
which = sys.argv[1]
q, t, mode = (reads, win, "HW") if which == "hw" else (reads, mates, "NW")
b = edlib_amd.PairBatch(q, t, mode=mode, task="distance")
b.run()
sys.stderr.write("==== second run\n")
st = b.run()
sys.stderr.write(json.dumps(st) + "\n")
b.close()
