import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import edlib_amd
from edlib_amd import synth
n = int(sys.argv[1]); tn = int(sys.argv[2]); task = sys.argv[3]
T = synth.random_dna(12345, tn)
R = synth.illumina_reads(T, n)
b = edlib_amd.SharedBatch(R["reads"], T, mode="HW", task=task)
st = b.run(); print("ok", n, tn, task, round(st["run_ms"],1), flush=True)
b.close()
