"""HW reads of 150 .. 10,000 bases against 5 Mb: rate per read length across the reads-per-lane groups (up to 256
bases) and the piece filter + window verification above them (edlib_amd/csrc/long_reads.hip); 16,384 Illumina-like
reads per length (1 % substitutions, 0.05 % insertions / deletions, 5 % unrelated), plus 150-base reads at 5 % error
(the k ladder of kernel A) and ONT-like 10 kb reads.  One JSON object.  A strided sample of every batch is checked
against the reference (test infrastructure) so that the rates are rates of correct results.
  --lengths 150,256,...   --n 16384   --sample 32"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import edlib_amd
from edlib_amd import synth
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--lengths", default="150,256,257,300,384,512,513,768,1024,1025,1500,2048,3000,4096,6000,8192,10000")
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--sample", type=int, default=24, help="reads per batch checked against the reference")
ap.add_argument("--no-extra", action="store_true")
ap.add_argument("--masked", action="store_true", help="ACGT + runs of N (five symbols: the eight-row Peq layout)")
args = ap.parse_args()

T = synth.masked_genome(12345, 5_000_000, frac_lower=0.0) if args.masked else synth.random_dna(12345, 5_000_000)
toff = np.array([0, len(T)], dtype=np.int64)
out = {}


def run(tag, R, n, m):
    b = edlib_amd.SharedBatch(R, T, mode="HW", task="distance")
    b.run(); st = b.run(); got = b.results_flat(); b.close()
    sel = np.linspace(0, n - 1, min(n, args.sample)).astype(np.int32)
    ref = O.pool_align(R.reshape(-1), np.arange(n + 1, dtype=np.int64) * m, T, toff, True, "HW", "distance", -1, select=sel)
    gl = got["locOff"]
    ok = bool(np.array_equal(got["editDistance"][sel], ref["editDistance"]) and
              np.array_equal(np.concatenate([got["ends"][gl[i]:gl[i + 1]] for i in sel]), ref["ends"]))
    out[tag] = {"run_ms": round(st["run_ms"], 1), "gcups": round(st["cells"] / st["run_ms"] / 1e6, 1),
                "word_steps": st["word_steps"], "path": st["path"], "sample_ok": ok,
                "ref_core_seconds_per_read": round(ref["wall_seconds"] * ref["threads"] / len(sel), 4)}
    print(tag, out[tag], file=sys.stderr, flush=True)


for m in [int(x) for x in args.lengths.split(",")]:
    n = args.n if m <= 4096 else max(1024, args.n * 4096 // m)       # the same bases per batch above 4096
    run("%d bp" % m, synth.illumina_reads(T, n, m=m)["reads"], n, m)
if not args.no_extra:
    n = args.n
    run("150 bp, 5 % error", synth.illumina_reads(T, n, m=150, sub=0.045, ins=0.0025, dele=0.0025)["reads"], n, 150)
    run("150 bp, 1 % error (again, same n)", synth.illumina_reads(T, n, m=150)["reads"], n, 150)
    n, m = 2048, 10000
    starts = (synth.rand_u64(777, n, 1) % np.uint64(len(T) - 12000)).astype(np.int64)
    R = np.stack([synth.mutate(T[s:s + 11500], 778, 0.04, 0.04, 0.04, stream=i)[0][:m] for i, s in enumerate(starts)])
    run("10000 bp, ONT-like 12 % error", np.ascontiguousarray(R), n, m)
print(json.dumps(out))
