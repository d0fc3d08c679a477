#!/usr/bin/env python3
"""The reference's flagship long-pair shape (test_data/perf_tests.sh:180-191, "Chromosome, NW"): the seven mutated copies
of a 1 Mb chromosome against it, NW, distance (-k -1) and path (-p).  One JSON line per pair: seconds per edlibAlign()
call of this library on the GPU (best of `--repeat`), the reference's seconds on one core (from the fixture -- build
container -- and, with --ref, measured on this box), and whether score / location / op bytes / both CIGAR md5s are the
fixture's (tests/golden/realdata/expected.json, made by the compiled reference).
    python tools/bench_chromosome.py [--percents 99,97,90] [--no-path] [--ref] [--repeat 2]"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import edlib_amd  # noqa: E402
from test_realdata import EXP, REAL, chromosome, read_fasta  # noqa: E402


def md5(b):
    return hashlib.md5(b).hexdigest()


def timed(fn, repeat):
    best, out = None, None
    for _ in range(repeat):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--percents", default="99,97,94,90,80,70,60")
    ap.add_argument("--no-path", action="store_true")
    ap.add_argument("--ref", action="store_true", help="also time the compiled reference on this box (one core)")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--batch", action="store_true", help="also: all selected pairs as ONE resident pair batch (distance, path)")
    a = ap.parse_args()
    want = [int(x) for x in a.percents.split(",")]
    t = chromosome()
    ref = None
    if a.ref:
        from oracle.oracle import load_ref
        ref = load_ref()
    edlib_amd.align_raw(b"ACGT", b"ACGT", "NW", "distance", -1)          # context, pools
    for c in EXP["chromosome"]:
        if c["percent"] not in want:
            continue
        q = read_fasta(os.path.join(REAL, "chromosome", c["query"]))
        line = {"pair": "mutated_%d_perc vs Chromosome_2890043_3890042_0" % c["percent"], "qlen": len(q), "tlen": len(t),
                "editDistance": c["editDistance"],
                "reference_s_fixture": {"distance": c["ref_seconds_distance"], "path": c["ref_seconds_path"]}}
        dt, got = timed(lambda: edlib_amd.align_raw(q, t, "NW", "distance", -1), a.repeat)
        line["gpu_s_distance"] = round(dt, 4)
        line["distance_ok"] = bool(got["status"] == 0 and got["editDistance"] == c["editDistance"] and got["endLocations"] == c["endLocations"])
        line["gcups_distance"] = round(len(q) * len(t) / dt / 1e9, 1)
        if not a.no_path:
            dt, got = timed(lambda: edlib_amd.align_raw(q, t, "NW", "path", -1), a.repeat)
            line["gpu_s_path"] = round(dt, 4)
            ok = got["status"] == 0 and got["alignment"] is not None and md5(got["alignment"]) == c["ops_md5"]
            if ok:
                ext = edlib_amd.cigar_from_alignment(got["alignment"], True)
                std = edlib_amd.cigar_from_alignment(got["alignment"], False)
                ok = md5((ext + "\n").encode()) == c["cigar_ext_md5"] and md5((std + "\n").encode()) == c["cigar_std_md5"]
            line["path_ok"] = bool(ok and got["startLocations"] == c["startLocations"] and got["alignmentLength"] == c["alignmentLength"])
        if ref is not None:
            t0 = time.perf_counter(); r = ref.align(q, t, "NW", "distance", -1); t1 = time.perf_counter()
            line["reference_s_here"] = {"distance": round(t1 - t0, 3)}
            assert r["editDistance"] == c["editDistance"]
            if not a.no_path and c["percent"] >= 90:
                t0 = time.perf_counter(); ref.align(q, t, "NW", "path", -1); line["reference_s_here"]["path"] = round(time.perf_counter() - t0, 3)
        print(json.dumps(line), flush=True)
    if a.batch:
        sel = [c for c in EXP["chromosome"] if c["percent"] in want]
        qs = [read_fasta(os.path.join(REAL, "chromosome", c["query"])) for c in sel]
        for task in (("distance",) if a.no_path else ("distance", "path")):
            b = edlib_amd.PairBatch(qs, [t] * len(qs), mode="NW", task=task)
            b.run(); st = b.run()
            res = b.results(raw=True)
            b.close()
            ok = all(r["editDistance"] == c["editDistance"] and (task == "distance" or md5(r["alignment"]) == c["ops_md5"]) for r, c in zip(res, sel))
            print(json.dumps({"batch_of": len(sel), "task": task, "run_ms": round(st["run_ms"], 2), "scan_ms": round(st["scan_ms"], 2),
                              "gcups": round(st["cells"] / st["run_ms"] / 1e6, 1), "ok": bool(ok)}), flush=True)


if __name__ == "__main__":
    main()
