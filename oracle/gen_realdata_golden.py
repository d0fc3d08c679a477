#!/usr/bin/env python3
"""Known answers on the reference's own REAL-DATA shapes (SURVEY.md §8c "[probe] config-1 goldens", the shapes of
/root/reference/test_data/perf_tests.sh:150-191), made by the compiled, unmodified reference (oracle/_ref/libedlib_ref.so).

  chromosome/   the seven "Chromosome, NW" pairs of perf_tests.sh:180-191: mutated_{60..99}_perc.fasta against
                Chromosome_2890043_3890042_0.fasta (1,000,000 x ~1,000,000 bases): NW score, location, md5 of the op
                bytes and of both CIGAR lines (TASK_PATH: the Hirschberg regime, edlib.cpp:1231-1396), and the
                reference's own seconds on one core of this container.
  mason/        every file of test_data/E_coli_DH1/mason_illumina_reads/{50,100,250,500}bp and 10kbp as ONE
                shared-target HW batch per directory against the 1 Mb chromosome (perf_tests.sh:153-163 names
                e_coli_DH1.fasta, which is a missing blob: .MISSING_LARGE_BLOBS), tasks distance / locations / path.
                Contains the §8(c) golden: the 250 bp read, HW -l => 108, (350889,351126) (350889,351127).
  prefixes/     test_data/E_coli_DH1/prefixes/*: SHW against the same chromosome (perf_tests.sh:167-177).
  myers         the FIXED-k block of perf_tests.sh:195-219 ("Myers": HW on the 10 kbp reads with k = 100 / 1000 / 10000,
                the same (file, k) selection), plus k = distance and distance - 1 for every file: distance and
                locations.  (Against the 1 Mb chromosome these reads are unrelated sequence, distance ~4,880: k = 100 and
                1000 answer -1 -- the band of a fixed k that never reaches the bottom row -- and k = 10000 finds them.)

The FASTA files are the reference's test DATA (not source); /root/reference does not exist on the GPU box, so they travel
as fixtures: the 1 Mb files xz-compressed (tests read them with lzma), the read files as they are.
Run from the repo root:   python oracle/gen_realdata_golden.py [--only chromosome|mason|prefixes|myers]
"""
import argparse
import hashlib
import json
import lzma
import os
import shutil
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/test_data"
DST = os.path.join(ROOT, "tests", "golden", "realdata")
CHROM_DIR = "Chromosome_2890043_3890042_0"
CHROM = "Chromosome_2890043_3890042_0.fasta"
PERCENTS = [99, 97, 94, 90, 80, 70, 60]
MASON = ["50bp", "100bp", "250bp", "500bp", "10kbp"]


def read_fasta(path):
    """first record of a FASTA file (plain or .xz) as bytes: apps/aligner/aligner.cpp:290-328."""
    op = lzma.open if path.endswith(".xz") else open
    seq = []
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if seq:
                    break
                continue
            seq.append(line.strip())
    return b"".join(seq)


def md5(b):
    return hashlib.md5(b).hexdigest()


def summarise(ref, r, with_ops):
    out = {"editDistance": r["editDistance"], "numLocations": r["numLocations"], "alphabetLength": r["alphabetLength"],
           "endLocations": r["endLocations"], "startLocations": r["startLocations"],
           "alignmentLength": r["alignmentLength"]}
    if with_ops and r["alignment"] is not None:
        ops = r["alignment"]
        out["ops_md5"] = md5(ops)
        ext, std = ref.cigar(ops, 1), ref.cigar(ops, 0)
        # the form the CLI prints (text + newline): what SURVEY.md §8c's md5s are taken over
        out["cigar_ext_md5"] = md5((ext + "\n").encode()); out["cigar_ext_len"] = len(ext)
        out["cigar_std_md5"] = md5((std + "\n").encode()); out["cigar_std_len"] = len(std)
        if len(ext) <= 4096:
            out["cigar_ext"] = ext
    return out


def chrom_job(p):
    from oracle.oracle import load_ref
    ref = load_ref()
    q = read_fasta(os.path.join(REF, CHROM_DIR, "mutated_%d_perc.fasta" % p))
    t = read_fasta(os.path.join(REF, CHROM_DIR, CHROM))
    t0 = time.time(); d = ref.align(q, t, "NW", "distance", -1); t1 = time.time()
    pr = ref.align(q, t, "NW", "path", -1); t2 = time.time()
    assert d["editDistance"] == pr["editDistance"]
    case = {"percent": p, "query": "mutated_%d_perc.fasta.xz" % p, "target": CHROM + ".xz", "qlen": len(q), "tlen": len(t),
            "ref_seconds_distance": round(t1 - t0, 3), "ref_seconds_path": round(t2 - t1, 3)}
    case.update(summarise(ref, pr, True))
    print(case, flush=True)
    return case


def batch_job(args):
    kind, sub, mode = args
    from oracle.oracle import load_ref
    ref = load_ref()
    t = read_fasta(os.path.join(REF, CHROM_DIR, CHROM))
    src = os.path.join(REF, "E_coli_DH1", kind, sub)
    files = sorted(os.listdir(src))
    cases = []
    for name in files:
        q = read_fasta(os.path.join(src, name))
        c = {"file": name, "qlen": len(q)}
        t0 = time.time()
        for task in ("distance", "locations", "path"):
            c[task] = summarise(ref, ref.align(q, t, mode, task, -1), task == "path")
        c["ref_seconds"] = round(time.time() - t0, 3)
        cases.append(c)
    print(kind, sub, mode, [(c["file"], c["distance"]["editDistance"], c["distance"]["numLocations"]) for c in cases], flush=True)
    return {"dir": "%s/%s" % (kind, sub), "mode": mode, "cases": cases}


MYERS_K = {100: ["e_coli_DH1_illumina_1x10000.fasta"],
           1000: ["e_coli_DH1_illumina_1x10000.fasta", "mutated_97_perc.fasta", "mutated_94_perc.fasta", "mutated_90_perc.fasta"],
           10000: None}                                   # None: every file of the directory (perf_tests.sh:214-219)


def myers_job(name):
    from oracle.oracle import load_ref
    ref = load_ref()
    t = read_fasta(os.path.join(REF, CHROM_DIR, CHROM))
    q = read_fasta(os.path.join(REF, "E_coli_DH1", "mason_illumina_reads", "10kbp", name))
    d = ref.align(q, t, "HW", "distance", -1)["editDistance"]
    ks = sorted(set([k for k, files in MYERS_K.items() if files is None or name in files] + [d, d - 1]))
    out = []
    for k in ks:
        c = {"file": name, "qlen": len(q), "k": k}
        for task in ("distance", "locations"):
            c[task] = summarise(ref, ref.align(q, t, "HW", task, k), False)
        out.append(c)
    print("myers", name, [(c["k"], c["distance"]["editDistance"]) for c in out], flush=True)
    return out


def myers_related_job(p):
    """the same three thresholds on RELATED 10 kb reads: bases [500000, 510000) of mutated_<p>_perc.fasta (the reference's
    Chromosome files, sliced here because its E. coli genome is a missing blob), HW against the chromosome"""
    from oracle.oracle import load_ref
    ref = load_ref()
    t = read_fasta(os.path.join(REF, CHROM_DIR, CHROM))
    q = read_fasta(os.path.join(REF, CHROM_DIR, "mutated_%d_perc.fasta" % p))[500000:510000]
    out = []
    for k in (100, 1000, 10000):
        c = {"slice_of": "mutated_%d_perc.fasta.xz" % p, "from": 500000, "to": 510000, "qlen": len(q), "k": k}
        for task in ("distance", "locations"):
            c[task] = summarise(ref, ref.align(q, t, "HW", task, k), False)
        out.append(c)
    print("myers_related", p, [(c["k"], c["distance"]["editDistance"], c["locations"]["startLocations"]) for c in out], flush=True)
    return out


def copy_data():
    os.makedirs(os.path.join(DST, "chromosome"), exist_ok=True)
    for name in [CHROM] + ["mutated_%d_perc.fasta" % p for p in PERCENTS]:
        dst = os.path.join(DST, "chromosome", name + ".xz")
        if not os.path.exists(dst):
            with open(os.path.join(REF, CHROM_DIR, name), "rb") as f, lzma.open(dst, "wb", preset=9) as g:
                g.write(f.read())
    for kind, subs in (("mason_illumina_reads", MASON), ("prefixes", MASON)):
        for sub in subs:
            d = os.path.join(DST, kind, sub)
            os.makedirs(d, exist_ok=True)
            for name in os.listdir(os.path.join(REF, "E_coli_DH1", kind, sub)):
                shutil.copyfile(os.path.join(REF, "E_coli_DH1", kind, sub, name), os.path.join(d, name))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--workers", type=int, default=4)
    a = ap.parse_args()
    copy_data()
    path = os.path.join(DST, "expected.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["generator"] = "oracle/gen_realdata_golden.py"
    with ProcessPoolExecutor(a.workers) as ex:
        futs = {}
        if a.only in ("", "mason"):
            futs["mason"] = [ex.submit(batch_job, ("mason_illumina_reads", s, "HW")) for s in MASON]
        if a.only in ("", "prefixes"):
            futs["prefixes"] = [ex.submit(batch_job, ("prefixes", s, "SHW")) for s in MASON]
        if a.only in ("", "chromosome"):
            futs["chromosome"] = [ex.submit(chrom_job, p) for p in PERCENTS]
        if a.only in ("", "myers"):
            futs["myers"] = [ex.submit(myers_job, n) for n in sorted(os.listdir(os.path.join(REF, "E_coli_DH1", "mason_illumina_reads", "10kbp")))]
            futs["myers_related"] = [ex.submit(myers_related_job, p) for p in (99, 97, 94, 90)]
        for k, fs in futs.items():
            doc[k] = [f.result() for f in fs]
            if k in ("myers", "myers_related"):
                doc[k] = [c for part in doc[k] for c in part]
            with open(path, "w") as f:
                json.dump(doc, f, indent=1)
    # the golden SURVEY.md §8c quotes
    for b in doc.get("mason", []):
        if b["dir"].endswith("250bp"):
            c = [c for c in b["cases"] if c["file"] == "e_coli_DH1_illumina_1x250.fasta"][0]["locations"]
            print("250 bp read, HW -l:", c["editDistance"], list(zip(c["startLocations"], c["endLocations"])))


if __name__ == "__main__":
    main()
