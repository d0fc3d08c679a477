#!/usr/bin/env python3
"""Known answers for BASELINE.json config 1: the reference's own CLI + library
(oracle/_ref/aligner_ref = /root/reference/apps/aligner/aligner.cpp + edlib/src/edlib.cpp, compiled by
tools/build_ref_clients.sh) on the reference's Phage files.

The eight FASTA files under tests/golden/phage/ are byte-for-byte the data files of
/root/reference/test_data/Enterobacteria_Phage_1/ (test DATA, not source): /root/reference does not exist on the
GPU box, so the fixtures travel with the repo.  This script re-copies them (when the reference is present), runs
`aligner_ref -m NW [-p -f CIG_EXT]` on every pair and writes tests/golden/phage/expected.json:
score, (start, end) location, md5 of the CIGAR line as the CLI prints it (text + '\n', the form SURVEY.md §8c
quotes), and the CIGAR's length.   Run from the repo root:  python oracle/gen_phage_golden.py
"""
import hashlib
import json
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DATA = "/root/reference/test_data/Enterobacteria_Phage_1"
DST = os.path.join(ROOT, "tests", "golden", "phage")
TARGET = "Enterobacteria_phage_1.fasta"
PERCENTS = [60, 70, 80, 90, 94, 97, 99]


def cli(exe, flags, query, target):
    out = subprocess.run([exe] + flags + [query, target], capture_output=True, text=True, timeout=600, check=True).stdout
    return out


def parse(out):
    """(score, [(start, end)], cigar line or None) of the single query of a CLI run."""
    m = re.search(r"^#0: (-?\d+)\s+(\d+)\s+\[(.*)\]", out, re.M)
    score = locs = None
    if m:
        score = int(m.group(1))
        locs = [(None if a == "?" else int(a), int(b)) for a, b in re.findall(r"\((\?|-?\d+), (-?\d+)\)", m.group(3))]
    m2 = re.search(r"score = (-?\d+)", out)
    if m2:
        score = int(m2.group(1))
    cig = None
    m3 = re.search(r"^Cigar:\n(.*)$", out, re.M)
    if m3:
        cig = m3.group(1)
    return score, locs, cig


def main():
    os.makedirs(DST, exist_ok=True)
    if os.path.isdir(REF_DATA):
        for name in [TARGET] + ["mutated_%d_perc.fasta" % p for p in PERCENTS]:
            shutil.copyfile(os.path.join(REF_DATA, name), os.path.join(DST, name))
    exe = os.path.join(ROOT, "oracle", "_ref", "aligner_ref")
    cases = []
    for p in PERCENTS:
        q, t = os.path.join(DST, "mutated_%d_perc.fasta" % p), os.path.join(DST, TARGET)
        score, locs, _ = parse(cli(exe, ["-m", "NW", "-l"], q, t))
        score2, _, cig = parse(cli(exe, ["-m", "NW", "-p", "-f", "CIG_EXT"], q, t))
        _, _, cigs = parse(cli(exe, ["-m", "NW", "-p", "-f", "CIG_STD"], q, t))
        assert score == score2
        cases.append({"percent": p, "query": "mutated_%d_perc.fasta" % p, "target": TARGET, "score": score,
                      "locations": locs,
                      "cigar_ext_md5": hashlib.md5((cig + "\n").encode()).hexdigest(), "cigar_ext_len": len(cig),
                      "cigar_std_md5": hashlib.md5((cigs + "\n").encode()).hexdigest(), "cigar_std_len": len(cigs)})
        print(cases[-1])
    with open(os.path.join(DST, "expected.json"), "w") as f:
        json.dump({"generator": "oracle/gen_phage_golden.py", "cases": cases}, f, indent=1)


if __name__ == "__main__":
    main()
