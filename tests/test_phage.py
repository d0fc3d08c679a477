"""BASELINE.json config 1: `edlib-aligner -m NW` on the reference's Enterobacteria phage P1 pairs
(94,481-base genome vs seven mutated copies, 1477 blocks).  The FASTA files are the
reference's own test data (tests/golden/phage/, copied by oracle/gen_phage_golden.py); expected.json holds
what the reference's CLI + library printed for them: score, location, md5 of the -p CIGAR line.

CPU part: the oracle restatement (and oracle/_ref where it travelled) reproduces the scores, so the fixtures
are pinned from both sides.  GPU part: the batch CLI, the reference's unmodified CLI linked to libedlib.so,
and the Python batch API all have to print / return exactly that -- distance by k-doubling on the lane rings,
the path through Hirschberg levels (reference apps/aligner/aligner.cpp:162-225, edlib.cpp:1231-1396)."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHAGE = os.path.join(ROOT, "tests", "golden", "phage")
with open(os.path.join(PHAGE, "expected.json")) as _f:
    CASES = json.load(_f)["cases"]
SURVEY_SCORES = {99: 990, 97: 2977, 94: 6042, 90: 9506, 80: 20333, 70: 30147, 60: 39829}
SURVEY_MD5 = {99: "0ae812d343f815d96c63da4bee0f7230", 90: "bde28046ba560850464380cd62fa713a",
              60: "ee6b5c761a4cd5a19c0b9bf2ecd079e0"}


def read_fasta(path):
    """first record of a FASTA file as bytes (the CLI's reader: apps/aligner/aligner.cpp:290-328)."""
    seq = []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if seq:
                    break
                continue
            seq.append(line.strip())
    return b"".join(seq)


def test_fixture_matches_survey():
    assert {c["percent"]: c["score"] for c in CASES} == SURVEY_SCORES
    for c in CASES:
        assert c["locations"] == [[0, 94480]]
        if c["percent"] in SURVEY_MD5:
            assert c["cigar_ext_md5"] == SURVEY_MD5[c["percent"]]
    assert len(read_fasta(os.path.join(PHAGE, CASES[0]["target"]))) == 94481


@pytest.mark.parametrize("case", CASES, ids=lambda c: "p%d" % c["percent"])
def test_oracle_scores(oracle, ref, case):
    q = read_fasta(os.path.join(PHAGE, case["query"]))
    t = read_fasta(os.path.join(PHAGE, case["target"]))
    for impl in (oracle, ref):
        if impl is None:
            continue
        got = impl.align(q, t, "NW", "distance", -1)
        assert got["editDistance"] == case["score"]
        assert got["endLocations"] == [94480] and got["numLocations"] == 1


# ------------------------------------------------------------------ GPU

def _cli(exe, flags, case):
    out = subprocess.run([exe] + flags + [os.path.join(PHAGE, case["query"]), os.path.join(PHAGE, case["target"])],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-800:] + out.stderr[-800:]
    return out.stdout


def _check_cli(exe, case):
    out = _cli(exe, ["-m", "NW", "-l"], case)
    m = re.search(r"^#0: (-?\d+)\s+(\d+)\s+\[ \((\d+), (\d+)\) \]", out, re.M)
    assert m, out[-800:]
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))) == (case["score"], 1, 0, 94480)
    for fmt, key in (("CIG_EXT", "cigar_ext"), ("CIG_STD", "cigar_std")):
        out = _cli(exe, ["-m", "NW", "-p", "-f", fmt], case)
        assert re.search(r"score = %d\b" % case["score"], out), out[:800]
        m = re.search(r"^Cigar:\n(.*)$", out, re.M)
        assert m, out[:800]
        assert len(m.group(1)) == case[key + "_len"]
        assert hashlib.md5((m.group(1) + "\n").encode()).hexdigest() == case[key + "_md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "p%d" % c["percent"])
def test_batch_cli_on_phage(case):
    exe = os.path.join(ROOT, "build", "edlib-aligner-batch")
    assert os.path.exists(exe), "build/edlib-aligner-batch missing: run make"
    _check_cli(exe, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c["percent"] in (60, 90, 99)], ids=lambda c: "p%d" % c["percent"])
def test_reference_cli_linked_to_this_library_on_phage(case):
    exe = os.path.join(ROOT, "oracle", "_ref", "aligner_amd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/aligner_amd was not prebuilt (needs /root/reference at build time)")
    _check_cli(exe, case)


@pytest.mark.gpu
def test_all_seven_pairs_as_one_batch(engine):
    """the seven pairs as ONE pair batch (Hirschberg levels of different jobs run together), TASK_PATH"""
    t = read_fasta(os.path.join(PHAGE, CASES[0]["target"]))
    qs = [read_fasta(os.path.join(PHAGE, c["query"])) for c in CASES]
    res = engine.align_pairs([np.frombuffer(q, dtype=np.uint8) for q in qs],
                             [np.frombuffer(t, dtype=np.uint8)] * len(qs), mode="NW", task="path", raw=True)
    for c, r in zip(CASES, res):
        assert r["editDistance"] == c["score"] and r["endLocations"] == [94480] and r["startLocations"] == [0]
        for ext, key in ((True, "cigar_ext"), (False, "cigar_std")):
            cig = engine.cigar_from_alignment(r["alignment"], extended=ext)
            assert len(cig) == c[key + "_len"]
            assert hashlib.md5((cig + "\n").encode()).hexdigest() == c[key + "_md5"]
