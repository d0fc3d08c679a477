"""The reference's own real-data shapes (SURVEY.md §8c "[probe] config-1 goldens"; /root/reference/test_data/perf_tests.sh:150-191):

  * the 250 bp Illumina read against the 1 Mb chromosome, HW -l  =>  108, (350889,351126) (350889,351127);
  * every file of test_data/E_coli_DH1/mason_illumina_reads/{50,100,250,500,10k}bp (HW) and prefixes/* (SHW) against
    Chromosome_2890043_3890042_0.fasta, each directory as ONE shared-target batch, tasks distance / locations / path;
  * the seven "Chromosome, NW" pairs (1,000,000 x ~1,000,000 bases): score, location, md5 of the op bytes and of both
    CIGAR lines -- the band beyond every lane ring (distances 9,927 ... 395,021) and the Hirschberg regime.

tests/golden/realdata/ holds the reference's data files (the 1 Mb ones xz-compressed) and expected.json, made by the
compiled reference (oracle/gen_realdata_golden.py).  CPU part: the C99 restatement (and oracle/_ref where it travelled)
reproduces the fixtures.  GPU part: the engine, through the C ABI."""
import hashlib
import json
import lzma
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "tests", "golden", "realdata")
with open(os.path.join(REAL, "expected.json")) as _f:
    EXP = json.load(_f)
CHROM = "Chromosome_2890043_3890042_0.fasta.xz"
FIELDS = ("editDistance", "numLocations", "alphabetLength", "endLocations", "startLocations", "alignmentLength")


def read_fasta(path):
    """first record of a FASTA file (plain or .xz) as bytes (the CLI's reader: apps/aligner/aligner.cpp:290-328)."""
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


_cache = {}


def chromosome():
    if "t" not in _cache:
        _cache["t"] = read_fasta(os.path.join(REAL, "chromosome", CHROM))
    return _cache["t"]


def md5(b):
    return hashlib.md5(b).hexdigest()


def check_result(got, want, cigar, name):
    """every field of EdlibAlignResult against the fixture; op bytes through their md5 and both CIGAR lines."""
    for f in FIELDS:
        assert got[f] == want[f], "%s: %s: got %r want %r" % (name, f, got[f], want[f])
    if "ops_md5" in want:
        assert got["alignment"] is not None, name
        assert md5(got["alignment"]) == want["ops_md5"], name + ": op bytes"
        ext, std = cigar(got["alignment"], True), cigar(got["alignment"], False)
        assert md5((ext + "\n").encode()) == want["cigar_ext_md5"] and len(ext) == want["cigar_ext_len"], name + ": CIG_EXT"
        assert md5((std + "\n").encode()) == want["cigar_std_md5"] and len(std) == want["cigar_std_len"], name + ": CIG_STD"
        if "cigar_ext" in want:
            assert ext == want["cigar_ext"], name
    else:
        assert got["alignment"] is None, name


def test_survey_golden_is_in_the_fixture():
    b = [b for b in EXP["mason"] if b["dir"].endswith("/250bp")][0]
    c = [c for c in b["cases"] if c["file"] == "e_coli_DH1_illumina_1x250.fasta"][0]["locations"]
    assert c["editDistance"] == 108
    assert list(zip(c["startLocations"], c["endLocations"])) == [(350889, 351126), (350889, 351127)]
    assert len(chromosome()) == 1000000
    assert [c["percent"] for c in EXP["chromosome"]] == [99, 97, 94, 90, 80, 70, 60]
    assert [c["editDistance"] for c in EXP["chromosome"]] == [9927, 31467, 62190, 99451, 201673, 306456, 395021]


def _batches(short_only):
    out = []
    for key in ("mason", "prefixes"):
        for b in EXP[key]:
            if short_only and b["dir"].endswith("10kbp"):
                continue
            out.append(b)
    return out


@pytest.mark.parametrize("batch", _batches(True), ids=lambda b: b["dir"])
def test_oracle_reproduces_read_fixtures(oracle, ref, batch):
    """the restatement (and the compiled reference where it is) on the short-read directories: all three tasks."""
    t = chromosome()
    for c in batch["cases"]:
        q = read_fasta(os.path.join(REAL, batch["dir"], c["file"]))
        assert len(q) == c["qlen"]
        for impl in (oracle, ref):
            if impl is None:
                continue
            for task in ("distance", "locations", "path"):
                got = impl.align(q, t, batch["mode"], task, -1)
                check_result(got, c[task], lambda ops, e: impl.cigar(ops, 1 if e else 0), "%s/%s %s" % (batch["dir"], c["file"], task))


def test_oracle_reproduces_chromosome_99(oracle):
    c = EXP["chromosome"][0]
    q = read_fasta(os.path.join(REAL, "chromosome", c["query"]))
    got = oracle.align(q, chromosome(), "NW", "distance", -1)
    assert (got["editDistance"], got["endLocations"], len(q)) == (9927, [999999], c["qlen"])


def _myers_query(c):
    if "file" in c:
        return read_fasta(os.path.join(REAL, "mason_illumina_reads", "10kbp", c["file"]))
    return read_fasta(os.path.join(REAL, "chromosome", c["slice_of"]))[c["from"]:c["to"]]


def _myers_id(c):
    return "%s-k%d" % (c.get("file", c.get("slice_of", "")).split(".")[0] + ("" if "file" in c else "-slice"), c["k"])


MYERS = EXP.get("myers", []) + EXP.get("myers_related", [])


def test_fixed_k_fixture_covers_the_reference_block():
    """perf_tests.sh:195-219: HW on the 10 kbp reads with k = 100 (one file), 1000 (four files), 10000 (all seven)"""
    got = {}
    for c in EXP["myers"]:
        got.setdefault(c["k"], set()).add(c["file"])
    assert len(got[100]) == 1 and len(got[1000]) == 4 and len(got[10000]) == 7
    assert all(c["distance"]["editDistance"] == -1 for c in EXP["myers"] if c["k"] in (100, 1000))
    assert sorted(c["distance"]["editDistance"] for c in EXP["myers_related"] if c["k"] == 1000) == [108, 305, 657, 967]


@pytest.mark.parametrize("case", [c for c in MYERS if c["k"] <= 1000], ids=_myers_id)
def test_oracle_reproduces_fixed_k_fixtures(oracle, case):
    """(the thresholds the restatement answers in about a second each; the rest on the GPU side)"""
    q = _myers_query(case)
    got = oracle.align(q, chromosome(), "HW", "locations", case["k"])
    check_result(got, case["locations"], None, _myers_id(case))


# ------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("case", MYERS, ids=_myers_id)
def test_gpu_fixed_k_single_calls(engine, case):
    """the reference's "Myers" block (test_data/perf_tests.sh:195-219): a fixed k on 10 kbp HW queries, one edlibAlign() each"""
    q = _myers_query(case)
    for task in ("distance", "locations"):
        got = engine.align_raw(q, chromosome(), "HW", task, case["k"])
        assert got["status"] == 0
        check_result(got, case[task], engine.cigar_from_alignment, "%s %s" % (_myers_id(case), task))


@pytest.mark.gpu
@pytest.mark.parametrize("k", sorted(set(c["k"] for c in MYERS)))
def test_gpu_fixed_k_as_shared_target_batches(engine, k):
    """every query that has a fixture at this k, as ONE shared-target batch (piece filter / verification with the caller's k)"""
    cases = [c for c in MYERS if c["k"] == k]
    got = engine.align_batch([_myers_query(c) for c in cases], chromosome(), mode="HW", task="locations", k=k, raw=True)
    for g, c in zip(got, cases):
        assert g["status"] == 0
        check_result(g, c["locations"], engine.cigar_from_alignment, _myers_id(c))


@pytest.mark.gpu
@pytest.mark.parametrize("batch", _batches(False), ids=lambda b: b["dir"])
@pytest.mark.parametrize("task", ["distance", "locations", "path"])
def test_gpu_read_directories_as_shared_target_batches(engine, batch, task):
    t = chromosome()
    qs = [read_fasta(os.path.join(REAL, batch["dir"], c["file"])) for c in batch["cases"]]
    got = engine.align_batch(qs, t, mode=batch["mode"], task=task, raw=True)
    for g, c in zip(got, batch["cases"]):
        assert g["status"] == 0
        check_result(g, c[task], engine.cigar_from_alignment, "%s/%s %s" % (batch["dir"], c["file"], task))


@pytest.mark.gpu
def test_gpu_survey_golden_single_call(engine):
    """the §8(c) golden through plain edlibAlign()."""
    q = read_fasta(os.path.join(REAL, "mason_illumina_reads", "250bp", "e_coli_DH1_illumina_1x250.fasta"))
    got = engine.align_raw(q, chromosome(), "HW", "locations", -1)
    assert got["editDistance"] == 108
    assert list(zip(got["startLocations"], got["endLocations"])) == [(350889, 351126), (350889, 351127)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", EXP["chromosome"], ids=lambda c: "p%d" % c["percent"])
def test_gpu_chromosome_pairs_distance(engine, case):
    q = read_fasta(os.path.join(REAL, "chromosome", case["query"]))
    got = engine.align_raw(q, chromosome(), "NW", "distance", -1)
    assert got["status"] == 0
    assert (got["editDistance"], got["endLocations"], got["numLocations"], got["alphabetLength"]) == \
        (case["editDistance"], case["endLocations"], 1, case["alphabetLength"])
    # a fixed k at / just under the distance (edlib.cpp:197-217: one pass, -1 above k)
    if case["percent"] in (99, 90):
        d = case["editDistance"]
        assert engine.align_raw(q, chromosome(), "NW", "distance", d)["editDistance"] == d
        assert engine.align_raw(q, chromosome(), "NW", "distance", d - 1)["editDistance"] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in EXP["chromosome"] if c["percent"] in (99, 97, 90, 60)], ids=lambda c: "p%d" % c["percent"])
def test_gpu_chromosome_pairs_path(engine, case):
    q = read_fasta(os.path.join(REAL, "chromosome", case["query"]))
    got = engine.align_raw(q, chromosome(), "NW", "path", -1)
    assert got["status"] == 0
    check_result(got, case, engine.cigar_from_alignment, "chromosome p%d" % case["percent"])


@pytest.mark.gpu
def test_gpu_chromosome_pairs_as_one_batch(engine):
    """all seven pairs in ONE pair batch (distance): the wide band next to itself at seven different widths."""
    qs = [read_fasta(os.path.join(REAL, "chromosome", c["query"])) for c in EXP["chromosome"]]
    got = engine.align_pairs(qs, [chromosome()] * len(qs), mode="NW", task="distance", raw=True)
    assert [g["editDistance"] for g in got] == [c["editDistance"] for c in EXP["chromosome"]]
    assert all(g["endLocations"] == [999999] for g in got)


@pytest.mark.gpu
@pytest.mark.parametrize("exe_name", ["build/edlib-aligner-batch", "oracle/_ref/aligner_amd"])
def test_gpu_cli_on_a_chromosome_pair(exe_name, tmp_path):
    """`edlib-aligner -m NW -l` and `-p -f CIG_EXT` on the 99 % Chromosome pair (the command line of
    test_data/perf_tests.sh:180-191), through the batch CLI and through the reference's unmodified CLI linked to this library"""
    import re
    import subprocess
    exe = os.path.join(ROOT, exe_name)
    if not os.path.exists(exe):
        pytest.skip("%s was not built" % exe_name)
    case = EXP["chromosome"][0]
    paths = []
    for name in (case["query"], case["target"]):
        p = str(tmp_path / name[:-3])
        with open(p, "wb") as f:
            f.write(b">" + name.encode() + b"\n")
            seq = read_fasta(os.path.join(REAL, "chromosome", name))
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + b"\n")
        paths.append(p)
    out = subprocess.run([exe, "-m", "NW", "-l"] + paths, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    m = re.search(r"^#0: (-?\d+)\s+(\d+)\s+\[ \((\d+), (\d+)\) \]", out.stdout, re.M)
    assert m and (int(m.group(1)), int(m.group(3)), int(m.group(4))) == (case["editDistance"], 0, 999999), out.stdout[-500:]
    out = subprocess.run([exe, "-m", "NW", "-p", "-f", "CIG_EXT"] + paths, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    m = re.search(r"^Cigar:\n(.*)$", out.stdout, re.M)
    assert m and len(m.group(1)) == case["cigar_ext_len"]
    assert md5((m.group(1) + "\n").encode()) == case["cigar_ext_md5"]
