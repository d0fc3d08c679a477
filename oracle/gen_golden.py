#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the COMPILED REFERENCE (TEST INFRASTRUCTURE).

Run in the build container, where /root/reference exists:

    make -C oracle            # builds oracle/_ref/libedlib_ref.so from the reference sources
    python oracle/gen_golden.py

Outputs (committed):
  tests/golden/kat.json       known-answer tests of the reference's own suites
                              (test/runTests.cpp:269-587, bindings/python/test.py:6-80,
                              README.md:63-71, edlib.h:45-54) with the value each suite
                              asserts ("asserts") plus the reference's full answer ("ref")
  tests/golden/fuzz_ref.json  seeded random cases (all modes x tasks x k x alphabets)
  tests/golden/synth_ref.json BASELINE.json-shaped cases (150 bp HW reads vs long targets,
                              10 kb NW pairs, 1 kb NW PATH pairs, a 94 kb pair)
Inputs of the last two are recipes for tests/golden_cases.py, not bytes.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(ROOT, "oracle")]
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from oracle.oracle import load_ref          # noqa: E402
import golden_cases as gc                   # noqa: E402

REF_ROOT = os.environ.get("EDLIB_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def run(ref, case):
    q, t = gc.materialise(case)
    d = ref.align(q, t, case["mode"], case["task"], case["k"], gc.eq_pairs(case))
    d["alignment_rle"] = gc.rle(d.pop("alignment"))
    if d["alignment_rle"] is not None:
        ops = gc.expected({"ref": d})["alignment"]
        d["cigar_ext"] = ref.cigar(ops, 1)
        d["cigar_std"] = ref.cigar(ops, 0)
    case["ref"] = d
    return case


def hexcase(name, q, t, mode, task="path", k=-1, eq=None, asserts=None):
    return {"name": name, "input": {"kind": "hex", "q": bytes(q).hex(), "t": bytes(t).hex()},
            "mode": mode, "task": task, "k": k, "eq": eq, "asserts": asserts or {}}


def c_array(src, func, name):
    """Pull `char name[N] = {...}` out of function `func` of runTests.cpp, zero-padded to N."""
    body = src[src.index("bool %s()" % func):]
    m = re.search(r"char %s\[(\d+)\]\s*=\s*\{([^}]*)\}" % name, body)
    n = int(m.group(1))
    vals = [int(x) for x in re.findall(r"-?\d+", m.group(2))]
    return bytes((v & 0xFF) for v in vals) + bytes(n - len(vals))


def kat_cases():
    src = open(os.path.join(REF_ROOT, "test", "runTests.cpp")).read()
    cases = []
    # runTests.cpp:269-425 test1..test11: every mode, TASK_PATH, checked against the O(mn) DP
    for i in range(1, 11):
        q = c_array(src, "test%d" % i, "query")
        t = c_array(src, "test%d" % i, "target")
        for mode in ("HW", "NW", "SHW"):
            cases.append(hexcase("runTests.test%d.%s" % (i, mode), q, t, mode,
                                 asserts={"matches_simple_dp": True}))
    # test11 (:413-425): CHAR_MIN, mid, CHAR_MAX, zero padded to 8
    q11 = bytes([0x80, 0xFF, 0x7F, 0, 0, 0, 0, 0]); t11 = bytes([0x80, 0x00, 0x7F, 0, 0, 0, 0, 0])
    for mode in ("HW", "NW", "SHW"):
        cases.append(hexcase("runTests.test11.%s" % mode, q11, t11, mode,
                             asserts={"matches_simple_dp": True}))
    # test12 (:427-442) IUPAC equalities, HW LOC, ed 0
    iupac = [list(p) for p in ["RA", "RG", "MA", "MC", "WA", "WT", "SC", "SG", "YC", "YT", "KG", "KT",
                               "VA", "VC", "VG", "HA", "HC", "HT", "DA", "DG", "DT", "BC", "BG", "BT"]]
    m12 = re.search(r'bool test12\(\).*?query = "([A-Z]+)";\s*const char\* target = "([A-Z]+)";', src, re.S)
    cases.append(hexcase("runTests.test12", m12.group(1).encode(), m12.group(2).encode(), "HW", "locations",
                         eq=iupac, asserts={"editDistance": 0}))
    cases.append(hexcase("runTests.test13", b"AA", b"B", "HW", "path", asserts={"editDistance": 2}))
    cases.append(hexcase("runTests.test14", b"AA", b"B", "SHW", "path", asserts={"editDistance": 2}))
    cases.append(hexcase("runTests.test15", b"AAABBB", b"BBBC", "HW", "locations", asserts={"editDistance": 3}))
    cases.append(hexcase("runTests.test16", b"BBBAAA", b"CBBB", "HW", "locations", asserts={"editDistance": 3}))
    # testCustomEqualityRelation (:535-553)
    cases.append(hexcase("runTests.customEquality", b"GTGNRTCARCGAANCTTTN",
                         b"GTGAGTCATCGAATCTTTGAACGCACCTTGCGCTCCTTGGT", "HW", "path",
                         eq=[list(p) for p in ["RA", "RG", "NA", "NC", "NT", "NG"]],
                         asserts={"editDistance": 1}))
    # testEmptySequences (:555-570)
    for mode in ("NW", "SHW", "HW"):
        cases.append(hexcase("runTests.empty.query.%s" % mode, b"", b"ACTG", mode,
                             asserts={"matches_simple_dp": True}))
        cases.append(hexcase("runTests.empty.target.%s" % mode, b"ACTG", b"", mode,
                             asserts={"matches_simple_dp": True}))
    # bindings/python/test.py:6-80
    cases.append(hexcase("py.telephone", b"telephone", b"elephant", "NW", "distance", asserts={"editDistance": 3}))
    cases.append(hexcase("py.equalities", b"ACTG", b"CACTRT", "HW", "path", eq=[["R", "A"], ["R", "G"]],
                         asserts={"editDistance": 0}))
    for mode in ("NW", "HW", "SHW"):
        cases.append(hexcase("py.nice.%s" % mode, b"TAAGGATGGTCCCATTC", b"AAGGGGTCTCATATC", mode, "path",
                             asserts={"query_aligned": "TAAGGATGGTCCCAT-TC",
                                      "matched_aligned": "-||||--||||.|||-||",
                                      "target_aligned": "-AAGG--GGTCTCATATC"}))
    for (q, t, mode, ed) in [(b"", b"elephant", "NW", 8), (b"telephone", b"", "NW", 9),
                             (b"", b"elephant", "HW", 0), (b"telephone", b"", "HW", 9),
                             (b"", b"elephant", "SHW", 0), (b"telephone", b"", "SHW", 9)]:
        cases.append(hexcase("py.empty.%s.%d" % (mode, ed), q, t, mode, "distance", asserts={"editDistance": ed}))
    # unicode: the binding maps the 12 distinct characters to bytes 0..11 (edlib.pyx:22-53)
    a, b = "ты милая", "ты гений"
    alph = {c: i for i, c in enumerate(sorted(set(a) | set(b)))}
    cases.append(hexcase("py.unicode", bytes(alph[c] for c in a), bytes(alph[c] for c in b), "NW", "distance",
                         asserts={"editDistance": 5, "alphabetLength": 12}))
    la = bytes(range(256))
    cases.append(hexcase("py.alphabet256", la * 3, la + la[::-1] + la, "NW", "distance",
                         asserts={"editDistance": 256}))
    # README.md:63-71 / apps/hello-world/helloWorld.c:5-7
    cases.append(hexcase("readme.hello", b"hello", b"world!", "NW", "distance", asserts={"editDistance": 5}))
    # edlib.h:45-46, 53-54
    cases.append(hexcase("header.shw", b"AACT", b"AACTGGC", "SHW", "distance", asserts={"editDistance": 0}))
    cases.append(hexcase("header.hw", b"ACT", b"CGACTGAC", "HW", "distance", asserts={"editDistance": 0}))
    # SURVEY.md §8c extra edge pins and Appendix B pitfalls (answers come from the reference below)
    for mode in ("HW", "SHW", "NW"):
        cases.append(hexcase("edge.ACT.%s" % mode, b"ACT", b"CGACTGAC", mode, "path"))
        cases.append(hexcase("edge.k0.%s" % mode, b"AB", b"CD", mode, "path", k=0))
        cases.append(hexcase("edge.bothEmpty.%s" % mode, b"", b"", mode, "path"))
        cases.append(hexcase("edge.m64.allMismatch.%s" % mode, b"A" * 64, b"B" * 10, mode, "path"))
        cases.append(hexcase("edge.m63.allMismatch.%s" % mode, b"A" * 63, b"B" * 10, mode, "path"))
        cases.append(hexcase("edge.m128.allMismatch.%s" % mode, b"A" * 128, b"B" * 100, mode, "path"))
        cases.append(hexcase("edge.m3.T5.%s" % mode, b"AAA", b"BBBBB", mode, "path"))
    cases.append(hexcase("edge.SHW.BBBAAA", b"BBBAAA", b"CBBB", "SHW", "path"))
    cases.append(hexcase("edge.homopolymer.HW", b"A" * 150, b"A" * 3000, "HW", "locations"))
    cases.append(hexcase("edge.homopolymer.starts", b"AAA", b"A" * 8, "HW", "locations"))
    cases.append(hexcase("edge.manyEnds", b"ACGT" * 10, b"T" * 50, "HW", "distance"))
    for k in (0, 1, 2, 3, 100):
        cases.append(hexcase("edge.fixedK.HW.k%d" % k, b"ACGTACGT", b"TTACGAACGTTT", "HW", "path", k=k))
        cases.append(hexcase("edge.fixedK.NW.k%d" % k, b"ACGTACGT", b"ACGAACGTTT", "NW", "path", k=k))
    return cases


def fuzz_cases():
    cases = []
    seed = 1000
    lens = [1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 150, 160, 161, 192, 200, 256, 257, 300, 400]
    tls = [1, 2, 5, 40, 64, 65, 150, 333, 700, 1500]
    for mode in ("NW", "SHW", "HW"):
        for task in ("distance", "locations", "path"):
            for j in range(56):
                seed += 1
                sigma = [1, 2, 3, 4, 4, 4, 5, 10, 20, 200][j % 10]
                m = lens[(j * 7 + seed) % len(lens)]
                k = [-1, -1, -1, 0, 2, 10, 40, 64, 150, 1000][(j // 3) % 10]
                if j % 2 == 0:
                    inp = {"kind": "rand", "seed": seed, "m": m, "tn": tls[(j * 3 + seed) % len(tls)], "sigma": sigma}
                else:
                    inp = {"kind": "mut", "seed": seed, "tn": m + (j % 5) * (20 if mode != "NW" else 1),
                           "sub": [0.0, 0.02, 0.1, 0.3][j % 4], "ins": [0.0, 0.01, 0.05][j % 3],
                           "del": [0.0, 0.01, 0.05][(j // 2) % 3]}
                c = {"name": "fuzz.%s.%s.%d" % (mode, task, j), "input": inp, "mode": mode, "task": task,
                     "k": k, "eq": None}
                if j % 9 == 0 and inp["kind"] == "rand" and sigma >= 3:
                    c["eq"] = [["A", "B"], ["C", "A"]]
                cases.append(c)
    return cases


def synth_cases():
    cases = []
    # config-2 shape: 150 bp reads, HW, k=-1, growing targets up to the full 5 Mb
    for tn, n_reads in ((20000, 48), (300000, 24), (5000000, 24)):
        for i in range(n_reads):
            cases.append({"name": "c2.T%d.read%d" % (tn, i),
                          "input": {"kind": "read", "seed": 12346, "tseed": 12345, "tn": tn, "m": 150,
                                    "index": i, "n": 64},
                          "mode": "HW", "task": "distance", "k": -1, "eq": None})
    # other read lengths through the same path
    for m in (36, 75, 100, 151, 250):
        for i in range(6):
            cases.append({"name": "c2.m%d.read%d" % (m, i),
                          "input": {"kind": "read", "seed": 777 + m, "tseed": 4242, "tn": 50000, "m": m,
                                    "index": i, "n": 64},
                          "mode": "HW", "task": ["distance", "locations", "path"][i % 3], "k": -1, "eq": None})
    # reads of 257..1024 bases (the 12 / 16 / 24 / 32-word groups of the lane-per-read kernels), and 1100 (pair path)
    for m in (257, 300, 384, 385, 512, 513, 700, 768, 769, 1024, 1100):
        for i in range(3):
            cases.append({"name": "c2.m%d.read%d" % (m, i),
                          "input": {"kind": "read", "seed": 777 + m, "tseed": 4243, "tn": 40000, "m": m,
                                    "index": i, "n": 8},
                          "mode": "HW", "task": ["distance", "locations", "path"][i % 3], "k": -1, "eq": None})
    # config-4 shape: 10 kb ONT-like NW pairs (4/4/4 %) and a lighter 1/1/1 %
    for i in range(6):
        cases.append({"name": "c4.pair%d" % i,
                      "input": {"kind": "mut", "seed": 12349 + i, "tn": 10000,
                                "sub": 0.04 if i < 4 else 0.01, "ins": 0.04 if i < 4 else 0.01,
                                "del": 0.04 if i < 4 else 0.01},
                      "mode": "NW", "task": "distance", "k": -1, "eq": None})
    # config-5 shape: 1 kb NW PATH pairs (3/1/1 %), below the 1 MiB traceback rule
    for i in range(32):
        cases.append({"name": "c5.pair%d" % i,
                      "input": {"kind": "mut", "seed": 12350 + i, "tn": 1000, "sub": 0.03, "ins": 0.01, "del": 0.01},
                      "mode": "NW", "task": "path", "k": -1, "eq": None})
    # PATH in the Hirschberg regime (column store >= 1 MiB, edlib.cpp:1188-1211): 10 kb pairs (config 4
    # with task=path), 3 kb pairs, tie-rich small alphabets, a long read in HW / SHW mode
    for i in range(3):
        cases.append({"name": "hirsch.c4.pair%d" % i,
                      "input": {"kind": "mut", "seed": 22000 + i, "tn": 10000,
                                "sub": [0.04, 0.01, 0.002][i], "ins": [0.04, 0.01, 0.002][i], "del": [0.04, 0.01, 0.002][i]},
                      "mode": "NW", "task": "path", "k": -1, "eq": None})
    for i in range(6):
        cases.append({"name": "hirsch.3kb.pair%d" % i,
                      "input": {"kind": "mut", "seed": 22100 + i, "tn": 3000 + 37 * i, "sub": 0.05, "ins": 0.02, "del": 0.02},
                      "mode": "NW", "task": "path", "k": -1, "eq": None})
    for i, sigma in enumerate((1, 2, 2, 3, 4)):
        cases.append({"name": "hirsch.ties.sigma%d.%d" % (sigma, i),
                      "input": {"kind": "rand", "seed": 22200 + i, "m": 2300 + 100 * i, "tn": 2500 + 64 * i, "sigma": sigma},
                      "mode": ["NW", "NW", "HW", "SHW", "NW"][i], "task": "path", "k": -1, "eq": None})
    for i, mode in enumerate(("HW", "SHW", "HW")):
        cases.append({"name": "hirsch.longread.%s.%d" % (mode, i),
                      "input": {"kind": "read", "seed": 22300 + i, "tseed": 22301, "tn": 60000, "m": 2400, "index": i, "n": 4},
                      "mode": mode, "task": "path", "k": -1, "eq": None})
    cases.append({"name": "hirsch.c1.pair",
                  "input": {"kind": "mut", "seed": 9100, "tn": 94481, "sub": 0.003, "ins": 0.001, "del": 0.001},
                  "mode": "NW", "task": "path", "k": -1, "eq": None})
    # config-1 shape: a 94 kb pair at three divergences (the Phage files cannot travel)
    for i, rate in enumerate((0.003, 0.03, 0.1)):
        cases.append({"name": "c1.pair%d" % i,
                      "input": {"kind": "mut", "seed": 9000 + i, "tn": 94481, "sub": rate, "ins": rate / 3, "del": rate / 3},
                      "mode": "NW", "task": "distance", "k": -1, "eq": None})
    # the divergence of the reference's mutated_60_perc file (its golden there: 39829), far above the
    # band limit of the ring kernel, and a PATH whose Hirschberg levels cannot use the banded scans
    cases.append({"name": "c1.pair3",
                  "input": {"kind": "mut", "seed": 9003, "tn": 94481, "sub": 0.3, "ins": 0.1, "del": 0.1},
                  "mode": "NW", "task": "distance", "k": -1, "eq": None})
    cases.append({"name": "hirsch.c1.div",
                  "input": {"kind": "mut", "seed": 9101, "tn": 94481, "sub": 0.03, "ins": 0.01, "del": 0.01},
                  "mode": "NW", "task": "path", "k": -1, "eq": None})
    for i, (mode, kk) in enumerate((("HW", -1), ("SHW", -1), ("NW", 3000), ("NW", 20000))):
        cases.append({"name": "c1.modes%d" % i,
                      "input": {"kind": "mut", "seed": 9200 + i, "tn": 30000, "sub": 0.1, "ins": 0.03, "del": 0.03},
                      "mode": mode, "task": "locations", "k": kk, "eq": None})
    return cases


def main():
    ref = load_ref()
    if ref is None:
        sys.exit("oracle/_ref/libedlib_ref.so missing: run `make -C oracle` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    for fname, cases in (("kat.json", kat_cases()), ("fuzz_ref.json", fuzz_cases()),
                         ("synth_ref.json", synth_cases())):
        done = [run(ref, c) for c in cases]
        with open(os.path.join(OUT, fname), "w") as f:
            json.dump({"generator": "oracle/gen_golden.py", "reference": "Martinsos/edlib v1.2.6 (compiled, unmodified)",
                       "cases": done}, f, separators=(",", ":"))
            f.write("\n")
        print(fname, len(done), "cases", os.path.getsize(os.path.join(OUT, fname)), "bytes")


if __name__ == "__main__":
    main()
