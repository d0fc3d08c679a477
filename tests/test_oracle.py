"""CPU: pin the oracle (the C99 restatement) before anything trusts it.

(a) the reference's own known answers, (b) answers the COMPILED reference gave on seeded
inputs (tests/golden/*.json, made by oracle/gen_golden.py), (c) an independent O(mT) DP,
(d) when oracle/_ref/libedlib_ref.so is present, a live differential fuzz."""
import random

import pytest

import golden_cases as gc
from conftest import load_golden

FIELDS = ("status", "editDistance", "endLocations", "startLocations", "numLocations",
          "alignment", "alignmentLength", "alphabetLength")


def run(impl, case):
    q, t = gc.materialise(case)
    return impl.align(q, t, case["mode"], case["task"], case["k"], gc.eq_pairs(case))


@pytest.mark.parametrize("fname", ["kat.json", "fuzz_ref.json", "synth_ref.json"])
def test_oracle_matches_reference_fixtures(oracle, fname):
    n = 0
    for case in load_golden(fname):
        want = gc.expected(case)
        got = run(oracle, case)
        if got["status"] == 2:          # Hirschberg regime: not restated
            assert case["task"] == "path"
            continue
        for f in FIELDS:
            assert got[f] == want[f], (case["name"], f, got[f], want[f])
        n += 1
    assert n > 50


def test_reference_suite_asserts(oracle):
    """Values asserted by runTests.cpp:427-553 and bindings/python/test.py:6-80."""
    for case in load_golden("kat.json"):
        a = case.get("asserts") or {}
        got = run(oracle, case)
        for key in ("editDistance", "alphabetLength"):
            if key in a:
                assert got[key] == a[key], (case["name"], key)


def test_oracle_vs_simple_dp(oracle):
    """runTests.cpp:86-213 in miniature: score and all end locations against an O(mT) DP."""
    import ctypes as C
    lib = oracle.lib
    lib.oracle_simple_dp.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_int)]
    rng = random.Random(5)
    for it in range(300):
        sigma = rng.choice([2, 4, 10])
        m, tn = rng.randrange(1, 200), rng.randrange(1, 400)
        q = bytes(rng.randrange(sigma) for _ in range(m))
        t = bytes(rng.randrange(sigma) for _ in range(tn))
        for mode_name, mode in (("NW", 0), ("SHW", 1), ("HW", 2)):
            score = C.c_int(); pos = C.POINTER(C.c_int)(); npos = C.c_int()
            lib.oracle_simple_dp(q, m, t, tn, mode, C.byref(score), C.byref(pos), C.byref(npos))
            got = oracle.align(q, t, mode_name, "distance", -1)
            assert got["editDistance"] == score.value
            ends = [e for e in got["endLocations"] if e >= 0]      # the DP has no position -1
            assert ends == [pos[i] for i in range(npos.value)]
            oracle.libc.free(pos)


def test_cigar(oracle):
    ops = bytes([0, 0, 1, 1, 1, 2, 1, 1, 3, 0, 0])                  # runTests.cpp:506-533
    assert oracle.cigar(ops, 1) == "2=3I1D2I1X2="
    assert oracle.cigar(ops, 0) == "2M3I1D2I3M"
    assert oracle.cigar(b"", 1) == ""
    assert oracle.cigar(bytes([4]), 1) is None and oracle.cigar(ops, 7) is None


def test_live_differential_fuzz(oracle, ref):
    if ref is None:
        pytest.skip("oracle/_ref/libedlib_ref.so not built here")
    rng = random.Random(11)
    for it in range(1500):
        sigma = rng.choice([1, 2, 4, 4, 20])
        m = rng.choice([1, 5, 63, 64, 65, 128, 150, rng.randrange(1, 300)])
        t = bytes(65 + rng.randrange(sigma) for _ in range(rng.randrange(1, 700)))
        if rng.random() < 0.5:
            a = rng.randrange(len(t))
            q = bytearray(t[a:a + m] or b"A")
            for _ in range(rng.randrange(0, 6)):
                q[rng.randrange(len(q))] = 65 + rng.randrange(sigma)
            q = bytes(q)
        else:
            q = bytes(65 + rng.randrange(sigma) for _ in range(m))
        mode = rng.choice(["NW", "SHW", "HW"]); task = rng.choice(["distance", "locations", "path"])
        k = rng.choice([-1, -1, 0, 3, 30, 500])
        a = oracle.align(q, t, mode, task, k); b = ref.align(q, t, mode, task, k)
        if a["status"] == 2:
            continue
        assert a == b, (mode, task, k, q, t)


def test_invalid_mode_values(oracle, ref):
    """a mode outside {NW, SHW, HW} (edlib.cpp:205-225: the distance is computed as NW but the end-location fix-up tests
    mode == NW, so endLocations stays NULL and numLocations 0; :177-179: with an empty input the status is ERROR).
    Pinned against the compiled reference where it is here; the restatement has to say the same."""
    cases = [(b"ACGTACGT", b"ACGTTCGT"), (b"AAAA", b"TTTTTT"), (b"ACGT" * 40, b"ACGA" * 41), (b"", b"ACGT"), (b"ACGT", b""), (b"", b"")]
    for mode in (3, 7, -1, 100):
        for task in ("distance", "locations"):       # (TASK_PATH with such a mode: the reference dereferences its NULL endLocations)
            for k in (-1, 0, 3):
                for q, t in cases:
                    got = oracle.align(q, t, mode, task, k)
                    if q and t:
                        assert got["status"] == 0
                        nw = oracle.align(q, t, "NW", "distance", k)
                        assert got["editDistance"] == nw["editDistance"], (mode, task, k, q, t)
                    if ref is not None:
                        want = ref.align(q, t, mode, task, k)
                        for f in ("status", "editDistance", "endLocations", "startLocations", "numLocations", "alignment", "alphabetLength"):
                            assert got[f] == want[f], (mode, task, k, q, t, f, got[f], want[f])
