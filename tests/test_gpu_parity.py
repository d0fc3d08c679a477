"""-m gpu: the HIP engine, through the C ABI, against the reference's answers.

Every field of EdlibAlignResult must be bit-identical (integer work: no tolerance)."""
import numpy as np
import pytest

import golden_cases as gc
from conftest import load_golden

pytestmark = pytest.mark.gpu

FIELDS = ("status", "editDistance", "endLocations", "startLocations", "numLocations",
          "alignment", "alignmentLength", "alphabetLength")


def check(engine_result, want, name):
    for f in FIELDS:
        assert engine_result[f] == want[f], "%s: field %s: got %r want %r" % (name, f, engine_result[f], want[f])


def run_case(engine, case):
    q, t = gc.materialise(case)
    return engine.align_raw(q, t, case["mode"], case["task"], case["k"], gc.eq_pairs(case))


@pytest.mark.parametrize("fname", ["kat.json", "fuzz_ref.json"])
def test_golden_single_calls(engine, fname):
    """edlibAlign() one call at a time (the drop-in entry point)."""
    bad = []
    for case in load_golden(fname):
        want = gc.expected(case)
        got = run_case(engine, case)
        for f in FIELDS:
            if got[f] != want[f]:
                bad.append((case["name"], f, got[f], want[f]))
                break
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:5])


def test_reference_asserts(engine):
    """What the reference's own suites assert (runTests.cpp:269-587, bindings/python/test.py)."""
    for case in load_golden("kat.json"):
        a = case.get("asserts") or {}
        q, t = gc.materialise(case)
        got = engine.align_raw(q, t, case["mode"], case["task"], case["k"], gc.eq_pairs(case))
        if "editDistance" in a:
            assert got["editDistance"] == a["editDistance"], case["name"]
        if "alphabetLength" in a:
            assert got["alphabetLength"] == a["alphabetLength"], case["name"]
        if "query_aligned" in a:
            nice = engine.getNiceAlignment(
                engine.align(q.decode(), t.decode(), mode=case["mode"], task="path"), q.decode(), t.decode())
            for k in ("query_aligned", "matched_aligned", "target_aligned"):
                assert nice[k] == a[k], (case["name"], k)


def test_cigar_kat(engine):
    ops = bytes([0, 0, 1, 1, 1, 2, 1, 1, 3, 0, 0])       # runTests.cpp:506-533
    assert engine.cigar_from_alignment(ops, True) == "2=3I1D2I1X2="
    assert engine.cigar_from_alignment(ops, False) == "2M3I1D2I3M"


def test_golden_synth_single_calls(engine):
    """BASELINE-shaped cases (150 bp HW reads vs 20 kb..5 Mb, 10 kb NW, 1 kb NW PATH, 94 kb NW)."""
    bad = []
    for case in load_golden("synth_ref.json"):
        want = gc.expected(case)
        got = run_case(engine, case)
        for f in FIELDS:
            if got[f] != want[f]:
                bad.append((case["name"], f, got[f], want[f]))
                break
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:5])


def _group(cases, key):
    groups = {}
    for c in cases:
        groups.setdefault(key(c), []).append(c)
    return groups


def test_golden_batches_shared_target(engine):
    """The c2.* fixtures again, but as ONE batch per target through the reads-per-lane kernel."""
    cases = [c for c in load_golden("synth_ref.json") if c["input"]["kind"] == "read"]
    groups = _group(cases, lambda c: (c["input"]["tseed"], c["input"]["tn"], c["input"]["m"], c["input"]["seed"],
                                      c["task"], c["mode"]))
    for key, cs in groups.items():
        qs = [gc.materialise(c)[0] for c in cs]
        t = gc.materialise(cs[0])[1]
        got = engine.align_batch(qs, t, mode=cs[0]["mode"], task=cs[0]["task"], k=-1, raw=True)
        for c, g in zip(cs, got):
            check(g, gc.expected(c), c["name"] + " (batched)")


def test_golden_batches_pairs(engine):
    """fuzz fixtures grouped by (mode, task, k, eq) and run as pair batches."""
    cases = load_golden("fuzz_ref.json")
    groups = _group(cases, lambda c: (c["mode"], c["task"], c["k"], str(c["eq"])))
    for key, cs in groups.items():
        qs, ts = zip(*[gc.materialise(c) for c in cs])
        got = engine.align_pairs(list(qs), list(ts), mode=cs[0]["mode"], task=cs[0]["task"], k=cs[0]["k"],
                                 additionalEqualities=gc.eq_pairs(cs[0]), raw=True)
        for c, g in zip(cs, got):
            check(g, gc.expected(c), c["name"] + " (pair batch)")
