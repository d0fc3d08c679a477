"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, has the
reference's struct layout, and its host-only entry points behave (no GPU needed).
Without a GPU the compute entry points must FAIL LOUDLY, never fall back."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("edlib.h", "edlib_amd.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"EDLIB_API\s+[^;(]*?\b(edlib\w+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_reference_api():
    names = declared_symbols()
    for n in ("edlibAlign", "edlibNewAlignConfig", "edlibDefaultAlignConfig",
              "edlibFreeAlignResult", "edlibAlignmentToCigar"):      # reference edlib.h:146-271
        assert n in names


def test_library_exports_every_declared_symbol():
    import edlib_amd
    out = subprocess.run(["nm", "-D", "--defined-only", edlib_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    missing = [n for n in declared_symbols() if n not in exported]
    assert not missing, missing
    L = edlib_amd.lib()
    for n in declared_symbols():
        assert hasattr(L, n)


def test_library_exports_only_declared_symbols():
    """The reference's shared library exports its five functions and nothing else; this one the
    declarations of include/*.h and nothing else -- no kernel handles, no C++ symbols (edlib_amd/csrc/exports.map)."""
    import edlib_amd
    out = subprocess.run(["nm", "-D", "--defined-only", edlib_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert defined == declared_symbols(), sorted(set(defined) ^ set(declared_symbols()))
    kinds = set(l.split()[-2] for l in out.splitlines() if l.strip())
    assert kinds == {"T"}, kinds


def test_struct_layout_matches_reference():
    import edlib_amd
    assert C.sizeof(edlib_amd.AlignConfig) == 32        # SURVEY.md §8b, x86-64 SysV
    assert C.sizeof(edlib_amd.AlignResult) == 48
    assert edlib_amd.AlignResult.endLocations.offset == 8
    assert edlib_amd.AlignResult.alignment.offset == 32
    assert edlib_amd.AlignResult.alphabetLength.offset == 44
    assert edlib_amd.EDLIB_MODE == {"NW": 0, "SHW": 1, "HW": 2}


def test_config_helpers():
    import edlib_amd
    L = edlib_amd.lib()
    d = L.edlibDefaultAlignConfig()                     # reference edlib.cpp:1477-1479
    assert (d.k, d.mode, d.task, d.additionalEqualitiesLength) == (-1, 0, 0, 0)
    assert not d.additionalEqualities
    c = L.edlibNewAlignConfig(7, 2, 1, None, 0)
    assert (c.k, c.mode, c.task) == (7, 2, 1)


def test_cigar_host_entry_point():
    import edlib_amd
    ops = bytes([0, 0, 1, 1, 1, 2, 1, 1, 3, 0, 0])      # runTests.cpp:506-533
    assert edlib_amd.cigar_from_alignment(ops, True) == "2=3I1D2I1X2="
    assert edlib_amd.cigar_from_alignment(ops, False) == "2M3I1D2I3M"
    assert edlib_amd.cigar_from_alignment(b"", True) == ""
    assert edlib_amd.cigar_from_alignment(bytes([0, 9]), True) is None
    L = edlib_amd.lib()
    assert L.edlibAlignmentToCigar(ops, len(ops), 5) is None        # bad format -> NULL


def test_no_silent_cpu_fallback():
    import edlib_amd
    if edlib_amd.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the -m gpu tests")
    r = edlib_amd.align_raw(b"ACGT", b"ACGA", "NW", "distance", -1)
    assert r["status"] == 1 and r["editDistance"] == -1 and r["endLocations"] is None
    with pytest.raises(Exception):
        edlib_amd.align("ACGT", "ACGA")
    with pytest.raises(RuntimeError):
        edlib_amd.SharedBatch([b"ACGT"], b"ACGTACGT")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under edlib_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "edlib_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower().replace("oracle-verified", ""), os.path.join(dirpath, f)


def test_negative_lengths_are_errors_everywhere():
    """edlibAlign() answers EDLIB_STATUS_ERROR for a negative length; the one-shot batch entry points must not turn
    it into an empty sequence (decided before any device work, so this runs without a GPU)."""
    import edlib_amd
    L = edlib_amd.lib()
    cfg = L.edlibDefaultAlignConfig()
    r = L.edlibAlign(b"ACGT", -1, b"ACGT", 4, cfg)
    assert r.status == 1 and r.editDistance == -1 and not r.endLocations
    qs = (C.c_char_p * 2)(b"ACGT", b"AC")
    res = (edlib_amd.AlignResult * 2)()
    assert L.edlibAlignBatchSharedTarget(qs, (C.c_int * 2)(4, -2), 2, b"ACGTT", 5, cfg, res) == 1
    assert "negative" in edlib_amd.last_error() and res[0].status == 1 and res[1].status == 1
    ts = (C.c_char_p * 2)(b"ACGT", b"AC")
    assert L.edlibAlignBatchPairs(qs, (C.c_int * 2)(4, 2), ts, (C.c_int * 2)(-4, 2), 2, cfg, res) == 1
    assert L.edlibAlignBatchSharedTarget(qs, (C.c_int * 2)(4, 2), 2, b"ACGTT", -5, cfg, res) == 1
    L.edlibAmdFreeResults(res, 2)               # nothing to free, must not crash
    L.edlibAmdTrim()                            # empty cache, must not crash


def test_every_environment_knob_is_documented():
    """every EDLIB_AMD_* variable the library reads (getenv in edlib_amd/csrc, os.environ in the Python front) has its row in
    INTEGRATION.md: a knob that steers routing is part of the boundary a maintainer sees"""
    import glob
    import re
    names = set()
    for path in glob.glob(os.path.join(ROOT, "edlib_amd", "csrc", "*")):
        with open(path, errors="replace") as f:
            names.update(re.findall(r'getenv\("(EDLIB_AMD_[A-Z0-9_]+)"\)', f.read()))
    with open(os.path.join(ROOT, "edlib_amd", "__init__.py")) as f:
        names.update(re.findall(r'environ(?:\.get)?\(?\[?"(EDLIB_AMD_[A-Z0-9_]+)"', f.read()))
    assert len(names) >= 10
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_every_routing_knob_has_a_test():
    """A switch that selects a code path is either exercised by a test (GPU tests set it, a fresh interpreter where it is
    read at load time) or it does not exist: round 4 shipped 16 switches that selected untested variants.  Exempt: the
    device selection a deployment needs (tested in test_gpu_robustness.py anyway), the thread count, and EDLIB_AMD_DEBUG /
    EDLIB_AMD_LIB, which select no result-producing path."""
    import glob
    import re
    names = set()
    for path in glob.glob(os.path.join(ROOT, "edlib_amd", "csrc", "*")):
        with open(path, errors="replace") as f:
            names.update(re.findall(r'getenv\("(EDLIB_AMD_[A-Z0-9_]+)"\)', f.read()))
    exempt = {"EDLIB_AMD_DEBUG", "EDLIB_AMD_HOST_THREADS"}
    tests = ""
    for path in glob.glob(os.path.join(ROOT, "tests", "test_gpu_*.py")):
        with open(path) as f:
            tests += f.read()
    untested = sorted(n for n in names - exempt if not re.search(n + r"\b", tests))
    assert not untested, untested
    assert len(names) <= 17, sorted(names)        # (round 6: + EDLIB_AMD_LANEPAIR)
