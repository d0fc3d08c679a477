"""-m gpu: edlibAlign() on small pairs runs as ONE kernel launch (edlib_amd/csrc/one_pair.hip: queries up to 1024 rows,
targets up to 4096 columns, identity equality); everything else, and what that kernel declines (more than 64 end locations,
a column store beyond its LDS budget), takes the general batch-of-one path.  Both must give the reference's answer in every
field (edlib.cpp:146-301); the known answers of the reference's own tests are part of tests/test_gpu_parity.py."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
FIELDS = ("status", "editDistance", "endLocations", "startLocations", "numLocations", "alignment", "alignmentLength",
          "alphabetLength")


def _mut(rng, s, rate, alpha):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(alpha)); out.append(ch); continue
        out.append(rng.choice(alpha) if x < rate else ch)
    return bytes(out) or bytes([alpha[0]])


def _cases(seed, n):
    rng = random.Random(seed)
    for _ in range(n):
        alpha = rng.choice([b"ACGT", b"ACGT", b"AB", b"A", bytes(range(33, 97)), bytes(range(256))])
        T = rng.choice([1, 2, 5, 63, 64, 65, 100, 128, 300, 1000, 2500, 4096])
        m = rng.choice([1, 2, 7, 63, 64, 65, 100, 127, 128, 129, 200, 500, 1000, 1024])
        t = bytes(rng.choice(alpha) for _ in range(T))
        kind = rng.random()
        if kind < 0.6 and T >= 2:
            a = rng.randrange(0, max(1, T - min(m, T) + 1))
            q = _mut(rng, t[a:a + m], rng.choice([0.0, 0.03, 0.1, 0.3]), alpha)[:1024]
        elif kind < 0.8:
            q = bytes(rng.choice(alpha) for _ in range(m))
        else:                                                            # low complexity: many end locations
            u = bytes(rng.choice(alpha) for _ in range(rng.choice([1, 2, 3])))
            t = (u * (T // len(u) + 1))[:T]
            q = (u * (m // len(u) + 1))[:m]
        yield q, t, rng.choice(["NW", "SHW", "HW", "HW"]), rng.choice(["distance", "locations", "path"]), \
            rng.choice([-1, -1, -1, 0, 1, 5, 30, 200, 5000])


def test_single_calls_against_the_reference(engine, checker):
    bad = 0
    for q, t, mode, task, k in _cases(4242, 700):
        got = engine.align_raw(q, t, mode, task, k)
        want = checker.align(q, t, mode, task, k)
        if want["status"] == 2:
            continue
        if any(got[f] != want[f] for f in FIELDS):
            bad += 1
            if bad <= 3:
                print("MISMATCH mode=%s task=%s k=%d m=%d T=%d\n got=%r\nwant=%r" % (mode, task, k, len(q), len(t), got, want))
    assert bad == 0


def test_limits_of_the_fused_kernel(engine, checker):
    """around 1024 x 4096, 64 end locations, and paths whose store does not fit"""
    rng = random.Random(7)
    t = bytes(rng.choice(b"ACGT") for _ in range(4200))
    for m, T in ((1024, 4096), (1025, 4096), (1024, 4097), (1000, 1000), (1023, 3000), (200, 4096), (64, 64), (65, 65)):
        q = _mut(rng, t[5:5 + m], 0.05, b"ACGT")[:m]
        for mode in ("NW", "HW", "SHW"):
            for task in ("distance", "path"):
                got = engine.align_raw(q, t[:T], mode, task, -1)
                want = checker.align(q, t[:T], mode, task, -1)
                assert all(got[f] == want[f] for f in FIELDS), (m, T, mode, task)
    for reps in (10, 63, 64, 65, 66, 300):                               # exactly `reps` end locations in HW mode
        q = b"ACGTTGCA"
        tt = (q + b"TTTT") * reps
        for task in ("distance", "locations"):
            got = engine.align_raw(q, tt[:4096], "HW", task, -1)
            want = checker.align(q, tt[:4096], "HW", task, -1)
            assert all(got[f] == want[f] for f in FIELDS), (reps, task)


def test_general_path_gives_the_same_answers():
    """EDLIB_AMD_ONEPAIR=0 (read when the library loads: a fresh interpreter) sends every call through the batch-of-one path"""
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import edlib_amd\nfrom oracle.oracle import load_ref, load_oracle\nfrom test_gpu_one_pair import _cases, FIELDS\n"
            "chk = load_ref() or load_oracle()\nn = 0\n"
            "for q, t, mode, task, k in _cases(99, 150):\n"
            "    g = edlib_amd.align_raw(q, t, mode, task, k); w = chk.align(q, t, mode, task, k)\n"
            "    assert w['status'] == 2 or all(g[f] == w[f] for f in FIELDS), (mode, task, k, len(q), len(t)); n += 1\n"
            "print('ok', n)\n" % (os.path.dirname(here), here))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, EDLIB_AMD_ONEPAIR="0"))
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-800:] + p.stderr[-2000:]


def test_threads_share_nothing(engine, checker):
    """concurrent callers: the mailbox and stream of the fused path are per host thread"""
    import threading
    cases = list(_cases(5, 60))
    want = [checker.align(q, t, mode, task, k) for q, t, mode, task, k in cases]
    errs = []

    def work(off):
        for i in range(off, len(cases), 4):
            q, t, mode, task, k = cases[i]
            g = engine.align_raw(q, t, mode, task, k)
            if want[i]["status"] != 2 and any(g[f] != want[i][f] for f in FIELDS):
                errs.append(i)
    th = [threading.Thread(target=work, args=(o,)) for o in range(4)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs, errs
