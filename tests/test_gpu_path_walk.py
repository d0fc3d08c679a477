"""The traceback walk over the two-plane column store (pair_kernels.hpp StoreEntry): NW paths of 300 random pairs of 1..270
bases against the reference, with the place where a walk leaves the reference's path in the failure message
(obtainAlignmentTraceback, edlib.cpp:942-1141: up > left > diagonal)."""
import random

import pytest

pytestmark = pytest.mark.gpu


def _pairs(seed, n, alphabet=b"ACG"):
    rng = random.Random(seed)
    qs, ts = [], []
    for _ in range(n):
        m = rng.randrange(1, 260)
        t = bytes(rng.choice(alphabet) for _ in range(m))
        q = bytearray()
        for ch in t:
            r = rng.random()
            if r < 0.04:
                continue
            if r < 0.08:
                q.append(rng.choice(alphabet))
            if r < 0.12:
                q.append(rng.choice(alphabet))
                continue
            q.append(ch)
        qs.append(bytes(q) or b"A")
        ts.append(t)
    return qs, ts


@pytest.mark.parametrize("seed,band", [(5, "1"), (6, "1"), (7, "0")])
def test_walk_matches_reference(engine, checker, monkeypatch, seed, band):
    monkeypatch.setenv("EDLIB_AMD_NWBAND", band)          # "0": the strips' layout instead of the rings'
    qs, ts = _pairs(seed, 300)
    got = engine.align_pairs(qs, ts, mode="NW", task="path", raw=True)
    bad = []
    for i, (q, t, g) in enumerate(zip(qs, ts, got)):
        w = checker.align(q, t, "NW", "path", -1)
        if g["editDistance"] == w["editDistance"] and g["alignment"] == w["alignment"]:
            continue
        ga, wa = g["alignment"] or b"", w["alignment"] or b""
        k = 0
        while k < min(len(ga), len(wa)) and ga[len(ga) - 1 - k] == wa[len(wa) - 1 - k]:
            k += 1
        r, c = len(q) - 1, len(t) - 1
        for op in wa[::-1][:k]:
            r, c = r - (op != 2), c - (op != 1)
        bad.append("pair %d m=%d T=%d: leaves the path after %d ops at r=%d c=%d" % (i, len(q), len(t), k, r, c))
    assert not bad, "\n".join(bad[:10])
