"""-m gpu: seeded differential fuzz of the batch entry points against the compiled reference (`checker`: oracle/_ref when it
travelled, else the C99 restatement), aimed at the band
logic of the reads kernel (word-count groups, thresholds around the first k of the doubling, low
complexity targets where the band stays tall, fixed k) and at the pair kernels (strips, banded NW)."""
import os
import random

import numpy as np
import pytest

from edlib_amd import synth

pytestmark = pytest.mark.gpu
FIELDS = ("status", "editDistance", "endLocations", "startLocations", "numLocations",
          "alignment", "alignmentLength", "alphabetLength")


def _mut(rng, s, rate, sigma_bytes):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(sigma_bytes)); out.append(ch); continue
        if x < rate:
            out.append(rng.choice(sigma_bytes)); continue
        out.append(ch)
    return bytes(out) or bytes([sigma_bytes[0]])


def test_fuzz_shared_target_batches(engine, checker):
    oracle = checker
    rng = random.Random(int(os.environ.get("EDLIB_FUZZ_SEED", "2024")))
    nbad = 0
    for it in range(int(os.environ.get("EDLIB_FUZZ_ITERS", "70"))):   # soak runs: EDLIB_FUZZ_ITERS=500 EDLIB_FUZZ_SEED=n
        sigma = rng.choice([1, 2, 2, 3, 4, 4, 4])
        alpha = b"ACGT"[:sigma]
        tn = rng.choice([40, 300, 1000, 5000, 20000, 70000])
        kind = rng.random()
        if kind < 0.25:                                   # low complexity / periodic target
            unit = bytes(rng.choice(alpha) for _ in range(rng.choice([1, 2, 3, 7, 31])))
            target = (unit * (tn // len(unit) + 1))[:tn]
        else:
            target = bytes(rng.choice(alpha) for _ in range(tn))
        nq = rng.choice([1, 3, 40, 70])
        lens = [rng.choice([1, 5, 31, 32, 33, 63, 64, 65, 96, 100, 128, 150, 159, 160, 161, 200, 255, 256])
                for _ in range(nq)]
        if rng.random() < 0.5:
            lens = [lens[0]] * nq                          # uniform length batch
        qs = []
        for m in lens:
            if rng.random() < 0.8 and tn > m:
                a = rng.randrange(0, tn - m + 1)
                qs.append(_mut(rng, target[a:a + m], rng.choice([0.0, 0.02, 0.06, 0.12, 0.3]), alpha))
            else:
                qs.append(bytes(rng.choice(b"ACGTN") for _ in range(m)))
        mode = rng.choice(["HW", "HW", "HW", "SHW", "NW"])
        task = rng.choice(["distance", "distance", "locations", "path"])
        k = rng.choice([-1, -1, -1, 0, 3, 7, 8, 9, 15, 16, 17, 40, 300])
        got = engine.align_batch(qs, target, mode=mode, task=task, k=k, raw=True)
        for q, g in zip(qs, got):
            want = oracle.align(q, target, mode, task, k)
            if want["status"] == 2:
                continue
            if any(g[f] != want[f] for f in FIELDS):
                nbad += 1
                if nbad <= 3:
                    print("MISMATCH it=%d mode=%s task=%s k=%d m=%d tn=%d sigma=%d\n got=%r\nwant=%r"
                          % (it, mode, task, k, len(q), tn, sigma, g, want))
    assert nbad == 0


def test_fuzz_pair_batches_long(engine, checker):
    oracle = checker
    """Pairs around the strip (64 blocks = 4096 rows) and band limits of the pair kernels."""
    rng = random.Random(77)
    for it in range(6):
        qs, ts = [], []
        for _ in range(6):
            tn = rng.choice([3000, 4096, 4200, 8300, 12000])
            t = synth.random_dna(rng.randrange(1 << 30), tn)
            rate = rng.choice([0.0, 0.01, 0.05, 0.2, 0.45])
            q, _ = synth.mutate(t, rng.randrange(1 << 30), rate, rate / 2, rate / 2)
            if rng.random() < 0.2:
                q = q[:len(q) // 2]                        # big length difference
            qs.append(q.tobytes() or b"A"); ts.append(t.tobytes())
        mode = rng.choice(["NW", "NW", "SHW", "HW"])
        k = rng.choice([-1, -1, 100, 1000, 5000])
        got = engine.align_pairs(qs, ts, mode=mode, task="distance", k=k, raw=True)
        for q, t, g in zip(qs, ts, got):
            want = oracle.align(q, t, mode, "distance", k)
            assert all(g[f] == want[f] for f in FIELDS), (it, mode, k, len(q), len(t), g, want)
