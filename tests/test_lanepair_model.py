"""The lane-per-pair NW scan (edlib_amd/csrc/lanepair_core.hpp, round 6) without a GPU: the DEVICE code compiled for the host
(tests/lanepair_host.cpp: the same header, one lane at a time) against the oracle -- the window's private row offset and its
virtual rows above the matrix, the slide every 32 columns, the dead tests and trims (reference: the band that follows the
scores, edlib.cpp:799-830), a wave that does not follow the lane's vote, the final decode (edlib.cpp:914-917).
Semantics checked: the value is exact iff it is <= K, and above K otherwise (Ukkonen)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    so = os.path.join(ROOT, "build", "liblanepair_host.so")
    src = os.path.join(ROOT, "tests", "lanepair_host.cpp")
    hdr = os.path.join(ROOT, "edlib_amd", "csrc", "lanepair_core.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.lanepair_host_nw.restype = ctypes.c_int
    lib.lanepair_host_nw.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    return lib


def _mutate(rng, t, sub, ins, dele):
    out = []
    for b in t:
        r = rng.random()
        if r < dele:
            continue
        out.append((b + 1 + rng.integers(3)) % 4 if r < dele + sub else b)
        if rng.random() < ins:
            out.append(rng.integers(4))
    return np.array(out, dtype=np.uint8)


def _text(codes):
    return bytes(b"ACGT"[x] for x in codes)


def _check(lib, checker, q, t, ks, windows, variants):
    ed = checker.align(_text(q), _text(t), "NW", "distance", -1)["editDistance"]
    n = 0
    for K in ks(ed, len(q), len(t)):
        for W in windows:
            for deny, extra_words, extra_blocks in variants:
                ws = ctypes.c_int(0)
                r = lib.lanepair_host_nw(q.tobytes(), len(q), t.tobytes(), len(t), K, W, deny, extra_words, extra_blocks, ctypes.byref(ws))
                if r == -2:
                    continue                                    # the band of this K does not fit W words
                n += 1
                if r == -3:
                    assert abs(len(t) - len(q)) > K and ed > K
                elif ed <= K:
                    assert r == ed, (len(q), len(t), ed, K, W, deny, extra_words, extra_blocks, r)
                else:
                    assert r > K, (len(q), len(t), ed, K, W, deny, extra_words, extra_blocks, r)
    return n


def test_random_pairs_every_threshold_class(host, checker):
    rng = np.random.default_rng(1)
    ks = lambda ed, m, T: sorted({max(0, ed - 1), ed, ed + 1, ed + 17, ed + 64, 2 * ed + 5, max(m, T)})
    # (seed, words beyond the lane's need, blocks beyond its target): the lane alone; trims refused at random; a wave whose
    # maxima exceed the lane's; both
    variants = ((0, 0, 0), (12345, 0, 0), (0, 2, 1), (777, 3, 2))
    n = 0
    for it in range(150):
        T = int(rng.integers(1, 1500))
        t = rng.integers(0, 4, T).astype(np.uint8)
        rate = rng.choice([0.0, 0.01, 0.05, 0.12, 0.3])
        q = _mutate(rng, t, rate, rate / 2, rate / 2) if rng.random() < 0.9 else rng.integers(0, 4, int(rng.integers(1, 1500))).astype(np.uint8)
        if len(q) == 0:
            continue
        n += _check(host, checker, q, t, ks, (8, 16, 48), variants)
    assert n > 5000


def test_edges_of_the_window(host, checker):
    """lengths around the word size, bands that start above the matrix by every offset mod 32, one-column targets"""
    rng = np.random.default_rng(2)
    ks = lambda ed, m, T: sorted({ed, ed + 1, ed + 31, ed + 32, ed + 33, max(m, T)})
    for m in (1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129):
        for T in (1, 2, 31, 32, 33, 64, 65, 100, 129):
            q = rng.integers(0, 4, m).astype(np.uint8)
            t = rng.integers(0, 4, T).astype(np.uint8)
            _check(host, checker, q, t, ks, (16, 48), ((0, 0, 0), (99, 1, 1)))
            _check(host, checker, q, q[:T] if T <= m else np.concatenate([q, t[: T - m]]), ks, (16, 48), ((0, 0, 0),))


def test_config4_like_pairs_and_what_the_trims_save(host, checker):
    """10 kb at 4 / 4 / 4 % (BASELINE config 4): exact at the level's K and at K = distance; the window falls well below the
    static band's 41 words on average"""
    from edlib_amd import synth
    qs, ts = synth.mutated_pairs(6, 10000, seed=12349, sub=0.04, ins=0.04, dele=0.04)
    code = np.zeros(256, np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    for i in range(len(qs)):
        q, t = code[np.frombuffer(qs[i].tobytes(), np.uint8)], code[np.frombuffer(ts[i].tobytes(), np.uint8)]
        ed = checker.align(qs[i].tobytes(), ts[i].tobytes(), "NW", "distance", -1)["editDistance"]
        for K in (1280, ed, ed - 1):
            ws = ctypes.c_int(0)
            r = host.lanepair_host_nw(q.tobytes(), len(q), t.tobytes(), len(t), K, 48, 0, 0, 0, ctypes.byref(ws))
            assert (r == ed) if ed <= K else (r > K)
            if K == 1280:
                assert ws.value / ((len(t) + 31) // 32) < 30.0         # 41 words without the trims
