"""CPU: the argument behind edlib_amd/csrc/long_reads.hip (piece filter + window verification of long HW reads), as a numpy
model (tests/filter_model.py) against the reference semantics -- every column with D[m][j] <= k lies in a window, the
restarted scans are exact where they are <= k, and the ladder's answer is the reference's answer (oracle on the same bytes)."""
import numpy as np
import pytest

import filter_model as F

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _case(rng, m, T, rate, sigma=4):
    t = _ACGT[rng.integers(0, sigma, T)]
    s = int(rng.integers(0, max(1, T - m)))
    q = t[s:s + m].copy()
    for _ in range(int(rate * m)):
        p = int(rng.integers(0, len(q)))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            q[p] = _ACGT[rng.integers(0, sigma)]
        elif kind == 1 and len(q) > 1:
            q = np.delete(q, p)
        else:
            q = np.insert(q, p, _ACGT[rng.integers(0, sigma)])
    return q, t


@pytest.mark.parametrize("seed", range(6))
def test_every_column_within_k_is_inside_a_window_and_exact_there(seed):
    """small pieces (32 rows, thresholds up to 5) so that many parts, windows and levels occur on small matrices"""
    rng = np.random.default_rng(seed)
    args = dict(kp_max=5, piece_rows=32, min_rows=12)
    for _ in range(12):
        m, T = int(rng.integers(40, 260)), int(rng.integers(300, 1500))
        q, t = _case(rng, m, T, float(rng.choice([0.0, 0.02, 0.06, 0.12])))
        m = len(q)
        full = F.bottom_row(q, t)
        for k in (0, 3, 9, 17, 30):
            plan = F.plan_level(m, k, **args)
            if not plan["ok"]:
                continue
            win = F.windows(F.candidates(q, t, plan), m, k, T)
            inside = np.zeros(T, dtype=bool)
            for lo, hi in win:
                inside[lo:hi + 1] = True
            assert inside[full <= k].all(), (seed, m, T, k, plan)
            for lo, hi in win:
                start = max(0, lo - m - k)
                re = F.bottom_row(q, t[start:hi + 1])[lo - start:]
                tr = full[lo:hi + 1]
                assert (re >= tr).all() and (re[tr <= k] == tr[tr <= k]).all()


@pytest.mark.parametrize("seed", range(4))
def test_ladder_answers_equal_the_reference(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    args = dict(kp_max=5, piece_rows=32, min_rows=12)
    done = 0
    for _ in range(14):
        m, T = int(rng.integers(40, 300)), int(rng.integers(300, 2000))
        q, t = _case(rng, m, T, float(rng.choice([0.0, 0.03, 0.08])))
        for k_user in (-1, 4, 25):
            got = F.align_hw(q, t, k_user=k_user, k0=3, **args)
            if got is None:
                continue
            want = oracle.align(q.tobytes(), t.tobytes(), "HW", "distance", k_user)
            assert got[0] == want["editDistance"], (seed, m, T, k_user)
            if got[0] >= 0:
                assert got[1] == [e for e in want["endLocations"] if e >= 0]
            done += 1
    assert done >= 20


def test_plan_with_the_library_constants():
    """256-row pieces, thresholds up to 56: the fewest parts whose threshold fits, pieces never overlap, handed back when a
    quarter of the piece would be errors"""
    for m in (257, 300, 512, 1025, 4096, 10000, 100000):
        for k in (0, 8, 56, 57, 128, 700, 1500, m // 3, m):
            p = F.plan_level(m, k)
            assert p["kp"] <= 56 and p["p"] * p["part"] <= m and p["rows"] <= min(256, p["part"])
            assert (p["p"] == 1) or (k // (p["p"] - 1) > 56)          # one part fewer would exceed the threshold cap
            if p["ok"]:
                assert 4 * p["kp"] <= p["rows"]
    assert F.plan_level(10000, 128)["p"] == 3 and F.plan_level(10000, 128)["kp"] == 42
    assert not F.plan_level(300, 200)["ok"] and F.plan_level(1025, 16)["p"] == 1


def test_real_constants_on_a_moderate_case(oracle):
    rng = np.random.default_rng(7)
    q, t = _case(rng, 700, 4000, 0.05)
    got = F.align_hw(q, t, k0=24)
    want = oracle.align(q.tobytes(), t.tobytes(), "HW", "distance", -1)
    assert got is not None and got[0] == want["editDistance"] and got[1] == want["endLocations"]
