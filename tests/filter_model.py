"""CPU model (numpy) of the piece filter + window verification that long HW reads run through on the GPU
(edlib_amd/csrc/long_reads.hip), used by tests/test_filter_model.py to check the ARGUMENT of that path against the
reference semantics without a GPU:

  * plan_level() is the same arithmetic as the C++ (parts, rows per piece, piece threshold, the "still specific" rule);
  * candidates() = 16-column blocks holding a column where a piece's semi-global score is <= kp (what the filter scan of
    the reads-per-lane kernel lists);
  * windows() = merged ranges of end columns the candidates allow for the whole query;
  * verify() = semi-global scores of the whole query restarted m + k columns before a window, only the window's own
    columns reported (what kernel W computes for a unit with `skip`).

bottom_row() is the textbook DP (reference semantics, SURVEY.md §8a-1: D[i][-1] = i, row -1 = 0 for HW)."""
import numpy as np

PIECE_ROWS = 256
KP_MAX = 56


def bottom_row(q, t, rows=None):
    """D[rows][j] for every target column j (HW: free start, free end), rows default len(q)"""
    q = np.asarray(q); t = np.asarray(t)
    m = len(q) if rows is None else rows
    idx = np.arange(1, m + 1)
    col = idx.copy()                                   # column -1: D[i][-1] = i
    out = np.empty(len(t), dtype=np.int64)
    for j, c in enumerate(t):
        diag = np.concatenate(([0], col[:-1])) + (q[:m] != c)
        tmp = np.minimum(diag, col + 1)
        # vertical moves: D[i] = min over i' <= i of tmp[i'] + (i - i'), and row -1 (value 0) above everything
        col = np.minimum(np.minimum.accumulate(tmp - idx) + idx, idx)
        out[j] = col[m - 1]
    return out


def plan_level(m, k, kp_max=KP_MAX, piece_rows=PIECE_ROWS, min_rows=48):
    p = k // (kp_max + 1) + 1
    part = m // p
    rows = min(piece_rows, part)
    kp = k // p
    return {"p": p, "part": part, "rows": rows, "kp": kp, "ok": rows >= min_rows and 4 * kp <= rows}


def candidates(q, t, plan):
    """[(end row of the piece, block)]"""
    out = []
    for i in range(plan["p"]):
        a = i * plan["part"]
        sc = bottom_row(q[a:a + plan["rows"]], t)
        for b in sorted(set((np.nonzero(sc <= plan["kp"])[0] // 16).tolist())):
            out.append((a + plan["rows"], b))
    return out


def windows(cands, m, k, T):
    w = []
    for end, b in cands:
        below = m - end
        lo, hi = 16 * b + below - k, 16 * b + 15 + below + k
        if hi < 0 or lo > T - 1:
            continue
        w.append([max(lo, 0), min(hi, T - 1)])
    w.sort()
    merged = []
    for lo, hi in w:
        if merged and lo <= merged[-1][1] + m + k:
            merged[-1][1] = max(merged[-1][1], hi)
        else:
            merged.append([lo, hi])
    return merged


def verify(q, t, win, k):
    """(best, [columns]) over the windows: restarted scans, columns of the warm-up are not reported"""
    m = len(q)
    best, cols = None, []
    for lo, hi in win:
        start = max(0, lo - m - k)
        sc = bottom_row(q, t[start:hi + 1])[lo - start:]
        for j, s in enumerate(sc):
            if s <= k and (best is None or s <= best):
                if best is None or s < best:
                    best, cols = int(s), []
                cols.append(lo + j)
    return best, cols


def align_hw(q, t, k_user=-1, k0=8, **plan_args):
    """(editDistance, endLocations) of the ladder, or None when a level is not filterable (handed back)"""
    m, T = len(q), len(t)
    kmax = m if (k_user < 0 or k_user > m) else k_user
    k = min(kmax, k0)
    while True:
        plan = plan_level(m, k, **plan_args)
        if not plan["ok"]:
            return None
        best, cols = verify(q, t, windows(candidates(q, t, plan), m, k, T), k)
        if best is not None:
            return best, cols
        if k >= kmax:
            return -1, []
        k = min(kmax, 2 * k)
