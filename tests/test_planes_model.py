"""CPU model of the two-plane column store and of the walk over it (edlib_amd/csrc/pair_kernels.hpp StoreEntry,
pair_kernels.hip traceback_kernel) against the textbook DP with the reference's move order up > left > diagonal
(obtainAlignmentTraceback, edlib.cpp:942-1141).  The block update is calculateBlock (edlib.cpp:412-447) on Python
integers; what is checked is the claim the GPU path rests on: x = Pv | Ph, y = ~Pv & (Ph | Xh) answer the three
questions of a cell (up, left, match) without scores."""
import random

M64 = (1 << 64) - 1


def planes(q, t, sigma):
    m, T = len(q), len(t)
    nb = (m + 63) // 64
    peq = [[0] * nb for _ in range(sigma)]
    for s in range(sigma):
        for i in range(nb * 64):
            if i >= m or q[i] == s:                     # padding rows are wildcards (buildPeq, edlib.cpp:373-375)
                peq[s][i // 64] |= 1 << (i % 64)
    pv, mv = [M64] * nb, [0] * nb
    store = [[None] * T for _ in range(nb)]
    for c in range(T):
        hin = 1                                          # NW: row -1 grows by one per column (edlib.cpp:779)
        for b in range(nb):
            eq = peq[t[c]][b]
            xv = eq | mv[b]
            if hin < 0:
                eq |= 1
            xh = ((((eq & pv[b]) + pv[b]) & M64) ^ pv[b]) | eq
            ph = mv[b] | (~(xh | pv[b]) & M64)
            mh = pv[b] & xh
            hout = 1 if ph >> 63 else (-1 if mh >> 63 else 0)
            phs, mhs = (ph << 1) & M64, (mh << 1) & M64
            if hin < 0:
                mhs |= 1
            if hin > 0:
                phs |= 1
            pv[b] = mhs | (~(xv | phs) & M64)
            mv[b] = phs & xv
            store[b][c] = (pv[b] | ph, (~pv[b] & M64) & (ph | xh))
            hin = hout
    return store


def walk(store, m, T):
    """traceback_kernel without the batching: ops from the end of the alignment to its start"""
    ops, r, c = [], m - 1, T - 1
    while True:
        x, y = store[r >> 6][c]
        bit = r & 63
        if (x & ~y) >> bit & 1:                          # up: INSERT
            ops.append(1)
            if r == 0:
                return ops + [2] * (c + 1)
            r -= 1
            continue
        if (x & y) >> bit & 1:                           # left: DELETE
            ops.append(2)
            c -= 1
            if c < 0:
                return ops + [1] * (r + 1)
            continue
        ops.append(0 if (y >> bit) & 1 else 3)           # diagonal: MATCH / MISMATCH
        c -= 1
        if c < 0:
            return ops + [1] * r
        if r == 0:
            return ops + [2] * (c + 1)
        r -= 1


def textbook(q, t):
    m, T = len(q), len(t)
    D = [[0] * (T + 1) for _ in range(m + 1)]
    for i in range(m + 1):
        D[i][0] = i
    for j in range(T + 1):
        D[0][j] = j
    for i in range(1, m + 1):
        for j in range(1, T + 1):
            D[i][j] = min(D[i - 1][j - 1] + (q[i - 1] != t[j - 1]), D[i - 1][j] + 1, D[i][j - 1] + 1)
    ops, i, j = [], m, T
    while i > 0 or j > 0:
        if i > 0 and D[i - 1][j] + 1 == D[i][j]:
            ops.append(1); i -= 1
        elif j > 0 and D[i][j - 1] + 1 == D[i][j]:
            ops.append(2); j -= 1
        else:
            ops.append(0 if D[i - 1][j - 1] == D[i][j] else 3); i -= 1; j -= 1
    return ops


def test_planes_answer_the_walk():
    rng = random.Random(11)
    for it in range(120):
        sigma = rng.choice([2, 3, 4])
        m = rng.choice([1, 5, 63, 64, 65, 100, 128, 129, 150])
        q = [rng.randrange(sigma) for _ in range(m)]
        t = []
        for ch in q:                                     # a mutated copy, so that paths cross block rows with indels around
            r = rng.random()
            if r < 0.05:
                continue
            if r < 0.10:
                t.append(rng.randrange(sigma))
            t.append(ch if r >= 0.15 else rng.randrange(sigma))
        t = t or [0]
        if it % 5 == 0:
            t = [rng.randrange(sigma) for _ in range(rng.choice([1, 7, 90]))]
        assert walk(planes(q, t, sigma), len(q), len(t)) == textbook(q, t), (it, m, len(t))


def walk_diagonals(store, m, T, L=32):
    """traceback32_kernel (ring32_kernels.hip) without the lanes: L cells of the current diagonal per trip, the run of
    diagonal moves up to the first cell where an indel move is possible, then up-moves (as many as the "up" plane shows
    in the cell's word) or one left move; the matrix boundary leaves a tail.  Ops from the end of the alignment to its start."""
    def xy(r, c):
        x, y = store[r >> 6][c]
        return (x >> (r & 63)) & 1, (y >> (r & 63)) & 1
    ops, r, c = [], m - 1, T - 1
    while True:
        j = L
        for i in range(L):
            ri, ci = r - i, c - i
            if ri < 0 or ci < 0 or xy(ri, ci)[0]:
                j = i
                break
        ops += [0 if xy(r - i, c - i)[1] else 3 for i in range(j)]
        r -= j; c -= j
        if j == L:
            continue
        if r < 0 or c < 0:
            return ops + ([1] * (r + 1) if c < 0 else [2] * (c + 1))
        # the cell that stopped the run: up-moves inside its 32-row word, else one left move
        ups = 0
        while ups <= (r & 31):
            x, y = xy(r - ups, c)
            if not (x and not y):
                break
            ups += 1
        if ups:
            ops += [1] * ups; r -= ups
        else:
            x, y = xy(r, c)
            assert x and y
            ops.append(2); c -= 1
        if r < 0 or c < 0:
            if r < 0 and c < 0:
                return ops
            return ops + ([1] * (r + 1) if c < 0 else [2] * (c + 1))


def test_diagonal_runs_walk_like_the_reference():
    rng = random.Random(12)
    for it in range(160):
        sigma = rng.choice([2, 3, 4])
        m = rng.choice([1, 5, 31, 32, 33, 63, 64, 65, 100, 128, 129, 150, 200])
        q = [rng.randrange(sigma) for _ in range(m)]
        t = []
        for ch in q:
            r = rng.random()
            if r < 0.05:
                continue
            if r < 0.10:
                t.append(rng.randrange(sigma))
            t.append(ch if r >= 0.15 else rng.randrange(sigma))
        if it % 7 == 3:                                  # a long gap on either side: runs of up / left moves across words
            cut = rng.randrange(len(t) + 1)
            t = t[:cut] + [rng.randrange(sigma) for _ in range(rng.choice([40, 70]))] + t[cut:]
        if it % 7 == 5 and len(t) > 80:
            cut = rng.randrange(len(t) - 70)
            t = t[:cut] + t[cut + 70:]
        t = t or [0]
        if it % 5 == 0:
            t = [rng.randrange(sigma) for _ in range(rng.choice([1, 7, 90]))]
        for L in (32, 64):
            assert walk_diagonals(planes(q, t, sigma), len(q), len(t), L) == textbook(q, t), (it, m, len(t), L)
