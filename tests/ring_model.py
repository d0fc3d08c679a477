"""A CPU model of the BAND of scan_pairs_ring_kernel (edlib_amd/csrc/pair_kernels.hip, MODE 0: NW with a fixed
threshold K), test infrastructure only: which 64-row blocks are alive at a column, what a block takes from above when
it starts and when its upstream has left the band, and how the score is decoded -- restated over Python integers so
that the band formula and ring_max_k can be checked against the oracle without a GPU.

    D = T - m,  p = (K - |D|) >> 1,  diagonals d = j - i in [dmin, dmax] = [min(0, D) - p, max(0, D) + p]
    block b lives for columns [max(0, 64 b + dmin), min(T - 1, 64 b + 63 + dmax)]
    a block starts as "+1 per row" below the bottom score of the block above at the previous column (64 b at column 0)
    a block whose upstream is outside the band takes hin = +1 (block 0: always +1, the NW top row)
    score = bottom score of the last block at column T - 1 minus the vertical deltas below row m - 1

Anti-diagonal schedule of the ring: block b updates column t - b at step t, on ring lane b % G; ring_fits() says
whether consecutive tenants of a lane never overlap in steps (the condition ring_max_k(G) has to guarantee).
"""

M64 = (1 << 64) - 1

# RH: rows per ring lane -- 64 (scan_pairs_ring_kernel) or 32 (scan_pairs_ring32_kernel, ring32_kernels.hip: the same
# rules on 32-row words).  geom = (dmin, dmax): a band WIDER than the unit's own, as the ring32 kernel uses when the
# units of a wave share one geometry (the extremes over the wave, accepted while dmax - dmin <= RH (G - 2)).


def ring_max_k(G, RH=64):
    return RH * (G - 2)


def band(m, T, K):
    D = T - m
    p = (K - abs(D)) >> 1
    return min(0, D) - p, max(0, D) + p


def lives(m, T, K, RH=64, geom=None):
    nb = (m + RH - 1) // RH
    dmin, dmax = geom if geom else band(m, T, K)
    out = []
    for b in range(nb):
        f, l = max(0, RH * b + dmin), min(T - 1, RH * b + RH - 1 + dmax)
        out.append((f, l) if f <= l else None)
    return out


def ring_fits(m, T, K, G, RH=64, geom=None):
    """block b + G starts (step first + b + G) strictly after the step at which block b closes (last + b + 1 is its
    closing event; the kernel lets the next tenant start in that same step)"""
    lv = lives(m, T, K, RH, geom)
    for b in range(len(lv) - G):
        if lv[b] is None or lv[b + G] is None:
            continue
        if lv[b][1] + b + 1 > lv[b + G][0] + b + G:
            return False
    return True


def popc(x):
    return bin(x).count("1")


def banded_nw(q, t, K, RH=64, geom=None):
    """the model's score: exact when the true distance is <= K, some value > K otherwise (None: K < |T - m|)"""
    m, T = len(q), len(t)
    if K < abs(T - m):
        return None
    M64 = (1 << RH) - 1
    nb = (m + RH - 1) // RH
    peq = {}
    for s in set(t):
        v = 0
        for i, ch in enumerate(q):
            if ch == s:
                v |= 1 << i
        peq[s] = [(v >> (RH * b)) & M64 for b in range(nb)]
    zero = [0] * nb
    lv = lives(m, T, K, RH, geom)
    P = [M64] * nb
    Mv = [0] * nb
    bscore = [0] * nb                                  # bottom score of block b after its last update
    alive_prev = [False] * nb
    for j in range(T):
        eqs = peq.get(t[j], zero)
        hout_prev, prev_alive = 1, False                # row -1: +1 per column (NW)
        new_bottom = list(bscore)
        alive = [lv[b] is not None and lv[b][0] <= j <= lv[b][1] for b in range(nb)]
        for b in range(nb):
            if not alive[b]:
                prev_alive = False
                continue
            if not alive_prev[b]:                       # the block starts: "+1 per row" below the block above
                P[b], Mv[b] = M64, 0
                # (block 0 always starts at column 0; a later block starts while the block above is alive, so
                # bscore[b - 1] is that block's computed bottom at column j - 1)
                above = RH * b if j == 0 else bscore[b - 1]
                cur = above + RH
            else:
                cur = bscore[b]
            hin = 1 if (b == 0 or not prev_alive) else hout_prev
            eq = eqs[b]
            pv, mv = P[b], Mv[b]
            hneg = 1 if hin < 0 else 0
            xv = eq | mv
            eq2 = eq | hneg
            xh = ((((eq2 & pv) + pv) & M64) ^ pv) | eq2
            ph = mv | (~(xh | pv) & M64)
            mh = pv & xh
            hout = ((ph >> (RH - 1)) & 1) - ((mh >> (RH - 1)) & 1)
            ph = (ph << 1) & M64
            mh = (mh << 1) & M64
            if hin < 0:
                mh |= 1
            elif hin > 0:
                ph |= 1
            P[b] = (mh | ~(xv | ph)) & M64
            Mv[b] = ph & xv
            new_bottom[b] = cur + hout
            hout_prev, prev_alive = hout, True
        bscore = new_bottom
        alive_prev = alive
    last = nb - 1
    if lv[last] is None or lv[last][1] != T - 1:
        return K + 1                                     # the last block is not alive at the stop column: above K
    sh = (m - 1) & (RH - 1)
    below = 0 if sh == RH - 1 else (M64 << (sh + 1)) & M64
    return bscore[last] - popc(P[last] & below) + popc(Mv[last] & below)
