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
    """scan_pairs_ring_kernel (RH = 64 H): tenants of a lane (RH + 1) G steps apart, a block's life RH + dmax - dmin steps
    (pair_kernels.hpp: ring_max_k; the whole-wave ring keeps a lane idle for its target ring's sake).  The rings of 32-row
    words (ring32_kernels.hip) keep the lane above a block idle until it closes: 32 (G - 2)."""
    if RH == 32:
        return RH * (G - 2)
    return (RH + 1) * (G - 1 if G == 64 else G) - RH


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


# ------------------------------------------------------------------------------------------------------------------
# The same scan LANE BY LANE: G ring lanes, block b on lane b % G at steps col + b, carries that travel one lane per
# step.  This restates what keeps the branch-light step of scan_pairs_ring_kernel correct, which the column-wise model
# above takes for granted:
#   * a lane outside its block's life SENDS +1 (and counts its score up by one per step);
#   * a block LISTENS to the lane above only while the block above is alive: from the step after it has taken that
#     block's last carry it takes +1 whatever arrives (`upstream_rule`) -- round 5.  Without the rule the lane above
#     has to stay idle until the block closes, which costs a ring lane: K <= 64 (G - 2) instead of 65 G - 64.
# Returns the score as the kernel decodes it (K + 1 when the last block is not alive at the stop column).

def ring_lanes_nw(q, t, K, G, RH=64, upstream_rule=True, geom=None, bottom_row=None):
    """geom = (dmin, dmax): a static band instead of the NW band of K (SHW inside [-K, K]: solveSemiGlobalUnits);
    bottom_row: a list that receives (column, D[m][column]) for every column at which the last block is at work -- what the
    SHW / HW modes of the kernel track (the row above the query is +1 per column here, as for NW and SHW)"""
    m, T = len(q), len(t)
    if geom is None and K < abs(T - m):
        return None
    W = (1 << RH) - 1
    nb = (m + RH - 1) // RH
    peq = {}
    for s in set(t):
        v = 0
        for i, ch in enumerate(q):
            if ch == s:
                v |= 1 << i
        peq[s] = [(v >> (RH * b)) & W for b in range(nb)]
    zero = [0] * nb
    dmin, dmax = geom if geom else band(m, T, K)
    first_col = lambda b: max(0, RH * b + dmin)
    last_col = lambda b: min(T - 1, RH * b + RH - 1 + dmax)
    NEVER = 1 << 60

    class Lane:
        pass
    lanes = []
    for rl in range(G):
        L = Lane()
        L.b, L.act, L.listen, L.P, L.M, L.bscore, L.carry, L.span = rl, False, True, W, 0, 0, 1, 0
        lanes.append(L)

    def arm(L, step):
        ok = L.b < nb and first_col(L.b) <= last_col(L.b)
        ts = first_col(L.b) + L.b if ok else NEVER
        L.span = last_col(L.b) + L.b - ts if ok else 0
        L.ev = ts - step if ok else NEVER
    for L in lanes:
        arm(L, 0)
    result = K + 1
    for step in range(T + nb + 1):
        x = [lanes[(rl - 1) % G].carry for rl in range(G)]
        up = [lanes[(rl - 1) % G].bscore for rl in range(G)]
        for rl, L in enumerate(lanes):
            if L.ev == 0 and L.act:
                te, tl = last_col(L.b - 1) + L.b + 1 if L.b > 0 else NEVER, last_col(L.b) + L.b
                if upstream_rule and L.listen and L.b > 0 and step == te and te <= tl:
                    L.listen = False                                   # the block above has sent its last carry
                    L.ev = tl + 1 - step
                else:                                                  # closing
                    if step - 1 - L.b == T - 1 and L.b == nb - 1:
                        sh = (m - 1) & (RH - 1)
                        below = 0 if sh == RH - 1 else (W << (sh + 1)) & W
                        result = L.bscore - popc(L.P & below) + popc(L.M & below)
                    L.act = False
                    L.b += G
                    L.listen = True
                    arm(L, step)
            if L.ev == 0 and not L.act:                                # the block starts with this step
                col = step - L.b
                L.P, L.M = W, 0
                above = RH * L.b if col == 0 else up[rl] - x[rl]
                L.bscore = above + RH
                L.act = True
                te, tl = last_col(L.b - 1) + L.b + 1 if L.b > 0 else NEVER, last_col(L.b) + L.b
                L.ev = L.span + 1
                if upstream_rule and L.b > 0 and te <= tl:
                    if te > step:
                        L.ev = te - step
                    else:                                              # (a band of one diagonal: it already has)
                        L.listen = False
        new_carry = []
        for rl, L in enumerate(lanes):
            L.ev -= 1
            if not L.act:
                new_carry.append(1)
                L.bscore += 1
                continue
            col = step - L.b
            hin = 1 if (L.b == 0 or not L.listen) else x[rl]
            eq = peq.get(t[col], zero)[L.b] if 0 <= col < T else 0
            pv, mv = L.P, L.M
            hneg = 1 if hin < 0 else 0
            xv = eq | mv
            eq2 = eq | hneg
            xh = ((((eq2 & pv) + pv) & W) ^ pv) | eq2
            ph = mv | (~(xh | pv) & W)
            mh = pv & xh
            hout = ((ph >> (RH - 1)) & 1) - ((mh >> (RH - 1)) & 1)
            ph = (ph << 1) & W
            mh = (mh << 1) & W
            if hin < 0:
                mh |= 1
            elif hin > 0:
                ph |= 1
            L.P = (mh | ~(xv | ph)) & W
            L.M = ph & xv
            L.bscore += hout
            new_carry.append(hout)
            if bottom_row is not None and L.b == nb - 1 and 0 <= col < T:
                sh = (m - 1) & (RH - 1)
                below = 0 if sh == RH - 1 else (W << (sh + 1)) & W
                bottom_row.append((col, L.bscore - popc(L.P & below) + popc(L.M & below)))
        for L, c in zip(lanes, new_carry):
            L.carry = c
    return result
