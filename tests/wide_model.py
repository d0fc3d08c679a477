"""A CPU model of scan_pairs_wide_kernel (edlib_amd/csrc/wide_kernels.hip), test infrastructure only: the DATA path of
the strip pipeline restated over Python integers -- which columns a strip of L blocks runs, what it takes from the strip
above (the 2-bit codes of its bottom row's horizontal deltas in 16-column granules, +1 per column beyond the upper strip's
last column, the absolute score it starts from), the codes folded into block scores every 16 steps, the NW score decode
and the last-column dump of a Hirschberg half.  The hand-off PROTOCOL (tags, polling, slots) is not modelled: strips run
one after the other here.  L (words per strip) is a parameter so that multi-strip cases stay small; the kernel has L = 64
lanes of one 32-row word (WB = 32).

    D = bandT - m,  p = (K - |D|) >> 1,  [dmin, dmax] = [min(0, D) - p, max(0, D) + p]          (NW; SHW / HW: everything)
    strip s: rows [WB L s, WB L (s + 1)), columns [max(0, WB L s + dmin), min(T - 1, WB L s + WB L - 1 + dmax)]
    granule of 16 columns: bit 15 - (c & 15) = "hout of column c is +1", bit 31 - (c & 15) = "is -1"
"""

WB = 32
M64 = (1 << WB) - 1                                       # (the word mask; the name is the 64-row model's)


def popc(x):
    return bin(x).count("1")


def geom(mode, m, T, bandT, K):
    if mode == 1 and bandT < 0:                          # SHW inside the band of threshold K: |i - j| <= K
        return -K, K
    if mode == 2 and bandT < 0:                          # HW inside the band of threshold K: starts in [0, T - m + K], K diagonals either side
        return -K, max(0, T - m) + 2 * K
    if mode != 0:
        return -(1 << 40), 1 << 40
    D = (bandT if bandT > 0 else T) - m
    p = (K - abs(D)) >> 1
    return min(0, D) - p, max(0, D) + p


def strip_range(s, L, T, dmin, dmax):
    r0 = WB * L * s
    c0 = min(max(0, r0 + dmin), T)
    c1 = min(T - 1, r0 + WB * L - 1 + dmax)
    return c0, c1


def block_step(pv, mv, eq, hin):
    """reference calculateBlock (edlib.cpp:412-447)"""
    hneg = 1 if hin < 0 else 0
    xv = eq | mv
    eq2 = eq | hneg
    xh = ((((eq2 & pv) + pv) & M64) ^ pv) | eq2
    ph = mv | (~(xh | pv) & M64)
    mh = pv & xh
    hout = ((ph >> (WB - 1)) & 1) - ((mh >> (WB - 1)) & 1)
    phu, mhu = ph, mh
    ph = (ph << 1) & M64
    mh = (mh << 1) & M64
    if hin < 0:
        mh |= 1
    elif hin > 0:
        ph |= 1
    return (mh | ~(xv | ph)) & M64, ph & xv, hout, phu, mhu


def wide_scan(q, t, mode, K, L=64, bandT=0, skip=0, pos_cap=1 << 30):
    """(score, count, last, positions, dump) as the kernel leaves them.  NW: score exact iff <= K (None: K < |D|);
    dump = {block: (P, M, score)} of the blocks alive at column T - 1."""
    m, T = len(q), len(t)
    nb = (m + WB - 1) // WB
    nstrips = (nb + L - 1) // L
    if mode == 0 and K < abs((bandT if bandT > 0 else T) - m):
        return None
    dmin, dmax = geom(mode, m, T, bandT, K)
    peq = {}
    for sy in set(t):
        v = 0
        for i, ch in enumerate(q):
            if ch == sy:
                v |= 1 << i
        peq[sy] = [(v >> (WB * b)) & M64 for b in range(nb)]
    zero = [0] * nb
    sh = (m - 1) & (WB - 1)
    row_above = 0 if mode == 2 else 1                     # hin at row -1
    stream = {}                                           # granules of the previous strip: group -> 32-bit word
    start_score = None
    prev_c1 = None
    score = count = None
    last = -1
    positions = []
    dump = {}
    best, cnt = K, 0
    for s in range(nstrips):
        c0, c1 = strip_range(s, L, T, dmin, dmax)
        if c0 > c1:
            break
        nc0, nc1 = strip_range(s + 1, L, T, dmin, dmax)
        next_live = s + 1 < nstrips and nc0 <= nc1
        r0 = WB * L * s
        nbS = min(L, nb - s * L)
        if c0 == 0:
            top = r0
        else:
            assert start_score is not None, "the strip above never passed column c0 - 1"
            top = start_score
        P = [M64] * nbS
        Mv = [0] * nbS
        bscore = [top + WB * (l + 1) for l in range(nbS)]
        accP = [0] * nbS                                   # the houts since the last fold, newest at bit 0
        accM = [0] * nbS
        sc = top + (m - r0)
        out_stream = {}
        out_start = None

        def hin_at(c):
            if s == 0:
                return row_above
            if c > prev_c1:
                return 1
            w = stream[c >> 4]
            return ((w >> (15 - (c & 15))) & 1) - ((w >> (31 - (c & 15))) & 1)

        for c in range(c0, c1 + 1):
            eqs = peq.get(t[c], zero)
            h = hin_at(c)
            for l in range(nbS):
                b = s * L + l
                P[l], Mv[l], h, phu, mhu = block_step(P[l], Mv[l], eqs[b], h)
                accP[l] = (accP[l] << 1) | (1 if h > 0 else 0)
                accM[l] = (accM[l] << 1) | (1 if h < 0 else 0)
                if mode != 0 and b == nb - 1:
                    sc += ((phu >> sh) & 1) - ((mhu >> sh) & 1)
                    if sc <= best and c >= skip:
                        if sc < best:
                            best, cnt = sc, 0
                            positions = []
                        if cnt < pos_cap:
                            positions.append(c)
                        cnt += 1
                        last = c
            # lane L-1 has finished column c (the kernel's fold / flush schedule, in this lane's own time)
            lastl = nbS - 1
            if next_live and c == nc0 - 1:
                out_start = bscore[lastl] + popc(accP[lastl]) - popc(accM[lastl])
            if (c & 15) == 15 or c == c1:
                if next_live:
                    up = 15 - (c & 15)
                    out_stream[c >> 4] = ((accP[lastl] << up) & 0xffff) | ((accM[lastl] << up) << 16)
                for l in range(nbS):
                    assert accP[l] < (1 << 16) and accM[l] < (1 << 16)
                    bscore[l] += popc(accP[l]) - popc(accM[l])
                    accP[l] = accM[l] = 0
        if c1 == T - 1:
            for l in range(nbS):
                dump[s * L + l] = (P[l], Mv[l], bscore[l])
        if s == nstrips - 1:
            if mode == 0:
                if c1 == T - 1:
                    l = nb - 1 - s * L
                    below = 0 if sh == WB - 1 else (M64 << (sh + 1)) & M64
                    score = bscore[l] - popc(P[l] & below) + popc(Mv[l] & below)
                    count, last = 1, T - 1
            else:
                score, count = (best if cnt > 0 else -1), cnt
        stream, start_score, prev_c1 = out_stream, out_start, c1
    return score, count, last, positions, dump
