"""A CPU model of the WAVE-LEVEL band logic of scan_reads_banded_kernel (edlib_amd/csrc/reads_kernels.hip), test
infrastructure only: the rules that decide how many 32-bit words of the column a wave computes, restated with Python
integers as bit vectors so that they can be checked against the oracle without a GPU.

What is modelled (and nothing else -- no LDS, no M0, no registers):
  * lanes of a wave share ONE band height nw (words), taken from the ladder of the group (every height up to 8 words;
    1, 2, 3, 4, 6, 8, 12, 16, 24, 32 above): band_height_up / band_height_down;
  * the column step on the first nw words only (carry and shifts cut at the band's bottom, HW: row -1 is all zeros);
  * checkpoints every 4 columns at one word, 8 at two, 16 above, aligned to blocks of 16 columns:
        grow   when ANY lane has  S <= k + c - 1   (S = computed score of the band's bottom row, k = the lane's best so far)
        shrink when ALL lanes pass the (A + B - span) / 2 bounds of band_quad;
  * a grown word enters as "+1 per row"; the bottom query row is followed only at full height;
  * k-doubling: pass 1 with every lane's threshold capped at kcap, lanes without a hit rerun uncapped.
The model returns, per lane, the best score and the columns attaining it -- which must equal the reference's HW answer.
"""

HEIGHTS_LONG = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)


def height_ok(nwd, h):
    return 1 <= h <= nwd and (nwd <= 8 or h in HEIGHTS_LONG or h == nwd)


def height_up(nwd, h):
    n = h + 1
    while n < nwd and not height_ok(nwd, n):
        n += 1
    return n


def height_down(nwd, h):
    n = h - 1
    while n > 1 and not height_ok(nwd, n):
        n -= 1
    return n


def group_words(m):
    w = (m + 31) // 32
    if w <= 8:
        return w
    return 12 if w <= 12 else 16 if w <= 16 else 24 if w <= 24 else 32


def popc(x):
    return bin(x).count("1")


def _score(pv, mv, rows):
    """computed score of row rows-1: the vertical deltas above it (HW: row -1 is 0)"""
    mask = (1 << rows) - 1
    return popc(pv & mask) - popc(mv & mask)


class Wave:
    """`queries`: list of byte strings of one word-count group; `nwd`: the group's words."""

    def __init__(self, queries, nwd, alphabet):
        self.q = queries
        self.nwd = nwd
        self.m = [len(q) for q in queries]
        self.peq = []
        for q in queries:
            rows = {}
            for s in alphabet:
                v = 0
                for i, ch in enumerate(q):
                    if ch == s:
                        v |= 1 << i
                rows[s] = v                       # rows at or past the query end stay 0 (the padding of the group)
            self.peq.append(rows)

    def scan(self, target, kinit, kcap=None, log=None):
        """one pass over the whole target (one segment, no warm-up); kinit: per-lane thresholds.
        Returns per lane (best, [columns]) -- best == threshold and no columns when nothing was found."""
        nwd, n = self.nwd, len(self.q)
        full = (1 << (32 * nwd)) - 1
        pv = [full] * n
        mv = [0] * n
        best = [min(k, kcap) if kcap is not None else k for k in kinit]
        pos = [[] for _ in range(n)]
        nw = nwd
        T = len(target)
        for col in range(T):
            mask = (1 << (32 * nw)) - 1
            sym = target[col]
            for l in range(n):
                eq = self.peq[l].get(sym, 0) & mask
                p, mm = pv[l] & mask, mv[l] & mask
                xv = eq | mm
                xh = ((((eq & p) + p) & mask) ^ p) | eq
                ph = (mm | (~(xh | p))) & mask
                mh = p & xh
                ph = (ph << 1) & mask                         # HW: zero shifted in at row -1
                mh = (mh << 1) & mask
                pv[l] = (mh | (~(xv | ph))) & mask
                mv[l] = ph & xv
                if nw == nwd:                                  # bottom row in the band: follow its score
                    sc = _score(pv[l], mv[l], self.m[l])
                    if sc <= best[l]:
                        if sc < best[l]:
                            best[l] = sc
                            pos[l] = []
                        pos[l].append(col)
            q = (col & 15) >> 2                                # quad inside the 16-column block
            if (col & 3) != 3:
                continue
            # ---- checkpoint after the quad (band_quad)
            if nw == 1:
                if nwd > 1 and any(_score(pv[l], mv[l], 32) <= best[l] + 3 for l in range(n)):
                    nw = self._grow(pv, mv, 1, 2)
            elif nw == 2:
                if q & 1:
                    s1 = [_score(pv[l], mv[l], 32) for l in range(n)]
                    s2 = [_score(pv[l], mv[l], 64) for l in range(n)]
                    if nwd > 2 and any(s2[l] <= best[l] + 7 for l in range(n)):
                        nw = self._grow(pv, mv, 2, 3)
                    else:
                        keep = False
                        for l in range(n):
                            k2 = 2 * best[l] + 9
                            sa, sb, sc = (_score(pv[l], mv[l], 40), _score(pv[l], mv[l], 48), _score(pv[l], mv[l], 56))
                            if (s1[l] <= best[l] + 4 or s1[l] + sa <= k2 or sa + sb <= k2 or sb + sc <= k2 or sc + s2[l] <= k2):
                                keep = True
                                break
                        if not keep:
                            nw = 1
            elif q == 3:
                cum = [[_score(pv[l], mv[l], 32 * (i + 1)) for i in range(nw)] for l in range(n)]
                if nw < nwd and any(cum[l][nw - 1] <= best[l] + 15 for l in range(n)):
                    nw = self._grow(pv, mv, nw, height_up(nwd, nw))
                else:
                    dn = height_down(nwd, nw)
                    keep = False
                    for l in range(n):
                        c = cum[l]
                        if c[dn - 1] <= best[l] + 16 or any(c[i - 1] + c[i] <= 2 * best[l] + 34 for i in range(dn, nw)):
                            keep = True
                            break
                    if not keep:
                        nw = dn
            if log is not None:
                log.append(nw)
        return [(best[l], pos[l]) for l in range(n)]

    @staticmethod
    def _grow(pv, mv, old, new):
        add = ((1 << (32 * (new - old))) - 1) << (32 * old)     # "+1 per row", edlib.cpp:605-608
        low = (1 << (32 * old)) - 1
        for l in range(len(pv)):
            pv[l] = (pv[l] & low) | add
            mv[l] = mv[l] & low
        return new

    def solve(self, target, kfirst=8):
        """pass 1 at min(m, kfirst), pass 2 uncapped for the lanes without a hit: (best, columns) per lane, best = None
        when even the full threshold m finds nothing"""
        kinit = list(self.m)
        first = self.scan(target, kinit, kcap=kfirst)
        out = list(first)
        todo = [l for l in range(len(self.q)) if not first[l][1] and self.m[l] > kfirst]
        if todo:
            sub = Wave([self.q[l] for l in todo], self.nwd, [])
            sub.peq = [self.peq[l] for l in todo]
            second = sub.scan(target, [self.m[l] for l in todo], kcap=None)
            for l, r in zip(todo, second):
                out[l] = r
        return [(b, p) if p else (None, []) for b, p in out]
