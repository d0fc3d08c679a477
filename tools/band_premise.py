#!/usr/bin/env python3
"""the premise of a two-reads-per-lane kernel on 16-row words (v_pk_*_u16: DESIGN.md 3b) settled on the CPU.

Such a kernel pays only while a WAVE's band is ONE 16-row word: the band must grow when any of its reads has a computed
bottom-row score S <= k + c - 1 at a checkpoint (interval c columns: the sound rule of scan_reads_banded_kernel, band_quad).
This script computes S EXACTLY -- the first 16 (and, for comparison, 32) rows of the HW matrix of every read of a sample of
bench.py's config-2 batch against a stretch of its target, as bit vectors in numpy, one lane per read (the top rows of a
semi-global scan depend on nothing below them) -- and counts, per wave of 64 / 128 reads and per checkpoint interval, the
checkpoints at which NO read of the wave asks for growth.  A band that grew also needs columns to shrink back, so the numbers
are upper bounds on the share of columns a one-word band would cover.
    python tools/band_premise.py [--reads 20480] [--columns 400000] > profiles/r06_band_premise.json"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edlib_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20480)
    ap.add_argument("--columns", type=int, default=400000)
    a = ap.parse_args()
    target = synth.random_dna(12345, 5_000_000)
    rd = synth.illumina_reads(target, 1_000_000, m=150, seed=12346)["reads"][: a.reads]
    tcol = target[: a.columns]
    code = np.zeros(256, np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    q = code[rd]                                        # reads x 150
    t = code[tcol]
    out = {"sample": "%d reads of bench.py's config-2 batch x the first %d target columns" % (a.reads, a.columns), "rows": {}}
    for rows in (16, 32):
        # Peq of the first `rows` rows, one word per read
        peq = np.zeros((4, a.reads), np.uint64)
        for r in range(rows):
            for s in range(4):
                peq[s] |= (q[:, r] == s).astype(np.uint64) << np.uint64(r)
        mask = np.uint64((1 << rows) - 1)
        pv = np.full(a.reads, mask, np.uint64); mv = np.zeros(a.reads, np.uint64)
        waves = {64: a.reads // 64, 128: a.reads // 128}
        wmin = {w: np.zeros((n, a.columns), np.int8) for w, n in waves.items()}
        one = np.uint64(1)
        pc = np.array([bin(i).count("1") for i in range(65536)], np.int16)
        def popc(x):
            return pc[(x & np.uint64(0xffff)).astype(np.int64)] + pc[((x >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)]
        for j in range(a.columns):
            eq = peq[t[j]]
            xv = eq | mv
            xh = ((((eq & pv) + pv) & mask) ^ pv) | eq
            ph = mv | (~(xh | pv) & mask)
            mh = pv & xh
            ph = (ph << one) & mask                    # HW: row -1 is all zeros
            mh = (mh << one) & mask
            pv = mh | (~(xv | ph) & mask)
            mv = ph & xv
            s = (popc(pv) - popc(mv)).astype(np.int8)   # D[rows - 1][j]: the vertical deltas above the bottom row
            for w, n in waves.items():
                wmin[w][:, j] = s[: n * w].reshape(n, w).min(axis=1)
        res = {}
        for w in waves:
            for k in (4, 6, 8):
                for c in (1, 2, 4):
                    cp = wmin[w][:, c - 1 :: c]          # the score a checkpoint sees, every c columns
                    res["wave %d, k = %d, checkpoint every %d" % (w, k, c)] = round(float((cp > k + c - 1).mean()), 4)
            res["wave %d: mean / 1st percentile of the wave minimum" % w] = [round(float(wmin[w].mean()), 2), int(np.percentile(wmin[w], 1))]
        out["rows"][str(rows)] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
