/*
 * oracle/edlib_oracle.c -- TEST INFRASTRUCTURE ONLY (see edlib_oracle.h).
 *
 * CPU restatement, in plain C99, of the algorithm of the reference hot path
 * /root/reference/edlib/src/edlib.cpp (Martinsos/edlib v1.2.6).  Every
 * function names the reference lines it follows.  It is written from the
 * algorithm, not transcribed: state is kept as structure-of-arrays columns,
 * the traceback decodes cells on demand instead of peeling bits incrementally,
 * and there is no C++.  Parity with the compiled reference is pinned by
 * tests/test_oracle.py (known answers + reference-generated fixtures + live
 * differential fuzz when oracle/_ref/libedlib_ref.so exists).
 *
 * Not restated: obtainAlignmentHirschberg (edlib.cpp:1231-1396); oracle_align
 * returns ORACLE_UNSUPPORTED where the reference would take that branch.
 */
#include "edlib_oracle.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t word_t;              /* edlib.cpp:15  Word            */
#define WBITS 64                      /* edlib.cpp:16  WORD_SIZE       */
#define TOPBIT ((word_t)1 << 63)      /* edlib.cpp:18  HIGH_BIT_MASK   */
#define REDUCE_PERIOD 2048            /* edlib.cpp:572,742 STRONG_REDUCE_NUM */

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int ceil_div(int a, int b) { return (a + b - 1) / b; }   /* edlib.cpp:453-455 */

/* ------------------------------------------------------------------ ivec */
typedef struct { int* v; int n, cap; } ivec;
static void ivec_push(ivec* a, int x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 16;
        a->v = (int*)realloc(a->v, sizeof(int) * (size_t)a->cap);
    }
    a->v[a->n++] = x;
}

/* ------------------------------------------------- alphabet + equality */

/* edlib.cpp:1417-1462 transformSequences: dense ids in first-appearance
 * order, query scanned first.  Returns the alphabet size. */
static int remap_alphabet(const char* q, int qn, const char* t, int tn,
                          unsigned char* qo, unsigned char* to,
                          unsigned char letters[256]) {
    int id_of[256];
    int sigma = 0;
    for (int i = 0; i < 256; i++) id_of[i] = -1;
    for (int pass = 0; pass < 2; pass++) {
        const char* s = pass ? t : q;
        unsigned char* o = pass ? to : qo;
        int n = pass ? tn : qn;
        for (int i = 0; i < n; i++) {
            unsigned char ch = (unsigned char)s[i];
            if (id_of[ch] < 0) { id_of[ch] = sigma; letters[sigma++] = ch; }
            o[i] = (unsigned char)id_of[ch];
        }
    }
    return sigma;
}

/* edlib.cpp:63-94 EqualityDefinition: identity plus symmetric extra pairs,
 * a pair counting only when both of its characters occur in the alphabet.
 * eq is a sigma*sigma byte matrix. */
static unsigned char* build_equality(const unsigned char letters[256], int sigma,
                                     const OracleEqualityPair* pairs, int npairs) {
    unsigned char* eq = (unsigned char*)calloc((size_t)sigma * sigma + 1, 1);
    for (int i = 0; i < sigma; i++) eq[i * sigma + i] = 1;
    if (pairs) {
        for (int p = 0; p < npairs; p++) {
            int a = -1, b = -1;
            for (int i = 0; i < sigma; i++) {
                if (a < 0 && letters[i] == (unsigned char)pairs[p].first) a = i;
                if (b < 0 && letters[i] == (unsigned char)pairs[p].second) b = i;
            }
            if (a >= 0 && b >= 0) eq[a * sigma + b] = eq[b * sigma + a] = 1;
        }
    }
    return eq;
}

/* edlib.cpp:358-384 buildPeq: (sigma+1) rows of nblk words; bit i of
 * row s, word b says query[64b+i] ~ s; rows past the query end count as
 * matches; the extra last row is all ones. */
static word_t* build_profile(int sigma, const unsigned char* q, int m,
                             const unsigned char* eq) {
    int nblk = ceil_div(m, WBITS);
    word_t* peq = (word_t*)malloc(sizeof(word_t) * (size_t)(sigma + 1) * nblk);
    for (int s = 0; s < sigma; s++) {
        for (int b = 0; b < nblk; b++) {
            word_t w = 0;
            for (int i = 0; i < WBITS; i++) {
                int r = b * WBITS + i;
                if (r >= m || eq[q[r] * sigma + s]) w |= (word_t)1 << i;
            }
            peq[(size_t)s * nblk + b] = w;
        }
    }
    for (int b = 0; b < nblk; b++) peq[(size_t)sigma * nblk + b] = ~(word_t)0;
    return peq;
}

/* ------------------------------------------------------ the block step */

/* edlib.cpp:412-447 calculateBlock (Myers' Advance_Block): one 64-row word,
 * vertical deltas (P,M) updated in place, horizontal delta in -> out. */
static inline int advance_block(word_t* P, word_t* M, word_t eq, int hin) {
    const word_t pv = *P, mv = *M;
    const word_t in_neg = (word_t)(hin < 0);
    const word_t in_pos = (word_t)(hin > 0);
    const word_t xv = eq | mv;
    eq |= in_neg;
    const word_t xh = (((eq & pv) + pv) ^ pv) | eq;
    word_t ph = mv | ~(xh | pv);
    word_t mh = pv & xh;
    const int hout = (int)(ph >> (WBITS - 1)) - (int)(mh >> (WBITS - 1));
    ph = (ph << 1) | in_pos;
    mh = (mh << 1) | in_neg;
    *P = mh | ~(xv | ph);
    *M = ph & xv;
    return hout;
}

/* edlib.cpp:470-482 getBlockCellValues: cells[0] is the bottom cell. */
static void decode_cells(word_t P, word_t M, int bottom, int cells[WBITS]) {
    int s = bottom;
    for (int i = 0; i < WBITS; i++) {
        cells[i] = s;
        const int bit = WBITS - 1 - i;
        s -= (int)((P >> bit) & 1);
        s += (int)((M >> bit) & 1);
    }
}

/* edlib.cpp:523-529 allBlockCellsLarger */
static int all_cells_above(word_t P, word_t M, int bottom, int k) {
    int cells[WBITS];
    decode_cells(P, M, bottom, cells);
    for (int i = 0; i < WBITS; i++) if (cells[i] <= k) return 0;
    return 1;
}

/* ------------------------------------------------- semi-global column scan */

/* edlib.cpp:550-704 myersCalcEditDistanceSemiGlobal (HW and SHW).
 * Blocks lo..hi form the Ukkonen band. Positions are malloc'd (NULL if none). */
static void scan_semiglobal(const word_t* peq, int nblk, int m,
                            const unsigned char* t, int tn, int k, int mode,
                            int* best_out, int** pos_out, int* npos_out) {
    const int W = nblk * WBITS - m;
    word_t* P = (word_t*)malloc(sizeof(word_t) * (size_t)nblk);
    word_t* M = (word_t*)malloc(sizeof(word_t) * (size_t)nblk);
    int* S = (int*)malloc(sizeof(int) * (size_t)nblk);
    ivec pos = {0, 0, 0};
    int best = -1;
    int lo = 0;
    int hi = imin(ceil_div(k + 1, WBITS), nblk) - 1;            /* :562 */
    int band_gone = 0;

    if (mode == ORACLE_MODE_HW) k = imin(m, k);                 /* :566-568 */
    for (int b = 0; b <= hi; b++) { P[b] = ~(word_t)0; M[b] = 0; S[b] = (b + 1) * WBITS; }  /* :575-579 */
    const int top_h = (mode == ORACLE_MODE_HW) ? 0 : 1;         /* :584 */

    for (int c = 0; c < tn; c++) {
        const word_t* eq = peq + (size_t)t[c] * nblk;           /* :587 */
        int h = top_h;
        for (int b = lo; b <= hi; b++) {                        /* :593-597 */
            h = advance_block(&P[b], &M[b], eq[b], h);
            S[b] += h;
        }
        /* :602-613 take one more block below, or drop blocks that left the band */
        if (hi < nblk - 1 && S[hi] - h <= k && ((eq[hi + 1] & 1) || h < 0)) {
            hi++;
            P[hi] = ~(word_t)0; M[hi] = 0;
            const int nh = advance_block(&P[hi], &M[hi], eq[hi], h);
            S[hi] = S[hi - 1] - h + WBITS + nh;
        } else {
            while (hi >= lo && S[hi] >= k + WBITS) hi--;
        }
        if (c % REDUCE_PERIOD == 0) {                           /* :619-623 */
            while (hi >= 0 && hi >= lo && all_cells_above(P[hi], M[hi], S[hi], k)) hi--;
        }
        if (mode == ORACLE_MODE_HW && hi == -1) hi = 0;         /* :628-630 */
        if (mode != ORACLE_MODE_HW) {                           /* :633-642 */
            while (lo <= hi && S[lo] >= k + WBITS) lo++;
            if (c % REDUCE_PERIOD == 0) {
                while (lo <= hi && all_cells_above(P[lo], M[lo], S[lo], k)) lo++;
            }
        }
        if (hi < lo) { band_gone = 1; break; }                  /* :645-654 */

        if (hi == nblk - 1) {                                   /* :658-673 */
            const int v = S[hi];
            if (v <= k && (best == -1 || v <= best)) {
                if (v != best) { pos.n = 0; best = v; k = best; }
                ivec_push(&pos, c - W);
            }
        }
    }

    if (!band_gone && hi == nblk - 1) {                         /* :681-693 */
        int cells[WBITS];
        decode_cells(P[hi], M[hi], S[hi], cells);
        for (int i = 0; i < W; i++) {
            const int v = cells[i + 1];
            if (v <= k && (best == -1 || v <= best)) {
                if (v != best) { pos.n = 0; k = best = v; }
                ivec_push(&pos, tn - W + i);
            }
        }
    }

    *best_out = best;
    *pos_out = NULL; *npos_out = 0;
    if (best != -1) {                                           /* :695-700 */
        *pos_out = (int*)malloc(sizeof(int) * (size_t)imax(pos.n, 1));
        memcpy(*pos_out, pos.v, sizeof(int) * (size_t)pos.n);
        *npos_out = pos.n;
    }
    free(pos.v); free(P); free(M); free(S);
}

/* ------------------------------------------------------ global column scan */

/* edlib.cpp:22-47 AlignmentData: band columns kept for the traceback.
 * Index (c * nblk + b); only blocks first[c]..last[c] of a column are valid. */
typedef struct {
    word_t* P; word_t* M; int* S; int* first; int* last;
    int nblk; int ncols;
} colstore;

static colstore* colstore_new(int nblk, int ncols) {
    colstore* st = (colstore*)malloc(sizeof(colstore));
    st->nblk = nblk; st->ncols = ncols;
    st->P = (word_t*)malloc(sizeof(word_t) * (size_t)nblk * ncols);
    st->M = (word_t*)malloc(sizeof(word_t) * (size_t)nblk * ncols);
    st->S = (int*)malloc(sizeof(int) * (size_t)nblk * ncols);
    st->first = (int*)malloc(sizeof(int) * (size_t)ncols);
    st->last = (int*)malloc(sizeof(int) * (size_t)ncols);
    return st;
}
static void colstore_free(colstore* st) {
    if (!st) return;
    free(st->P); free(st->M); free(st->S); free(st->first); free(st->last); free(st);
}

/* edlib.cpp:835-870: is any cell of block b (column c) still inside the
 * diagonal band?  lower=1 tests against the lower band edge (used when
 * trimming hi), lower=0 against the upper edge (trimming lo). */
static int block_in_diag_band(word_t P, word_t M, int bottom, int b, int nblk, int W,
                              int k, int m, int tn, int c, int lower) {
    int cells[WBITS];
    decode_cells(P, M, bottom, cells);
    const int ncell = (b == nblk - 1) ? WBITS - W : WBITS;
    int r = b * WBITS + ncell - 1;
    for (int i = WBITS - ncell; i < WBITS; i++, r--) {
        if (cells[i] > k) continue;
        if (lower) { if (r <= k - cells[i] - tn + c + m + 1) return 1; }
        else       { if (r >= cells[i] - k - tn + c + m) return 1; }
    }
    return 0;
}

/* edlib.cpp:730-928 myersCalcEditDistanceNW.
 * keep == 1: store every column (findAlignment); stop_col >= 0: run up to and
 * including that column, store only it (ncols==1) and return. */
static int scan_global(const word_t* peq, int nblk, int m,
                       const unsigned char* t, int tn, int k,
                       int* best_out, int* pos_out,
                       int keep, int stop_col, colstore** store_out) {
    const int W = nblk * WBITS - m;
    if (store_out) *store_out = NULL;
    if (stop_col > -1 && keep) return ORACLE_ERROR;             /* :736-739 */
    if (k < abs(tn - m)) { *best_out = *pos_out = -1; return ORACLE_OK; }   /* :744-747 */
    k = imin(k, imax(m, tn));                                   /* :749 */

    int lo = 0;
    int hi = imin(nblk, ceil_div(imin(k, (k + m - tn) / 2) + 1, WBITS)) - 1;   /* :755 */
    word_t* P = (word_t*)malloc(sizeof(word_t) * (size_t)nblk);
    word_t* M = (word_t*)malloc(sizeof(word_t) * (size_t)nblk);
    int* S = (int*)malloc(sizeof(int) * (size_t)nblk);
    for (int b = 0; b <= hi; b++) { P[b] = ~(word_t)0; M[b] = 0; S[b] = (b + 1) * WBITS; }

    colstore* st = NULL;
    if (keep) st = colstore_new(nblk, tn);                      /* :766-771 */
    else if (stop_col > -1) st = colstore_new(nblk, 1);

    int status_best = -1, status_pos = -1, done = 0;
    for (int c = 0; c < tn && !done; c++) {
        const word_t* eq = peq + (size_t)t[c] * nblk;
        int h = 1;                                              /* :779 */
        for (int b = lo; b <= hi; b++) {
            h = advance_block(&P[b], &M[b], eq[b], h);
            S[b] += h;
        }
        /* :792-794 tighten k from what is still reachable */
        k = imin(k, S[hi] + imax(tn - c - 1, m - ((1 + hi) * WBITS - 1) - 1)
                    + (hi == nblk - 1 ? W : 0));

        /* :799-809 grow by one block if it is not yet under the band */
        if (hi + 1 < nblk &&
            !((hi + 1) * WBITS - 1 > k - S[hi] + 2 * WBITS - 2 - tn + c + m)) {
            hi++;
            P[hi] = ~(word_t)0; M[hi] = 0;
            const int nh = advance_block(&P[hi], &M[hi], eq[hi], h);
            S[hi] = S[hi - 1] - h + WBITS + nh;
            h = nh;
        }
        /* :814-820 drop bottom blocks outside the band */
        while (hi >= lo &&
               (S[hi] >= k + WBITS ||
                ((hi + 1) * WBITS - 1 > k - S[hi] + 2 * WBITS - 2 - tn + c + m + 1))) hi--;
        /* :825-830 drop top blocks outside the band */
        while (lo <= hi &&
               (S[lo] >= k + WBITS ||
                ((lo + 1) * WBITS - 1 < S[lo] - k - tn + m + c))) lo++;

        if (c % REDUCE_PERIOD == 0) {                           /* :835-870 */
            while (hi >= lo &&
                   !block_in_diag_band(P[hi], M[hi], S[hi], hi, nblk, W, k, m, tn, c, 1)) hi--;
            while (lo <= hi &&
                   !block_in_diag_band(P[lo], M[lo], S[lo], lo, nblk, W, k, m, tn, c, 0)) lo++;
        }
        if (hi < lo) { status_best = status_pos = -1; done = 2; break; }    /* :874-878 */

        if (keep) {                                             /* :883-893 */
            for (int b = lo; b <= hi; b++) {
                st->P[(size_t)c * nblk + b] = P[b];
                st->M[(size_t)c * nblk + b] = M[b];
                st->S[(size_t)c * nblk + b] = S[b];
            }
            st->first[c] = lo; st->last[c] = hi;
        }
        if (c == stop_col) {                                    /* :896-908 */
            for (int b = lo; b <= hi; b++) { st->P[b] = P[b]; st->M[b] = M[b]; st->S[b] = S[b]; }
            st->first[0] = lo; st->last[0] = hi;
            status_best = -1; status_pos = stop_col; done = 1;
        }
    }

    if (!done) {                                                /* :914-925 */
        if (hi == nblk - 1) {
            int cells[WBITS];
            decode_cells(P[hi], M[hi], S[hi], cells);
            if (cells[W] <= k) { status_best = cells[W]; status_pos = tn - 1; }
        }
    }
    *best_out = status_best; *pos_out = status_pos;
    if (store_out) *store_out = st; else colstore_free(st);
    free(P); free(M); free(S);
    return ORACLE_OK;
}

/* --------------------------------------------------------------- traceback */

/* Value of cell (row r, column c) from the stored band column, or -1 when its
 * block was not kept for that column.  Same arithmetic as the bit peeling at
 * edlib.cpp:986-993. */
static int stored_cell(const colstore* st, int c, int r) {
    const int b = r / WBITS;
    if (b < st->first[c] || b > st->last[c]) return -1;
    const word_t P = st->P[(size_t)c * st->nblk + b], M = st->M[(size_t)c * st->nblk + b];
    int s = st->S[(size_t)c * st->nblk + b];
    for (int bit = WBITS - 1; bit > r % WBITS; bit--) {
        s -= (int)((P >> bit) & 1);
        s += (int)((M >> bit) & 1);
    }
    return s;
}
/* Vertical delta of cell (r,c): +1, 0 or -1 (cell minus the cell above). */
static int stored_vdelta(const colstore* st, int c, int r) {
    const int b = r / WBITS, bit = r % WBITS;
    const word_t P = st->P[(size_t)c * st->nblk + b], M = st->M[(size_t)c * st->nblk + b];
    return (int)((P >> bit) & 1) - (int)((M >> bit) & 1);
}

/* edlib.cpp:942-1141 obtainAlignmentTraceback: walk from (m-1, tn-1) to the
 * origin; candidate order up (INSERT) > left (DELETE) > diagonal; ops are
 * produced back-to-front and reversed at the end (:1138-1139). */
static void traceback(int m, int tn, int best, const colstore* st,
                      unsigned char** aln_out, int* len_out) {
    unsigned char* ops = (unsigned char*)malloc((size_t)(m + tn) + 1);
    int n = 0;
    int r = m - 1, c = tn - 1, cur = best;
    for (;;) {
        int u, l, ul;
        u = cur - stored_vdelta(st, c, r);                      /* :1007-1013 */
        if (c == 0) { l = r + 1; ul = r; }                      /* :976-980 */
        else {
            l = stored_cell(st, c - 1, r);                      /* :986-994 */
            if (l != -1) ul = l - stored_vdelta(st, c - 1, r);  /* :996-1000 */
            else {                                              /* :1001-1005 */
                const int b = r / WBITS;
                ul = (b - 1 >= st->first[c - 1] && b - 1 <= st->last[c - 1])
                         ? st->S[(size_t)(c - 1) * st->nblk + b - 1] : -1;
            }
        }
        if (u + 1 == cur) {                                     /* :1020-1052 up */
            cur = u;
            ops[n++] = 1;
            if (r == 0) { for (int i = 0; i < c + 1; i++) ops[n++] = 2; break; }
            r--;
        } else if (l != -1 && l + 1 == cur) {                   /* :1054-1083 left */
            cur = l;
            ops[n++] = 2;
            c--;
            if (c == -1) { for (int i = 0; i < r + 1; i++) ops[n++] = 1; break; }
        } else if (ul != -1) {                                  /* :1085-1131 diagonal */
            ops[n++] = (unsigned char)(ul == cur ? 0 : 3);
            cur = ul;
            c--;
            if (c == -1) { for (int i = 0; i < r; i++) ops[n++] = 1; break; }
            if (r == 0) { for (int i = 0; i < c + 1; i++) ops[n++] = 2; break; }
            r--;
        } else {
            break;                                              /* :1131-1134 */
        }
    }
    for (int i = 0, j = n - 1; i < j; i++, j--) { unsigned char x = ops[i]; ops[i] = ops[j]; ops[j] = x; }
    *aln_out = (unsigned char*)realloc(ops, (size_t)imax(n, 1));
    *len_out = n;
}

/* edlib.cpp:1161-1213 obtainAlignment (traceback branch only). */
static int find_path(const unsigned char* q, int m, const unsigned char* t, int tn,
                     const unsigned char* eq, int sigma, int best,
                     unsigned char** aln_out, int* len_out) {
    if (m == 0 || tn == 0) {                                    /* :1168-1175 */
        *len_out = m + tn;
        *aln_out = (unsigned char*)malloc((size_t)imax(*len_out, 1));
        memset(*aln_out, m == 0 ? 2 : 1, (size_t)*len_out);
        return ORACLE_OK;
    }
    const int nblk = ceil_div(m, WBITS);
    const long long bytes = (2ll * sizeof(word_t) + sizeof(int)) * nblk * tn
                            + 2ll * sizeof(int) * tn;           /* :1188-1189 */
    if (bytes >= 1024 * 1024) return ORACLE_UNSUPPORTED;        /* Hirschberg regime */
    word_t* peq = build_profile(sigma, q, m, eq);
    colstore* st = NULL;
    int sc, ps;
    scan_global(peq, nblk, m, t, tn, best, &sc, &ps, 1, -1, &st);   /* :1194-1198 */
    traceback(m, tn, best, st, aln_out, len_out);               /* :1202 */
    colstore_free(st);
    free(peq);
    return ORACLE_OK;
}

/* ----------------------------------------------------------- top level */

/* edlib.cpp:146-301 edlibAlign. */
OracleAlignResult oracle_align(const char* query, int m, const char* target, int tn,
                               int k_cfg, int mode, int task,
                               const OracleEqualityPair* pairs, int npairs) {
    OracleAlignResult res;
    res.status = ORACLE_OK; res.editDistance = -1;
    res.endLocations = res.startLocations = NULL; res.numLocations = 0;
    res.alignment = NULL; res.alignmentLength = 0; res.alphabetLength = 0;

    unsigned char* q = (unsigned char*)malloc((size_t)imax(m, 1));
    unsigned char* t = (unsigned char*)malloc((size_t)imax(tn, 1));
    unsigned char letters[256];
    const int sigma = remap_alphabet(query, m, target, tn, q, t, letters);
    res.alphabetLength = sigma;                                 /* :162 */

    if (m == 0 || tn == 0) {                                    /* :166-184 */
        if (mode == ORACLE_MODE_NW) {
            res.editDistance = imax(m, tn);
            res.endLocations = (int*)malloc(sizeof(int));
            res.endLocations[0] = tn - 1; res.numLocations = 1;
        } else if (mode == ORACLE_MODE_SHW || mode == ORACLE_MODE_HW) {
            res.editDistance = m;
            res.endLocations = (int*)malloc(sizeof(int));
            res.endLocations[0] = -1; res.numLocations = 1;
        } else res.status = ORACLE_ERROR;
        free(q); free(t);
        return res;
    }

    const int nblk = ceil_div(m, WBITS);                        /* :187-190 */
    unsigned char* eq = build_equality(letters, sigma, pairs, npairs);
    word_t* peq = build_profile(sigma, q, m, eq);

    int k = k_cfg, auto_k = 0;                                  /* :197-202 */
    if (k < 0) { auto_k = 1; k = WBITS; }
    int nw_pos;
    do {                                                        /* :204-217 */
        if (mode == ORACLE_MODE_HW || mode == ORACLE_MODE_SHW) {
            free(res.endLocations);
            scan_semiglobal(peq, nblk, m, t, tn, k, mode,
                            &res.editDistance, &res.endLocations, &res.numLocations);
        } else {
            scan_global(peq, nblk, m, t, tn, k, &res.editDistance, &nw_pos, 0, -1, NULL);
        }
        k *= 2;
    } while (auto_k && res.editDistance == -1);

    if (res.editDistance >= 0) {
        if (mode == ORACLE_MODE_NW) {                           /* :221-225 */
            res.endLocations = (int*)malloc(sizeof(int));
            res.endLocations[0] = tn - 1; res.numLocations = 1;
        }
        if (task == ORACLE_TASK_LOC || task == ORACLE_TASK_PATH) {   /* :228-272 */
            res.startLocations = (int*)malloc(sizeof(int) * (size_t)imax(res.numLocations, 1));
            if (mode == ORACLE_MODE_HW) {
                unsigned char* rt = (unsigned char*)malloc((size_t)tn);
                unsigned char* rq = (unsigned char*)malloc((size_t)m);
                for (int i = 0; i < tn; i++) rt[i] = t[tn - 1 - i];
                for (int i = 0; i < m; i++) rq[i] = q[m - 1 - i];
                word_t* rpeq = build_profile(sigma, rq, m, eq);
                for (int i = 0; i < res.numLocations; i++) {
                    const int end = res.endLocations[i];
                    if (end == -1) { res.startLocations[i] = 0; continue; }   /* :237-249 */
                    int sb, sn; int* sp;
                    scan_semiglobal(rpeq, nblk, m, rt + tn - end - 1, end + 1,
                                    res.editDistance, ORACLE_MODE_SHW, &sb, &sp, &sn);
                    res.startLocations[i] = end - sp[sn - 1];   /* :260 */
                    free(sp);
                }
                free(rt); free(rq); free(rpeq);
            } else {
                for (int i = 0; i < res.numLocations; i++) res.startLocations[i] = 0;
            }
        }
        if (task == ORACLE_TASK_PATH) {                         /* :276-289 */
            const int s = res.startLocations[0], e = res.endLocations[0];
            const int st = find_path(q, m, t + s, e - s + 1, eq, sigma, res.editDistance,
                                     &res.alignment, &res.alignmentLength);
            if (st != ORACLE_OK) res.status = st;
        }
    }
    free(peq); free(eq); free(q); free(t);
    return res;
}

void oracle_free_result(OracleAlignResult* r) {                 /* edlib.cpp:1481-1485 */
    free(r->endLocations); free(r->startLocations); free(r->alignment);
    r->endLocations = r->startLocations = NULL; r->alignment = NULL;
}

/* edlib.cpp:303-350 edlibAlignmentToCigar: run-length encode the op codes. */
char* oracle_cigar(const unsigned char* aln, int len, int format) {
    if (format != ORACLE_CIGAR_EXTENDED && format != ORACLE_CIGAR_STANDARD) return NULL;
    const char* sym = (format == ORACLE_CIGAR_STANDARD) ? "MIDM" : "=IDX";
    for (int i = 0; i < len; i++) if (aln[i] > 3) return NULL;
    /* worst case: every op its own run: "1X" per op */
    char* out = (char*)malloc((size_t)len * 2 + 1 + 11);
    size_t w = 0;
    int i = 0;
    while (i < len) {
        const char ch = sym[aln[i]];
        int run = 0;
        while (i < len && sym[aln[i]] == ch) { run++; i++; }
        char digits[12]; int nd = 0;
        while (run) { digits[nd++] = (char)('0' + run % 10); run /= 10; }
        while (nd) out[w++] = digits[--nd];
        out[w++] = ch;
    }
    out[w++] = '\0';
    return (char*)realloc(out, w);
}

/* Independent O(m*T) dynamic programme (purpose of test/SimpleEditDistance.h). */
int oracle_simple_dp(const unsigned char* q, int m, const unsigned char* t, int tn,
                     int mode, int* score, int** positions, int* npos) {
    *positions = NULL; *npos = 0; *score = -1;
    if (m == 0 || tn == 0) return ORACLE_ERROR;
    int* col = (int*)malloc(sizeof(int) * (size_t)(m + 1));
    ivec pos = {0, 0, 0};
    int best = -1;
    for (int i = 0; i <= m; i++) col[i] = i;                    /* column before the target */
    for (int c = 0; c < tn; c++) {
        int diag = col[0];
        col[0] = (mode == ORACLE_MODE_HW) ? 0 : c + 1;
        for (int i = 1; i <= m; i++) {
            const int sub = diag + (q[i - 1] == t[c] ? 0 : 1);
            diag = col[i];
            col[i] = imin(sub, imin(col[i] + 1, col[i - 1] + 1));
        }
        if (mode != ORACLE_MODE_NW || c == tn - 1) {
            const int v = col[m];
            if (best == -1 || v <= best) {
                if (v < best) pos.n = 0;
                best = v;
                ivec_push(&pos, c);
            }
        }
    }
    *score = best; *positions = pos.v; *npos = pos.n;
    free(col);
    return ORACLE_OK;
}
