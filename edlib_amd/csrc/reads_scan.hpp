// reads_scan.hpp -- device templates of the banded and the full-height HW scans of the reads-per-lane family
// (scan_reads_banded_kernel, scan_reads_full_kernel and what they are made of).  Included by reads_kernels.hip (groups of up
// to 8 words, the piece filter) and reads_kernels_long.hip (groups of 12 / 16 / 24 / 32 words): two translation units, so that
// the long groups -- most of the compile time, code size grows with the square of the word count -- build in parallel.
#pragma once
#include "lds_check.hpp"
#include "reads_kernels.hpp"
#include "reads_column_asm.hpp"

namespace edlib_amd {

typedef uint32_t u32;

// ------------------------------------------------- the banded HW scan (k-doubling)

// Ukkonen band + k-doubling of the reference (edlib.cpp:197-217, 562, 602-630) re-expressed per WAVE:
//   * the wave computes only the first `nw` words of the column (nw is wave-uniform, 1..NWD); rows
//     below are known to exceed every lane's threshold k = best-so-far (<= min(kinit, kcap));
//   * nw is re-evaluated from the computed score S of the band's bottom row (popcounts of Pv/Mv: with HW's zero
//     row -1, D[r] = sum of the vertical deltas above r) every c = 4 columns at one word, 8 at two, 16 above
//     (band_quad has the rules and why they are sound):
//        S <= k + c - 1 (any lane) -> take one more word, entering as "+1 per row" like the reference's
//                               new block (edlib.cpp:605-608: P = ~0, M = 0)
//        all lanes: every cell of the last word and the rows the next interval needs above it exceed k -> drop
//                               the word (edlib.cpp:610-612; bounds from the word's boundary scores)
//   * the bottom query row (bit (m-1)%32 of the last word) is only in the band while nw == NWD, so
//     its score is only followed then; e = score - best - 1 is tracked instead of score, its sign bit is
//     OR-ed into `flag`, and once per 4 columns a WAVE-UNIFORM test (ballot) enters the rare path,
//     which updates best/count with selects and stores the position under an exec mask.
// Op selection follows tools/valu_ubench.hip: x+x instead of v_lshlrev (half rate on gfx950),
// v_lshrrev + v_and instead of v_bfe, no SGPR operands in the hot VALU ops.
// Two active words are one 64-bit value: v_lshl_add_u64 and v_lshlrev_b64 take 4 cycles for 64 bits
// (tools/valu_ubench.hip) where the 32-bit carry chain and v_alignbit pairs take 8 and 6.
// v_bitop3_b32 truth tables: bit i of the immediate is f(a,b,c) with i = a*4 + b*2 + c
#define BITOP3_XOR_OR(a, b, c)   __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xde)   /* (a ^ c) | b   */
#define BITOP3_OR_NOR(a, b, c)   __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xf1)   /* a | ~(b | c)  */

template <int NWD>
__device__ __forceinline__ void column_step_eq2(const u32 eq0, const u32 eq1, u32 (&Pv)[NWD], u32 (&Mv)[NWD])
{
    typedef unsigned long long u64;
    // booleans stay 32-bit, one v_bitop3_b32 per 3-input function; only the add and the two shifts
    // see the 64-bit register pair
    const u32 t0 = eq0 & Pv[0], t1 = eq1 & Pv[1];
    const u64 t = ((u64)t1 << 32) | t0, pv = ((u64)Pv[1] << 32) | Pv[0];
    u64 s;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(s) : "v"(t), "v"(pv));
    const u32 Xh0 = BITOP3_XOR_OR((u32)s, eq0, Pv[0]), Xh1 = BITOP3_XOR_OR((u32)(s >> 32), eq1, Pv[1]);
    const u32 Ph0 = BITOP3_OR_NOR(Mv[0], Xh0, Pv[0]), Ph1 = BITOP3_OR_NOR(Mv[1], Xh1, Pv[1]);
    const u32 Mh0 = Pv[0] & Xh0, Mh1 = Pv[1] & Xh1;
    u64 ph, mh;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(ph) : "v"(((u64)Ph1 << 32) | Ph0));
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(mh) : "v"(((u64)Mh1 << 32) | Mh0));
    const u32 Xv0 = eq0 | Mv[0], Xv1 = eq1 | Mv[1];
    Pv[0] = BITOP3_OR_NOR((u32)mh, Xv0, (u32)ph);  Pv[1] = BITOP3_OR_NOR((u32)(mh >> 32), Xv1, (u32)(ph >> 32));
    Mv[0] = (u32)ph & Xv0;                          Mv[1] = (u32)(ph >> 32) & Xv1;
}

template <int NWD>
__device__ __forceinline__ void column_step_eq1(const u32 eq0, u32 (&Pv)[NWD], u32 (&Mv)[NWD])
{
    const u32 s = (eq0 & Pv[0]) + Pv[0];                       // carry out of the band's top word is not needed
    const u32 Xh = BITOP3_XOR_OR(s, eq0, Pv[0]);
    const u32 Ph = BITOP3_OR_NOR(Mv[0], Xh, Pv[0]);
    const u32 Mh = Pv[0] & Xh;
    u32 ph, mh;
    asm("v_add_u32 %0, %1, %1" : "=v"(ph) : "v"(Ph));
    asm("v_add_u32 %0, %1, %1" : "=v"(mh) : "v"(Mh));
    const u32 Xv = eq0 | Mv[0];
    Pv[0] = BITOP3_OR_NOR(mh, Xv, ph);
    Mv[0] = ph & Xv;
}

// All Peq rows of the wave's 64 queries live in LDS as [word][symbol][lane] (1 KB per word).  The kernel runs ONE
// wave per workgroup, so its slice starts at LDS address 0 and the row of a column's (wave-uniform) symbol is
// M0 = symbol << 8, fetched with ds_read_addtid_b32 (address = M0 + 1024 * word + 4 * lane: no address VGPR, no
// VALU).  The target is read in the EXPANDED form the row fetch wants (pack_target_rows_kernel): 16 bits per
// column holding that row offset, 16 columns per s_load_dwordx8, so a column costs ONE scalar instruction --
// s_pack_ll_b32_b16 m0, pair, 0 (even column) or s_lshr_b32 m0, pair, 16 (odd column) -- where round 1 spent three
// (shift, mask, or-in the slice base) plus the extraction of the quad's byte (tools/narrow2_ubench.hip: every
// non-VALU instruction of the one-word column costs about a cycle of the 20 its ten VALU ops take).
// s_lshr_b32 writes SCC: declared clobbered (hipcc keeps carries and loop conditions there; without it a
// s_add_u32 / s_addc_u32 pair around the block computed a wild address).  M0 needs one wait state before an
// add-TID LDS instruction: s_nop here, the column's first VALU op in the hand-scheduled one-word quad.
#define EDLIB_AMD_M0_EVEN(P) "s_pack_ll_b32_b16 m0, " P ", 0\n\t"
#define EDLIB_AMD_M0_ODD(P)  "s_lshr_b32 m0, " P ", 16\n\t"
#define EDLIB_AMD_RD(N, OFF) "ds_read_addtid_b32 " N " offset:" #OFF "\n\t"
#define EDLIB_AMD_RDW(N, W) "ds_read_addtid_b32 " N " offset:%[w" #W "]\n\t"      /* word W of the row: offset W * S * 256 */

// rows of ONE column (J = 0..3 of the quad held in the SGPR pair lo / hi), NA words; S symbols per word (row stride 256 S).
// Up to eight words per asm statement (operand count); heights 12 and 16 take two, each writing M0 itself (nothing
// tells the compiler that M0 is live between two statements).
template <int NA, int W0, int CNT, int J, int S>
__device__ __forceinline__ void lds_rows_chunk(u32 (&n)[NA], const u32 pr)
{
    static_assert(CNT >= 1 && CNT <= 8 && W0 + CNT <= NA, "chunk of band words");
#define EDLIB_AMD_WOFF [w0] "n"((W0 + 0) * S * 256), [w1] "n"((W0 + 1) * S * 256), [w2] "n"((W0 + 2) * S * 256), [w3] "n"((W0 + 3) * S * 256), \
                       [w4] "n"((W0 + 4) * S * 256), [w5] "n"((W0 + 5) * S * 256), [w6] "n"((W0 + 6) * S * 256), [w7] "n"((W0 + 7) * S * 256)
#define EDLIB_AMD_ROWS_ASM(READS, OUTS)                                                                              \
    { if constexpr (J & 1) asm volatile(EDLIB_AMD_M0_ODD("%[pr]") "s_nop 0\n\t" READS : OUTS : [pr] "s"(pr), EDLIB_AMD_WOFF : "memory", "scc");   \
      else asm volatile(EDLIB_AMD_M0_EVEN("%[pr]") "s_nop 0\n\t" READS : OUTS : [pr] "s"(pr), EDLIB_AMD_WOFF : "memory", "scc"); }
#define O1 "=v"(n[W0])
#define O2 O1, "=v"(n[W0 + (CNT > 1 ? 1 : 0)])
#define O3 O2, "=v"(n[W0 + (CNT > 2 ? 2 : 0)])
#define O4 O3, "=v"(n[W0 + (CNT > 3 ? 3 : 0)])
#define O5 O4, "=v"(n[W0 + (CNT > 4 ? 4 : 0)])
#define O6 O5, "=v"(n[W0 + (CNT > 5 ? 5 : 0)])
#define O7 O6, "=v"(n[W0 + (CNT > 6 ? 6 : 0)])
#define O8 O7, "=v"(n[W0 + (CNT > 7 ? 7 : 0)])
#define R1 EDLIB_AMD_RDW("%0", 0)
#define R2 R1 EDLIB_AMD_RDW("%1", 1)
#define R3 R2 EDLIB_AMD_RDW("%2", 2)
#define R4 R3 EDLIB_AMD_RDW("%3", 3)
#define R5 R4 EDLIB_AMD_RDW("%4", 4)
#define R6 R5 EDLIB_AMD_RDW("%5", 5)
#define R7 R6 EDLIB_AMD_RDW("%6", 6)
#define R8 R7 EDLIB_AMD_RDW("%7", 7)
    if constexpr (CNT == 1) EDLIB_AMD_ROWS_ASM(R1, O1)
    if constexpr (CNT == 2) EDLIB_AMD_ROWS_ASM(R2, O2)
    if constexpr (CNT == 3) EDLIB_AMD_ROWS_ASM(R3, O3)
    if constexpr (CNT == 4) EDLIB_AMD_ROWS_ASM(R4, O4)
    if constexpr (CNT == 5) EDLIB_AMD_ROWS_ASM(R5, O5)
    if constexpr (CNT == 6) EDLIB_AMD_ROWS_ASM(R6, O6)
    if constexpr (CNT == 7) EDLIB_AMD_ROWS_ASM(R7, O7)
    if constexpr (CNT == 8) EDLIB_AMD_ROWS_ASM(R8, O8)
}
template <int NA, int J, int S>
__device__ __forceinline__ void lds_rows_request(u32 (&n)[NA], const u32 lo, const u32 hi)
{
    static_assert(NA >= 1 && NA <= 32, "band height");
    const u32 pr = (J < 2) ? lo : hi;
    lds_rows_chunk<NA, 0, (NA < 8 ? NA : 8), J, S>(n, pr);
    if constexpr (NA > 8) lds_rows_chunk<NA, 8, (NA - 8 < 8 ? NA - 8 : 8), J, S>(n, pr);
    if constexpr (NA > 16) lds_rows_chunk<NA, 16, (NA - 16 < 8 ? NA - 16 : 8), J, S>(n, pr);
    if constexpr (NA > 24) lds_rows_chunk<NA, 24, NA - 24, J, S>(n, pr);
}
template <int NA, int W0, int CNT>
__device__ __forceinline__ void lds_rows_wait_chunk(u32 (&n)[NA])
{
#define N_(i) "+v"(n[W0 + (CNT > i ? i : 0)])
    if constexpr (CNT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0));
    if constexpr (CNT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0), N_(1));
    if constexpr (CNT == 3) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0), N_(1), N_(2));
    if constexpr (CNT == 4) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0), N_(1), N_(2), N_(3));
    if constexpr (CNT == 5) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0), N_(1), N_(2), N_(3), N_(4));
    if constexpr (CNT == 6) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0), N_(1), N_(2), N_(3), N_(4), N_(5));
    if constexpr (CNT == 7) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0), N_(1), N_(2), N_(3), N_(4), N_(5), N_(6));
    if constexpr (CNT == 8) asm volatile("s_waitcnt lgkmcnt(0)" : N_(0), N_(1), N_(2), N_(3), N_(4), N_(5), N_(6), N_(7));
#undef N_
}
template <int NA>
__device__ __forceinline__ void lds_rows_wait(u32 (&n)[NA])
{
    lds_rows_wait_chunk<NA, 0, (NA < 8 ? NA : 8)>(n);                                     // the waits after the first are free
    if constexpr (NA > 8) lds_rows_wait_chunk<NA, 8, (NA - 8 < 8 ? NA - 8 : 8)>(n);
    if constexpr (NA > 16) lds_rows_wait_chunk<NA, 16, (NA - 16 < 8 ? NA - 16 : 8)>(n);
    if constexpr (NA > 24) lds_rows_wait_chunk<NA, 24, NA - 24>(n);
}

// ---- the band of one word carries the rows of the NEXT quad across quads (requested while the current
// quad computes, one s_waitcnt per quad).  word-0 rows of the quad in (lo, hi):
__device__ __forceinline__ void quad_rows_request1(u32 (&n)[4], const u32 lo, const u32 hi)
{
    asm volatile(EDLIB_AMD_M0_EVEN("%[lo]") "s_nop 0\n\t" EDLIB_AMD_RD("%[n0]", 0)
                 EDLIB_AMD_M0_ODD("%[lo]") "s_nop 0\n\t" EDLIB_AMD_RD("%[n1]", 0)
                 EDLIB_AMD_M0_EVEN("%[hi]") "s_nop 0\n\t" EDLIB_AMD_RD("%[n2]", 0)
                 EDLIB_AMD_M0_ODD("%[hi]") "s_nop 0\n\t" EDLIB_AMD_RD("%[n3]", 0)
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3])
                 : [lo] "s"(lo), [hi] "s"(hi) : "memory", "scc");
}
__device__ __forceinline__ void quad_rows_wait1(u32 (&n)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3])); }

// One column of the one-word band on row register C, with the request of row N of the NEXT quad folded in: MSET
// writes M0, the v_and after it is the wait state an add-TID LDS instruction needs after an M0 write.  Ten VALU
// ops, all full rate (column_step_eq1 in the compiler's hands gives the same ten; written out so that the four
// requests sit where they cost nothing and the quad ends in ONE s_waitcnt).
#define EDLIB_AMD_NB_COL(C, MSET, N)                                    \
    MSET                                                                \
    "v_and_b32 %[t], " C ", %[pv]\n\t"                                  \
    EDLIB_AMD_RD(N, 0)                                                  \
    "v_add_u32 %[t], %[t], %[pv]\n\t"                                   \
    "v_bitop3_b32 %[x], %[t], " C ", %[pv] bitop3:0xde\n\t"             \
    "v_bitop3_b32 %[p], %[mv], %[x], %[pv] bitop3:0xf1\n\t"             \
    "v_and_b32 %[m], %[pv], %[x]\n\t"                                   \
    "v_add_u32 %[p], %[p], %[p]\n\t"                                    \
    "v_add_u32 %[m], %[m], %[m]\n\t"                                    \
    "v_or_b32 %[x], " C ", %[mv]\n\t"                                   \
    "v_bitop3_b32 %[pv], %[m], %[x], %[p] bitop3:0xf1\n\t"              \
    "v_and_b32 %[mv], %[p], %[x]\n\t"
__device__ __forceinline__ void quad_one_word(u32 (&c)[4], u32& Pv, u32& Mv, const u32 nlo, const u32 nhi)
{
    u32 n0, n1, n2, n3, t, x, p, m;
    asm volatile(
        EDLIB_AMD_NB_COL("%[c0]", EDLIB_AMD_M0_EVEN("%[lo]"), "%[n0]")
        EDLIB_AMD_NB_COL("%[c1]", EDLIB_AMD_M0_ODD("%[lo]"), "%[n1]")
        EDLIB_AMD_NB_COL("%[c2]", EDLIB_AMD_M0_EVEN("%[hi]"), "%[n2]")
        EDLIB_AMD_NB_COL("%[c3]", EDLIB_AMD_M0_ODD("%[hi]"), "%[n3]")
        "s_waitcnt lgkmcnt(0)"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [t] "=&v"(t), [x] "=&v"(x), [p] "=&v"(p), [m] "=&v"(m),
          [pv] "+v"(Pv), [mv] "+v"(Mv)
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [lo] "s"(nlo), [hi] "s"(nhi)
        : "memory", "scc");
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// CHAIN (scan_reads_full_kernel of a strip of a taller query): the strip's row -1 is the bottom row of the strip above, whose
// horizontal delta of this column arrives in the two low bits of `cin` (bit 0: +1, bit 1: -1; consumed), and the
// delta of the strip's own bottom row (bit 31 of its last word) is shifted into `cout` from the top: after 16 columns
// cout holds them in column order.  These are the hin / hout terms of calculateBlock (edlib.cpp:412-447) that the top
// strip does without (HW row -1: hin = 0).
template <int NA, int NWD, bool CHAIN = false>
__device__ __forceinline__ void column_step_hw(const u32 (&Eq)[NA], u32 (&Pv)[NWD], u32 (&Mv)[NWD],
                                               int& e, int& flag, const u32 sh, u32& cin, u32& cout)
{
    if constexpr (!CHAIN && NWD <= 8 && NA >= 2) {
        // one asm statement of VOP3 encodings behind an alignment fence (reads_column_asm.hpp: a compiled column mixes 4- and
        // 8-byte encodings, and a stream of full- and half-rate instructions only issues at the sum of their rates in one phase)
        u32 t_, s_, xh_, ph0_, ph1_, mh0_, mh1_, phs_, mhs_, xv_, Pn[NA], Mn[NA];
        unsigned long long cy_;
        if constexpr (NA == NWD) {
            int score = e, scoreN;
            RC_COLUMN_DISPATCH(NA, RC_WORD0_HW)
            e = scoreN;
            flag |= e;
        } else {
            RC_COLUMN_DISPATCH_NS(NA, RC_WORD0_HW)
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) { Pv[i] = Pn[i]; Mv[i] = Mn[i]; }
        return;
    }
    u32 Ph[NA], Mh[NA];
    u32 carry = 0;
    u32 hpos = 0, hneg = 0;
    if constexpr (CHAIN) { hpos = cin & 1u; hneg = (cin >> 1) & 1u; cin >>= 2; }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const u32 eq = (CHAIN && i == 0) ? (Eq[0] | hneg) : Eq[i];      // Eq |= hinIsNeg (:423)
        const u32 t = eq & Pv[i];
        u32 cout_;
        const u32 s = __builtin_addc(t, Pv[i], carry, &cout_);
        carry = cout_;
        const u32 Xh = (s ^ Pv[i]) | eq;
        Ph[i] = Mv[i] | ~(Xh | Pv[i]);
        Mh[i] = Pv[i] & Xh;
    }
    if constexpr (CHAIN) {
        const u32 x2 = (Ph[NA - 1] >> 31) | ((Mh[NA - 1] >> 30) & 2u);
        cout = __builtin_amdgcn_alignbit(x2, cout, 2);                  // (x2 << 30) | (cout >> 2)
    }
    if (NA == NWD) {                                   // bottom row is in the band: follow its score
        if constexpr (NWD <= 8) {
            e += (int)((Ph[NA - 1] >> sh) & 1u);
            e -= (int)((Mh[NA - 1] >> sh) & 1u);
        } else {
            // groups of 12 / 16 / 24 / 32 words hold queries of 9..12 / 13..16 / 17..24 / 25..32 words: sh is m - 1
            // itself, the row's word (one of the last four or eight) is picked per lane
            const u32 lw = sh >> 5;
            u32 phs = Ph[NWD - 1], mhs = Mh[NWD - 1];
#pragma unroll
            for (int d = 2; d <= (NWD > 16 ? 8 : 4); ++d) {
                const bool here = lw == (u32)(NWD - d);
                phs = here ? Ph[NWD - d] : phs;
                mhs = here ? Mh[NWD - d] : mhs;
            }
            e += (int)((phs >> (sh & 31u)) & 1u);
            e -= (int)((mhs >> (sh & 31u)) & 1u);
        }
        flag |= e;
    }
#pragma unroll
    for (int i = NA - 1; i >= 0; --i) {
        u32 ph, mh;
        if (i > 0) {
            ph = __builtin_amdgcn_alignbit(Ph[i], Ph[i - 1], 31);
            mh = __builtin_amdgcn_alignbit(Mh[i], Mh[i - 1], 31);
        } else if constexpr (CHAIN) {                   // << 1 with the delta of the row above shifted in (:435-441)
            ph = (Ph[0] << 1) | hpos;
            mh = (Mh[0] << 1) | hneg;
        } else {                                        // << 1 with a zero shifted in (HW row -1)
            asm("v_add_u32 %0, %1, %1" : "=v"(ph) : "v"(Ph[0]));
            asm("v_add_u32 %0, %1, %1" : "=v"(mh) : "v"(Mh[0]));
        }
        const u32 Xv = Eq[i] | Mv[i];
        Pv[i] = mh | ~(Xv | ph);
        Mv[i] = ph & Xv;
    }
}

// Band heights the banded kernel is unrolled for: every height up to 8 words; the groups of 12, 16, 24 and 32 words
// (reads of 257..1024 bases) step 1, 2, 3, 4, 6, 8, 12, 16, 24, 32 -- the code of a height is four unrolled quads of NA
// words each, and a band that tall is moving fast anyway.
template <int NWD> __host__ __device__ constexpr bool band_height_ok(int h)
{
    return h >= 1 && h <= NWD && (NWD <= 8 || h <= 4 || h == 6 || h == 8 || h == 12 || h == 16 || h == 24 || h == NWD);
}
template <int NWD> __host__ __device__ constexpr int band_height_up(int h)   { int n = h + 1; while (n < NWD && !band_height_ok<NWD>(n)) ++n; return n; }
template <int NWD> __host__ __device__ constexpr int band_height_down(int h) { int n = h - 1; while (n > 1 && !band_height_ok<NWD>(n)) --n; return n; }

static_assert(band_height_up<5>(3) == 4 && band_height_down<5>(3) == 2 && band_height_up<8>(7) == 8, "up to 8 words: every height");
static_assert(band_height_up<12>(4) == 6 && band_height_up<12>(8) == 12 && band_height_down<12>(12) == 8 && band_height_down<12>(6) == 4, "12-word ladder");
static_assert(band_height_up<16>(12) == 16 && band_height_down<16>(16) == 12, "16-word ladder");
static_assert(band_height_up<24>(16) == 24 && band_height_down<24>(24) == 16 && band_height_up<24>(12) == 16, "24-word ladder");
static_assert(band_height_up<32>(24) == 32 && band_height_down<32>(32) == 24 && band_height_up<32>(16) == 24, "32-word ladder");
// the per-lane bottom row (word (m-1)/32) must be outside every height below the full one: the group's shortest
// read has NWD - 4 (NWD - 8 above 16 words) full words above its last one
static_assert(band_height_down<12>(12) <= 12 - 4 && band_height_down<16>(16) <= 16 - 4 && band_height_down<24>(24) <= 24 - 8 &&
              band_height_down<32>(32) <= 32 - 8, "bottom row inside the band only at full height");
// (the groups of 10 / 14 words hold reads of 9..10 / 13..14 words: the bottom row sits in one of the last two)
static_assert(band_height_up<10>(8) == 10 && band_height_down<10>(10) <= 10 - 2 && band_height_up<14>(12) == 14 && band_height_down<14>(14) <= 14 - 2,
              "10- and 14-word ladders");

struct HwTrack {            // per-lane tracking state of the banded kernel
    int best, cnt, cap;
    int* pos;
};

// Rows carried from quad to quad by the one-word band (the next quad's rows, already in registers); local to a run
// of that band height (nothing of it is live at any other height).
typedef u32 QuadRows[4];

// Four columns (one SGPR pair of the expanded target: lo / hi) with NA active words, then -- depending on NA and
// on the position Q of the quad in its 16-column block -- the band checkpoint.  nlo / nhi: the NEXT quad's pair
// (the one-word band requests its rows while this quad computes).  Returns the new number of active words.
//
// Checkpoint intervals: a band of ONE word is re-examined every 4 columns, of two words every 8, of more
// every 16 (at the end of the block).  With an interval of c columns the band must grow when the computed score S
// of its bottom row is <= k + c - 1: a cell <= k below the band at column j + d (d <= c) has its diagonal
// predecessor <= k in the band's bottom row at column j + d - 1 (values never decrease along a diagonal), which is
// then exact, and horizontal neighbours differ by at most 1, so S(j) <= k + d - 1 <= k + c - 1.  (Round 1 grew at
// S <= k + c; the one unit matters: against unrelated sequence the score 32 rows down hovers around 13, and with
// k = 6 a wave meets S <= 10 at 0.5 % of its checkpoints but S <= 9 at 0.06 %.)  A new word enters as "+1 per row"
// like the reference's new block (edlib.cpp:605-608).
// FILTER (the piece filter of long reads, long_reads.hip; its own instantiation, so that the scans of whole reads carry none of
// it): the threshold stays where it is and the lane lists the 16-column blocks that hold a column scoring <= best, each once.
template <int NA, int NWD, int Q, int S, bool CHECK = true, bool CHAIN = false, bool FILTER = false>
__device__ __forceinline__ int band_quad(const u32 lo, const u32 hi, const u32 nlo, const u32 nhi, QuadRows& qr,
                                         const int colBase, const int colEnd, const bool track, u32 (&Pv)[NWD],
                                         u32 (&Mv)[NWD], int& e, int& flag, HwTrack& tr, const u32 sh, const int lastRows,
                                         u32& cin, u32& cout)
{
    int eh[4];
    if constexpr (NA == 1 && NWD > 1) {
        quad_one_word(qr, Pv[0], Mv[0], nlo, nhi);                              // nothing tracked: the bottom row is outside
    } else {
        // straight-line code: the Peq rows of a column arrive from LDS while the previous column is computed.
        // The first request of a quad is exposed; the other waves of the SIMD cover it.
        u32 nx[NA];
        lds_rows_request<NA, 0, S>(nx, lo, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lds_rows_wait<NA>(nx);
            u32 eq[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) eq[i] = nx[i];
            if (j == 0) lds_rows_request<NA, 1, S>(nx, lo, hi);
            if (j == 1) lds_rows_request<NA, 2, S>(nx, lo, hi);
            if (j == 2) lds_rows_request<NA, 3, S>(nx, lo, hi);
            if constexpr (NA == 2 && NWD > 2) column_step_eq2<NWD>(eq[0], eq[1], Pv, Mv);   // bottom row outside: nothing tracked
            else column_step_hw<NA, NWD, CHAIN>(eq, Pv, Mv, e, flag, sh, cin, cout);
            eh[j] = e;
        }
    }
    if (NA == NWD) {
        if (FILTER && track && __builtin_amdgcn_ballot_w64(flag < 0) != 0ull) {   // wave-uniform
            // the four columns of a quad share their 16-column block
            const int blk = colBase >> 4;
            bool hit = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) hit = hit || (eh[j] < 0 && colBase + j < colEnd);
            // the last listed block is re-read (rare path) rather than kept in a register of the hot loop; a list that
            // has overflowed is discarded by the host, so duplicates past the cap do not matter
            bool fresh = hit;
            if (hit && tr.cnt > 0 && tr.cnt <= tr.cap) fresh = tr.pos[tr.cnt - 1] != blk;
            if (fresh && tr.cnt < tr.cap) tr.pos[tr.cnt] = blk;
            tr.cnt += fresh ? 1 : 0;
            flag = 0;                                                   // e stays relative to the fixed threshold
        } else if (!FILTER && track && __builtin_amdgcn_ballot_w64(flag < 0) != 0ull) {   // wave-uniform
            const int bestIn = tr.best;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = colBase + j;
                const int sc = eh[j] + bestIn + 1;
                const bool hit = (sc <= tr.best) && (col < colEnd);     // edlib.cpp:658-673
                const bool better = hit && (sc < tr.best);
                tr.cnt = better ? 0 : tr.cnt;
                tr.best = better ? sc : tr.best;
                if (hit && tr.cnt < tr.cap) tr.pos[tr.cnt] = col;
                tr.cnt += hit ? 1 : 0;
            }
            e = eh[3] + bestIn - tr.best;                               // rebase e on the new best
            flag = 0;
        }
    }
    // ---- band checkpoints.  Scores are computed values: exact when <= k, otherwise upper bounds that still
    // exceed k, which is all the rules use.  HW: the row above the band's top is all zeros.
    if constexpr (!CHECK) return NA;                                        // scan_reads_full_kernel: the band is the query
    else if constexpr (NA == 1) {
        if constexpr (NWD > 1) {
            // S1 = popc(Pv) - popc(Mv) <= k + 3: second word.  Written so that the loop-invariant k + 3 rides
            // in the accumulator operand of v_bcnt: two v_bcnt and one v_cmp per quad
            const int up = __popc(Pv[0]), dn = __popc(Mv[0]) + (tr.best + 3);
            if (__builtin_amdgcn_ballot_w64(up <= dn) != 0ull) {
                Pv[1] = ~0u; Mv[1] = 0u;                                    // "+1 per row", edlib.cpp:605-608
                if (NWD == 2) { e = (up - __popc(Mv[0])) + lastRows - tr.best - 1; flag = 0; }
                return 2;
            }
        }
        return 1;
    } else if constexpr (NA == 2) {
        if (Q & 1) {                                                        // every 8 columns
            const int S1 = __popc(Pv[0]) - __popc(Mv[0]);
            const int S2 = S1 + __popc(Pv[1]) - __popc(Mv[1]);
            if constexpr (NWD > 2) {
                if (__builtin_amdgcn_ballot_w64(S2 <= tr.best + 7) != 0ull) {
                    Pv[2] = ~0u; Mv[2] = 0u;
                    if (NWD == 3) { e = S2 + lastRows - tr.best - 1; flag = 0; }
                    return 3;
                }
            }
            // back to one word when (a) every cell of the second word exceeds k -- in a span of 8 rows between
            // scores A (row above it) and B (its last row) a cell j rows down is >= max(A - j, B - (8 - j))
            // >= (A + B - 8) / 2 -- and (b) the bottom 4 rows of the first word do too
            const int k2 = 2 * tr.best + 9;
            const int sa = S1 + __popc(Pv[1] & 0xffu) - __popc(Mv[1] & 0xffu);
            const int sb = S1 + __popc(Pv[1] & 0xffffu) - __popc(Mv[1] & 0xffffu);
            const int sc = S1 + __popc(Pv[1] & 0xffffffu) - __popc(Mv[1] & 0xffffffu);
            const bool keep = (S1 <= tr.best + 4) || (S1 + sa <= k2) || (sa + sb <= k2) || (sb + sc <= k2) || (sc + S2 <= k2);
            if (__builtin_amdgcn_ballot_w64(keep) == 0ull) return 1;
        }
        return 2;
    } else {
        if (Q == 3) {                                                       // end of the block: every 16 columns
            constexpr int DN = band_height_down<NWD>(NA);                   // NA - 1 up to 8 words
            int cum[NA];                                                    // score of the last row of every word
            {
                int acc = 0;
#pragma unroll
                for (int i = 0; i < NA; ++i) { acc += __popc(Pv[i]) - __popc(Mv[i]); cum[i] = acc; }
            }
            const int Sb = cum[NA - 1];
            if constexpr (NA < NWD) {
                if (__builtin_amdgcn_ballot_w64(Sb <= tr.best + 15) != 0ull) {
                    constexpr int UP = band_height_up<NWD>(NA);             // NA + 1 up to 8 words
#pragma unroll
                    for (int i = NA; i < UP; ++i) { Pv[i < NWD ? i : 0] = ~0u; Mv[i < NWD ? i : 0] = 0u; }
                    // row m-1 is lastRows + 32 (NWD - 1 - NA) rows below the band's bottom row
                    if (UP == NWD) { e = Sb + lastRows + 32 * (NWD - 1 - NA) - tr.best - 1; flag = 0; }
                    return UP;
                }
            }
            // drop the words from DN on when (a) every cell of them exceeds k: in a word between the scores A (row
            // above it) and B (its last row) a cell j rows down is >= max(A - j, B - (32 - j)) >= (A + B - 32) / 2,
            // and (b) the new bottom 16 rows do too
            bool keep = cum[DN - 1] <= tr.best + 16;
#pragma unroll
            for (int i = DN; i < NA; ++i) keep = keep || (cum[i - 1] + cum[i] <= 2 * tr.best + 34);
            if (__builtin_amdgcn_ballot_w64(keep) == 0ull) return DN;
        }
        return NA;
    }
}

typedef u32 u32x8 __attribute__((ext_vector_type(8)));

// S: Peq rows per word = target symbols rounded up to 4, 8 or 16 (a genome with N, soft-masked lower case, IUPAC
// codes).  LDS per wave = NWD * S * 256 bytes: 8 waves per SIMD at S = 4 and up to 5 words, 4 at S = 8, 2 at S = 16.
template <int NWD, int S, bool FILTER = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((S == 4 && NWD <= 5) ? 8 : 1, 8)))
scan_reads_banded_kernel(const ReadScanArgs a)
{
    const int lane = threadIdx.x;
    const int rblk = blockIdx.x;                                      // one wave per workgroup
    const int idx = rblk * 64 + lane;
    const bool live = idx < a.nlanes;
    const int slot = live ? (a.slotmap ? a.slotmap[idx] : idx) : 0;

    u32 Pv[NWD], Mv[NWD];
    const int m = a.qlen[slot];
    const u32 sh = NWD <= 8 ? (u32)(m - 1) & 31u : (u32)(m - 1);      // row m-1 inside the last word (12 / 16 words: m - 1, column_step_hw)
    const int lastRows = m - 32 * (NWD - 1);                          // query rows in the last word (12 / 16 words: may be <= 0)
    // the four Peq rows of the wave's queries: HBM -> LDS, [word][symbol][lane].  The only LDS object of a
    // one-wave workgroup: it sits at LDS address 0, which is what lets M0 be the bare row offset.
    __shared__ __attribute__((aligned(1024))) u32 s_eq[NWD][S][64];
    {
        const size_t pb = (size_t)(slot >> 6) * S * NWD * 64 + (slot & 63);
#pragma unroll
        for (int d = 0; d < NWD; ++d) {
#pragma unroll
            for (int sy = 0; sy < S; ++sy) s_eq[d][sy][lane] = a.peq[pb + (size_t)(sy * NWD + d) * 64];
            Pv[d] = ~0u;                                             // column -1: D[i][-1] = i+1
            Mv[d] = 0u;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // (s_eq at LDS address 0 is checked on the host at the launch: lds_check.hpp)
    HwTrack tr;
    {
        const int k0 = a.kinit[slot];
        tr.best = k0 < a.kcap ? k0 : a.kcap;
    }
    tr.cnt = 0;
    {
        // lanes past nlanes (the tail of the last wave) own no record: they must not even read the tables
        const long long item = (long long)idx * a.numSegments + blockIdx.y;   // (lane, segment) record
        tr.cap = live ? (a.posCap ? a.posCap[item] : a.cap) : 0;
        tr.pos = a.segPos + (!live ? 0 : (a.posOff ? a.posOff[item] : item * a.cap));
    }
    int e = m - tr.best - 1;                                          // score at column -1 is m
    int flag = 0;

    const int T = a.targetLength;
    const int c0 = blockIdx.y * a.segLen;
    int c1 = c0 + a.segLen; if (c1 > T) c1 = T;
    int cw = c0 - a.warm; if (cw < 0) cw = 0;
    const int b0 = cw >> 4, bmain = c0 >> 4, bend = (c1 + 15) >> 4;   // blocks of 16 columns
    // constant address space: the blocks are read with s_load_dwordx8 whatever the asm blocks clobber (a plain global
    // pointer turns into vector loads behind the first "memory" clobber, and the rows' "s" operands into VGPRs)
    typedef const u32x8 __attribute__((address_space(4))) * TargetBlocks;
    const TargetBlocks tx = (TargetBlocks)(unsigned long long)a.trows;
    int nw = NWD;
    u32 noChain = 0;                                                  // (the chain of strips is scan_reads_full_kernel's)
    unsigned int bandWork = 0;                                        // sum of nw over the quads (wave-uniform)
    // One inner loop per band height: a quad that leaves the height unchanged falls into the next quad of the same
    // code with every live value where it was.  The 16 columns of a block are four unrolled quads of straight-line
    // code; a run that starts in the middle of a block (the height changed there) first finishes that block quad
    // by quad (an entry switch into the unrolled body made hipcc shuffle the rows and Pv / Mv between registers
    // after every quad: 8 v_mov per 40 useful instructions).  cur / nxt: blocks b and b + 1 in SGPRs (the buffer
    // is padded by two blocks); the load of block b + 2's predecessor is issued a whole block ahead of its use.
    int b = b0, q = 0;
    u32x8 cur = tx[b0], nxt = tx[b0 + 1];
    while (b < bend) {
        switch (nw) {
#define QUAD(NA, Q)                                                                                             \
            nw = band_quad<(NA <= NWD ? NA : NWD), NWD, Q, S, true, false, FILTER>(cur[2 * Q], cur[2 * Q + 1],     \
                     Q < 3 ? cur[(2 * Q + 2) & 7] : nxt[0], Q < 3 ? cur[(2 * Q + 3) & 7] : nxt[1], qr,          \
                     b * 16 + Q * 4, c1, b >= bmain /* warm-up columns record nothing */, Pv, Mv, e, flag, tr,  \
                     sh, lastRows, noChain, noChain);
#define ADVANCE { q = 0; ++b; cur = nxt; nxt = tx[b + 1]; }
#define CASE(NA) case NA:                                                                                       \
            if constexpr (band_height_ok<NWD>(NA)) {                                                            \
                const int q0 = b * 4 + q;                                                                       \
                QuadRows qr;                                                                                    \
                /* the one-word band starts with the rows of its first quad in registers */                     \
                if (NA == 1 && NWD > 1) {                                                                       \
                    const u32 l = q == 0 ? cur[0] : q == 1 ? cur[2] : q == 2 ? cur[4] : cur[6];                 \
                    const u32 h = q == 0 ? cur[1] : q == 1 ? cur[3] : q == 2 ? cur[5] : cur[7];                 \
                    quad_rows_request1(qr, l, h); quad_rows_wait1(qr);                                          \
                }                                                                                               \
                /* head: the rest of a block entered in the middle (a quad always advances the position) */     \
                if (q == 1) { QUAD(NA, 1) q = 2; }                                                              \
                if (q == 2 && nw == NA) { QUAD(NA, 2) q = 3; }                                                  \
                if (q == 3 && nw == NA) { QUAD(NA, 3) ADVANCE }                                                 \
                /* steady state: whole blocks, four quads of straight-line code per trip */                     \
                while (q == 0 && nw == NA && b < bend) {                                                        \
                    QUAD(NA, 0) if (nw != NA) { q = 1; break; }                                                 \
                    QUAD(NA, 1) if (nw != NA) { q = 2; break; }                                                 \
                    QUAD(NA, 2) if (nw != NA) { q = 3; break; }                                                 \
                    QUAD(NA, 3) ADVANCE                                                                         \
                }                                                                                               \
                bandWork += (unsigned int)NA * (unsigned int)(b * 4 + q - q0);                                  \
            }                                                                                                   \
            break;
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(12) CASE(14) CASE(16) CASE(24) CASE(32)
#undef CASE
#undef ADVANCE
#undef QUAD
        }
    }
    if (live) {
        const long long it = (long long)(blockIdx.x * 64 + threadIdx.x) * a.numSegments + blockIdx.y;
        a.segBest[it] = tr.best;
        a.segCnt[it] = tr.cnt;
    }
    if (a.wordSteps && lane == 0) atomicAdd(a.wordSteps, (unsigned long long)bandWork * 4ull * 64ull);
}

// Every row of every column, HW mode, with the banded kernel's data path (Peq rows in LDS picked by M0, the target as
// row offsets through s_load_dwordx8, one wave per workgroup) but no band bookkeeping at all: what the leftovers of the
// k-doubling run on when a sample shows that their band is the whole query (unrelated reads).  Against
// scan_reads_kernel (register-resident rows, a scalar 4-way branch per column) it has no symbol dispatch and takes 4, 8 or
// 16 symbols; against scan_reads_banded_kernel at full height it has no checkpoints and a third of the code.
// CHAIN: the lane is one STRIP of a taller query (rows [1024 s, 1024 s + 32 NWD) of it): the horizontal deltas of the strip
// above come in through a.chainIn (one dword per 16 columns: two bits per column), those of the strip's own bottom row go
// out through a.chainOut, both laid out [segment][block of 16 columns][lane of the producing launch].  A query of any
// length then runs at this kernel's cost per row (10 VALU ops per 32 rows and column, every lane busy) as a sequence of
// launches, one per strip level, where kernel W spends ~40 instructions per 64-row block in a wave whose lanes are only
// as busy as the query is tall.  Strips above the last one follow no score (their threshold is -1: nothing qualifies).
template <int NWD, int S, bool CHAIN = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((S == 4 && NWD <= 5) ? 7 : 1, 8)))
scan_reads_full_kernel(const ReadScanArgs a)
{
    const int lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    const bool live = idx < a.nlanes;
    const int slot = live ? (a.slotmap ? a.slotmap[idx] : idx) : 0;
    u32 Pv[NWD], Mv[NWD];
    const int m = a.qlen[slot];
    const u32 sh = NWD <= 8 ? (u32)(m - 1) & 31u : (u32)(m - 1);
    const int lastRows = m - 32 * (NWD - 1);
    __shared__ __attribute__((aligned(1024))) u32 s_eq[NWD][S][64];
    {
        const size_t pb = (size_t)(slot >> 6) * S * NWD * 64 + (slot & 63);
#pragma unroll
        for (int d = 0; d < NWD; ++d) {
#pragma unroll
            for (int sy = 0; sy < S; ++sy) s_eq[d][sy][lane] = a.peq[pb + (size_t)(sy * NWD + d) * 64];
            Pv[d] = ~0u; Mv[d] = 0u;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // (s_eq at LDS address 0 is checked on the host at the launch: lds_check.hpp)
    HwTrack tr;
    tr.best = a.kinit[slot];
    tr.cnt = 0;
    {
        const long long item = (long long)idx * a.numSegments + blockIdx.y;
        tr.cap = live ? (a.posCap ? a.posCap[item] : a.cap) : 0;
        tr.pos = a.segPos + (!live ? 0 : (a.posOff ? a.posOff[item] : item * a.cap));
    }
    int e = m - tr.best - 1, flag = 0;
    if constexpr (CHAIN) if (a.rowBase) e += a.rowBase[slot];        // column -1: D[i][-1] = i + 1 counts the rows of the strips above
    const int T = a.targetLength;
    const int c0 = blockIdx.y * a.segLen;
    int c1 = c0 + a.segLen; if (c1 > T) c1 = T;
    int cw = c0 - a.warm; if (cw < 0) cw = 0;
    const int b0 = cw >> 4, bmain = c0 >> 4, bend = (c1 + 15) >> 4;
    typedef const u32x8 __attribute__((address_space(4))) * TargetBlocks;
    const TargetBlocks tx = (TargetBlocks)(unsigned long long)a.trows;
    u32x8 cur = tx[b0];
    QuadRows qr;
    // chain streams of this (lane, segment): dword (segment * chainBlocks + block - b0) * lanes + lane
    const u32* cinP = nullptr; u32* coutP = nullptr;
    if constexpr (CHAIN) {
        const long long sb = (long long)blockIdx.y * a.chainBlocks;
        if (a.chainIn && live) cinP = a.chainIn + sb * a.chainInLanes + a.chainSrc[idx];
        if (a.chainOut && live) coutP = a.chainOut + sb * a.nlanes + idx;
    }
    u32 cin = 0, cout = 0;
    for (int b = b0; b < bend; ++b) {
        const u32x8 nxt = tx[b + 1];
        if constexpr (CHAIN) cin = cinP ? cinP[(long long)(b - b0) * a.chainInLanes] : 0u;
#define QUADF(Q) (void)band_quad<NWD, NWD, Q, S, false, CHAIN>(cur[2 * Q], cur[2 * Q + 1], 0u, 0u, qr, b * 16 + Q * 4, c1, b >= bmain, \
                                                              Pv, Mv, e, flag, tr, sh, lastRows, cin, cout);
        QUADF(0) QUADF(1) QUADF(2) QUADF(3)
#undef QUADF
        if constexpr (CHAIN) if (coutP) coutP[(long long)(b - b0) * a.nlanes] = cout;
        cur = nxt;
    }
    if (live) {
        const long long it = (long long)idx * a.numSegments + blockIdx.y;
        a.segBest[it] = tr.best;
        a.segCnt[it] = tr.cnt;
    }
}


}  // namespace edlib_amd
