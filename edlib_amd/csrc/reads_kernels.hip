// reads_kernels.hip -- gfx950 kernels for "many short queries, one shared target".
//
// Replaces, for BASELINE.json configs 2/3, the reference's per-query call of
// myersCalcEditDistanceSemiGlobal (edlib.cpp:550-704) + buildPeq (:358-384) +
// calculateBlock (:412-447) and the k-doubling loop around it (:197-217).  Design:
//
//   * one wave64 = 64 queries ("slots") x one segment of the target; every
//     lane owns one query, so the target symbol of a column is WAVE-UNIFORM:
//     it comes from the packed target as a scalar and picks which of the lane's
//     Peq rows feeds the column -- in the banded kernel the target is stored as
//     the LDS row offset itself (16 bits per column, s_load_dwordx8) and goes
//     into M0 with one scalar instruction for a ds_read_addtid_b32 (4, 8 or 16
//     rows per word: targets of up to 16 symbols); in the plain kernel a scalar
//     4-way branch over register rows (2-bit packed target, four symbols).  No
//     cross-lane traffic and no divergence in the DP itself.
//   * the query column lives in VGPRs as NWD 32-bit words (Pv, Mv) instead of
//     the reference's 64-bit blocks: 150 rows need 5 words (160 rows) rather
//     than 3 blocks (192 rows).  The 64-bit add of calculateBlock becomes a
//     v_add_co/v_addc_co carry chain, the <<1 a v_alignbit chain, and every
//     3-input boolean a single v_bitop3_b32 (gfx950).  10 VALU ops per word
//     per column.
//   * the score of the bottom query row is followed directly at bit (m-1) of
//     the last word, so no wildcard padding (the reference's W) is needed and
//     end positions are produced un-shifted.
//   * HW is shift-invariant: a segment that starts 2m-1 columns early from the
//     fresh state reproduces the exact bottom-row scores of its own columns,
//     so the target is cut into segments for load balance and for small
//     batches; merge_segments() joins them.
//
// Two scan kernels:
//   scan_reads_kernel<NWD, MODE>        every row of every column (SHW, NW; HW for the leftovers of the
//                                       k-doubling whose band is the whole query, and with EDLIB_AMD_BAND=0).
//                                       All outputs are functions of the full DP matrix, so no band is
//                                       needed for correctness and the kernel never branches on data.
//   scan_reads_banded_kernel<NWD, S>    HW: Ukkonen band per wave + k-doubling (the bench kernel, §3b); one wave per
//                                       workgroup, S = 4 / 8 / 16 Peq rows per word.
#include "reads_kernels.hpp"
#include "reads_scan.hpp"
#include "reads_column_asm.hpp"

namespace edlib_amd {

// ------------------------------------------------------------ target packing

__global__ void __launch_bounds__(256)
pack_target_2bit_kernel(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ lut,
                        int T, u32* __restrict__ tpk, int nwords)
{
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const int base = w * 16;
    u32 out = 0;
    if (base + 16 <= T) {
        const uint4 v = *reinterpret_cast<const uint4*>(raw + base);   // 16 B per lane, coalesced
        const u32 d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i)
            out |= (u32)(s_lut[(d[i >> 2] >> ((i & 3) * 8)) & 0xFF] & 3) << (2 * i);
    } else {
        for (int i = 0; i < 16 && base + i < T; ++i)
            out |= (u32)(s_lut[raw[base + i]] & 3) << (2 * i);
    }
    tpk[w] = out;
}

// Expanded target for the banded kernel: 16 bits per column = LDS row offset of the column's Peq rows
// (symbol << 8), so that one scalar instruction per column puts it into M0.  32 columns per thread.
__global__ void __launch_bounds__(256)
pack_target_rows_kernel(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ lut, int T,
                        u32* __restrict__ trows, int ndwords)
{
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int w = blockIdx.x * blockDim.x + threadIdx.x;               // dword = 2 columns
    if (w >= ndwords) return;
    const int c = 2 * w;
    const u32 a0 = c < T ? (u32)(s_lut[raw[c]] & 15) : 0u, a1 = c + 1 < T ? (u32)(s_lut[raw[c + 1]] & 15) : 0u;
    trows[w] = (a0 << 8) | (a1 << 24);
}

hipError_t launch_pack_target_rows(const uint8_t* raw, const uint8_t* lut, int T, u32* trows, int ndwords,
                                   hipStream_t stream)
{
    if (ndwords == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_target_rows_kernel, dim3((ndwords + 255) / 256), dim3(256), 0, stream,
                       raw, lut, T, trows, ndwords);
    return hipGetLastError();
}

hipError_t launch_pack_target_2bit(const uint8_t* raw, const uint8_t* lut, int T, u32* tpk,
                                   hipStream_t stream)
{
    const int nwords = (T + 15) / 16;
    if (nwords == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_target_2bit_kernel, dim3((nwords + 255) / 256), dim3(256), 0, stream,
                       raw, lut, T, tpk, nwords);
    return hipGetLastError();
}

// ------------------------------------------------------------------ buildPeq

// reference buildPeq (edlib.cpp:358-384) restricted to the (<= 16) symbols of the
// target: bit i of row s says "query[i] equals target symbol s".  eqtbl[byte] is
// the 16-bit set of target symbols a query byte equals (identity plus
// additionalEqualities, edlib.cpp:63-94).  Rows at or past the query end stay 0.
// Output layout [readBlock][sym (S = 4, 8 or 16)][word][lane]: a wave loads a row as one 256 B line.
//
// One wave builds the rows of its 64 reads COOPERATIVELY: for read j the lanes load 64 consecutive query bytes
// (one coalesced line), look their symbol sets up, and a ballot per symbol IS the two 32-bit Peq words of those 64
// rows, which lane j keeps.  (Round 1 gave every lane its own read and walked it byte
// by byte: 64 different lines per load instruction, 8.9 GB of fetch per 1M reads against 150 MB of queries.)
// Symbols are done four at a time (4 x NWD row registers); the query bytes are re-read per group from L1 / L2.
template <int NWD>
__global__ void __launch_bounds__(256)
build_peq_reads_kernel(const uint8_t* __restrict__ reads, const long long* __restrict__ qoff,
                       const int* __restrict__ perm, int nslots, int S,
                       const uint16_t* __restrict__ eqtbl, const u32* __restrict__ tpres, int kcfg,
                       u32* __restrict__ peq, int* __restrict__ qlen, int* __restrict__ kinit,
                       int* __restrict__ alphaExtra)
{
    __shared__ uint16_t s_eq[256];
    s_eq[threadIdx.x] = eqtbl[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);               // read block of this wave
    if (blk * 64 >= nslots) return;
    const int slot = blk * 64 + lane;
    const int r = slot < nslots ? perm[slot] : -1;
    long long off = 0; int m = 1;
    if (r >= 0) { off = qoff[r]; m = (int)(qoff[r + 1] - off); }
    int extra = 0;
    for (int g0 = 0; g0 < S; g0 += 4) {
        u32 E[4][NWD];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int d = 0; d < NWD; ++d) E[s][d] = 0;
        for (int j = 0; j < 64; ++j) {                                 // read j of the wave (wave-uniform)
            const int rj = __builtin_amdgcn_readlane(r, j);
            if (rj < 0) continue;
            const int mj = __builtin_amdgcn_readlane(m, j);
            const long long offj = ((long long)__builtin_amdgcn_readlane((int)(off >> 32), j) << 32)
                                 | (u32)__builtin_amdgcn_readlane((int)off, j);
            // bytes of this read that do not occur in the target, counted once each (alphabetLength): wave-uniform
            // 256-bit set, only touched when a lane holds such a byte (first symbol group only)
            unsigned long long seen0 = 0, seen1 = 0, seen2 = 0, seen3 = 0;
            int extraJ = 0;
            const bool mine = lane == j;                               // the lane that keeps read j's rows
#pragma unroll
            for (int c = 0; c < (NWD + 1) / 2; ++c) {                  // 64 rows = two words per trip
                const int i = 64 * c + lane;
                const bool in = i < mj;
                const u32 by = in ? reads[offj + i] : 0u;
                const u32 mask = in ? ((u32)s_eq[by] >> g0) : 0u;
                const unsigned long long b0 = __builtin_amdgcn_ballot_w64((mask & 1u) != 0), b1 = __builtin_amdgcn_ballot_w64((mask & 2u) != 0);
                const unsigned long long b2 = __builtin_amdgcn_ballot_w64((mask & 4u) != 0), b3 = __builtin_amdgcn_ballot_w64((mask & 8u) != 0);
                constexpr int w0 = 0;
                const int d0 = 2 * c, d1 = (2 * c + 1 < NWD) ? 2 * c + 1 : w0;
                E[0][d0] = (mine ? (u32)b0 : E[0][d0]); E[1][d0] = (mine ? (u32)b1 : E[1][d0]);
                E[2][d0] = (mine ? (u32)b2 : E[2][d0]); E[3][d0] = (mine ? (u32)b3 : E[3][d0]);
                if (2 * c + 1 < NWD) {
                    E[0][d1] = (mine ? (u32)(b0 >> 32) : E[0][d1]); E[1][d1] = (mine ? (u32)(b1 >> 32) : E[1][d1]);
                    E[2][d1] = (mine ? (u32)(b2 >> 32) : E[2][d1]); E[3][d1] = (mine ? (u32)(b3 >> 32) : E[3][d1]);
                }
                if (g0 == 0) {
                    unsigned long long np = __builtin_amdgcn_ballot_w64(in && !((tpres[by >> 5] >> (by & 31)) & 1u));
                    while (np) {                                       // rare: a query byte the target does not have
                        const int l = __builtin_ctzll(np);
                        const u32 v = (u32)__builtin_amdgcn_readlane((int)by, l);
                        const unsigned long long bit = 1ull << (v & 63);
                        unsigned long long& sw = (v >> 6) == 0 ? seen0 : (v >> 6) == 1 ? seen1 : (v >> 6) == 2 ? seen2 : seen3;
                        if (!(sw & bit)) { sw |= bit; ++extraJ; }
                        np &= ~__builtin_amdgcn_ballot_w64(in && by == v);
                    }
                }
            }
            if (g0 == 0 && mine) extra = extraJ;
        }
        if (slot < nslots) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int d = 0; d < NWD; ++d)
                    peq[((size_t)(blk * S + g0 + s) * NWD + d) * 64 + lane] = E[s][d];
        }
    }
    if (slot >= nslots) return;
    qlen[slot] = m;
    // candidates are columns scoring <= min(k, m): HW clamps k to m (edlib.cpp:566-568) and
    // for SHW the best score never exceeds m either (the empty prefix costs m)
    kinit[slot] = (kcfg < 0 || kcfg > m) ? m : kcfg;
    alphaExtra[slot] = extra;
}

template <int NWD>
static hipError_t launch_build_peq_t(const uint8_t* reads, const long long* qoff, const int* perm,
                                     int nslots, int S, const uint16_t* eqtbl, const u32* tpres, int kcfg,
                                     u32* peq, int* qlen, int* kinit, int* alphaExtra,
                                     hipStream_t stream)
{
    hipLaunchKernelGGL(build_peq_reads_kernel<NWD>, dim3((nslots + 255) / 256), dim3(256), 0, stream,
                       reads, qoff, perm, nslots, S, eqtbl, tpres, kcfg, peq, qlen, kinit, alphaExtra);
    return hipGetLastError();
}

hipError_t launch_build_peq_reads(int nwords, int syms, const uint8_t* reads, const long long* qoff,
                                  const int* perm, int nslots, const uint16_t* eqtbl,
                                  const u32* tpres, int kcfg, u32* peq, int* qlen, int* kinit,
                                  int* alphaExtra, hipStream_t stream)
{
    if (nslots == 0) return hipSuccess;
    if (syms != 4 && syms != 8 && syms != 16) return hipErrorInvalidValue;
    switch (nwords) {
#define CASE(N) case N: return launch_build_peq_t<N>(reads, qoff, perm, nslots, syms, eqtbl, tpres, kcfg, peq, qlen, kinit, alphaExtra, stream);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(12) CASE(14) CASE(16) CASE(24) CASE(32)
#undef CASE
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------- the scan

// One column of the Myers recurrence over NWD 32-bit words held in VGPRs
// (reference calculateBlock, edlib.cpp:412-447, with hin fixed by the top
// boundary: 0 for HW, +1 for SHW/NW -- hin is never negative at row -1, so the
// "Eq |= hinIsNeg" term vanishes).  Also advances the bottom-row score by the
// horizontal delta of row m-1 (bit `sh` of the last word).
template <int NWD, int MODE>
__device__ __forceinline__ void column_step(const u32 (&Eq)[NWD], u32 (&Pv)[NWD], u32 (&Mv)[NWD],
                                            int& score, const u32 sh)
{
    if constexpr (NWD <= 8) {
        // one asm statement: VOP3 encodings behind an alignment fence (reads_column_asm.hpp: why)
        u32 t_, s_, xh_, ph0_, ph1_, mh0_, mh1_, phs_, mhs_, xv_, Pn[NWD], Mn[NWD];
        int scoreN;
        unsigned long long cy_;
        if constexpr (MODE == 2) { RC_COLUMN_DISPATCH(NWD, RC_WORD0_HW) } else { RC_COLUMN_DISPATCH(NWD, RC_WORD0_NW) }
#pragma unroll
        for (int i = 0; i < NWD; ++i) { Pv[i] = Pn[i]; Mv[i] = Mn[i]; }
        score = scoreN;
        return;
    }
    u32 Ph[NWD], Mh[NWD];
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < NWD; ++i) {
        const u32 t = Eq[i] & Pv[i];
        u32 cout;
        const u32 s = __builtin_addc(t, Pv[i], carry, &cout);     // v_add_co / v_addc_co chain
        carry = cout;
        const u32 Xh = (s ^ Pv[i]) | Eq[i];
        Ph[i] = Mv[i] | ~(Xh | Pv[i]);
        Mh[i] = Pv[i] & Xh;
    }
    score += (int)__builtin_amdgcn_ubfe(Ph[NWD - 1], sh, 1) + __builtin_amdgcn_sbfe(Mh[NWD - 1], sh, 1);
#pragma unroll
    for (int i = NWD - 1; i >= 0; --i) {
        u32 ph, mh;
        if (i > 0) {
            ph = __builtin_amdgcn_alignbit(Ph[i], Ph[i - 1], 31);   // (Ph << 1) across words
            mh = __builtin_amdgcn_alignbit(Mh[i], Mh[i - 1], 31);
        } else {
            ph = (Ph[0] << 1) | (MODE == 2 ? 0u : 1u);             // row -1: HW 0, SHW/NW +1 (edlib.cpp:584,779)
            mh = Mh[0] << 1;
        }
        const u32 Xv = Eq[i] | Mv[i];
        Pv[i] = mh | ~(Xv | ph);
        Mv[i] = ph & Xv;
    }
}

// The distinct asm comments keep the four bodies from being tail-merged back
// into one body fed by v_mov/v_cndmask of the Peq row.
#define EDLIB_AMD_DISPATCH_COLUMN(sym)                                                     \
    switch (sym) {                                                                         \
        case 0:  column_step<NWD, MODE>(E0, Pv, Mv, score, sh); asm volatile("; sym0"); break; \
        case 1:  column_step<NWD, MODE>(E1, Pv, Mv, score, sh); asm volatile("; sym1"); break; \
        case 2:  column_step<NWD, MODE>(E2, Pv, Mv, score, sh); asm volatile("; sym2"); break; \
        default: column_step<NWD, MODE>(E3, Pv, Mv, score, sh); asm volatile("; sym3"); break; \
    }

// Record column `col` if it ties or improves the best bottom-row score
// (reference edlib.cpp:658-673: "colScore <= k", "positions.clear()", "k = bestScore").
#define EDLIB_AMD_TRACK(col)                                   \
    if (score <= best) {                                       \
        if (score < best) { best = score; cnt = 0; }           \
        if (cnt < cap) pos[cnt] = (col);                       \
        ++cnt;                                                 \
    }

template <int NWD, int MODE>
__global__ void __launch_bounds__(256)
scan_reads_kernel(const ReadScanArgs a)
{
    const int lane = threadIdx.x & 63;
    const int rblk = blockIdx.x * 4 + (threadIdx.x >> 6);          // 4 waves / workgroup
    const int seg = blockIdx.y;
    const int idx = rblk * 64 + lane;                              // lane index in this launch
    if (rblk * 64 >= a.nlanes) return;                             // whole wave out of range
    const bool live = idx < a.nlanes;
    const int slot = live ? (a.slotmap ? a.slotmap[idx] : idx) : 0;

    u32 E0[NWD], E1[NWD], E2[NWD], E3[NWD], Pv[NWD], Mv[NWD];
    {
        const size_t pb = (size_t)(slot >> 6) * 4 * NWD * 64 + (slot & 63);
#pragma unroll
        for (int d = 0; d < NWD; ++d) {
            E0[d] = a.peq[pb + (size_t)(0 * NWD + d) * 64];
            E1[d] = a.peq[pb + (size_t)(1 * NWD + d) * 64];
            E2[d] = a.peq[pb + (size_t)(2 * NWD + d) * 64];
            E3[d] = a.peq[pb + (size_t)(3 * NWD + d) * 64];
            Pv[d] = ~0u;                                           // column -1: D[i][-1] = i+1 (edlib.cpp:575-579)
            Mv[d] = 0u;
        }
    }
    const int m = a.qlen[slot];
    const u32 sh = (u32)(m - 1) & 31u;
    int score = m;
    int best = a.kinit[slot];
    int cnt = 0;
    const long long item = (long long)idx * a.numSegments + seg;   // (lane, segment) record
    // lanes past nlanes (the tail of the last wave) own no record: they must not even read the tables
    const int cap = !live ? 0 : (a.posCap ? a.posCap[item] : a.cap);
    int* pos = a.segPos + (!live ? 0 : (a.posOff ? a.posOff[item] : item * a.cap));

    const int T = a.targetLength;
    const int c0 = seg * a.segLen;                                 // multiple of 16
    int c1 = c0 + a.segLen; if (c1 > T) c1 = T;
    int cw = c0 - a.warm; if (cw < 0) cw = 0;
    cw &= ~15;

    // warm-up columns [cw, c0): state only, nothing is recorded (HW segments)
    for (int w = cw >> 4; w < (c0 >> 4); ++w) {
        const u32 tw = a.tpk[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 sym = (tw >> (2 * j)) & 3u;
            EDLIB_AMD_DISPATCH_COLUMN(sym)
        }
    }
    // full words of the segment
    const int wend = c1 >> 4;
    for (int w = c0 >> 4; w < wend; ++w) {
        const u32 tw = a.tpk[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 sym = (tw >> (2 * j)) & 3u;
            EDLIB_AMD_DISPATCH_COLUMN(sym)
            if (MODE != 0) { EDLIB_AMD_TRACK(w * 16 + j) }
        }
    }
    // ragged tail of the target (last segment only)
    const int rem = c1 - (wend << 4);
    if (rem > 0) {
        u32 tw = a.tpk[wend];
        for (int j = 0; j < rem; ++j) {
            const u32 sym = tw & 3u;
            tw >>= 2;
            EDLIB_AMD_DISPATCH_COLUMN(sym)
            if (MODE != 0) { EDLIB_AMD_TRACK(wend * 16 + j) }
        }
    }
    if (live) {
        const size_t o = (size_t)idx * a.numSegments + seg;
        a.segBest[o] = (MODE == 0) ? score : best;                 // NW: D[m][T] (edlib.cpp:914-917)
        a.segCnt[o] = (MODE == 0) ? 1 : cnt;
    }
}

template <int NWD>
static hipError_t launch_scan_mode(int mode, const ReadScanArgs& a, hipStream_t stream)
{
    const int nrblk = (a.nlanes + 63) / 64;
    dim3 grid((nrblk + 3) / 4, a.numSegments), block(256);
    switch (mode) {
        case 0: hipLaunchKernelGGL((scan_reads_kernel<NWD, 0>), grid, block, 0, stream, a); break;
        case 1: hipLaunchKernelGGL((scan_reads_kernel<NWD, 1>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((scan_reads_kernel<NWD, 2>), grid, block, 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_scan_reads(int nwords, int mode, const ReadScanArgs& a, hipStream_t stream)
{
    if (a.nlanes == 0) return hipSuccess;
    switch (nwords) {
#define CASE(N) case N: return launch_scan_mode<N>(mode, a, stream);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------- launchers (groups of up to 8 words; longer: reads_kernels_long.hip)

hipError_t launch_scan_reads_full_long(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream);
hipError_t launch_scan_reads_banded_long(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream);

template <int S, bool CHAIN>
static hipError_t launch_scan_reads_full_s(int nwords, const ReadScanArgs& a, hipStream_t stream)
{
    dim3 grid((a.nlanes + 63) / 64, a.numSegments), block(64);
    switch (nwords) {
#define CASE(N) case N: EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_full_kernel<N, S, CHAIN>), N * S * 256); \
                        hipLaunchKernelGGL((scan_reads_full_kernel<N, S, CHAIN>), grid, block, 0, stream, a); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_scan_reads_full(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream)
{
    if (a.nlanes == 0) return hipSuccess;
    if (nwords > kMaxReadWords) return launch_scan_reads_full_long(nwords, syms, a, stream);
    const bool chain = a.chainIn != nullptr || a.chainOut != nullptr;
    switch (syms) {
        case 4: return chain ? launch_scan_reads_full_s<4, true>(nwords, a, stream) : launch_scan_reads_full_s<4, false>(nwords, a, stream);
        case 8: return chain ? launch_scan_reads_full_s<8, true>(nwords, a, stream) : launch_scan_reads_full_s<8, false>(nwords, a, stream);
        case 16: return chain ? launch_scan_reads_full_s<16, true>(nwords, a, stream) : launch_scan_reads_full_s<16, false>(nwords, a, stream);
    }
    return hipErrorInvalidValue;
}

template <int S>
static hipError_t launch_scan_reads_banded_s(int nwords, const ReadScanArgs& a, hipStream_t stream)
{
    const int nrblk = (a.nlanes + 63) / 64;
    dim3 grid(nrblk, a.numSegments), block(64);
    switch (nwords) {
#define CASE(N) case N: if (a.filter) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<N, S, true>), N * S * 256); \
                                            hipLaunchKernelGGL((scan_reads_banded_kernel<N, S, true>), grid, block, 0, stream, a); } \
                        else { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<N, S, false>), N * S * 256); \
                               hipLaunchKernelGGL((scan_reads_banded_kernel<N, S, false>), grid, block, 0, stream, a); } break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_scan_reads_banded(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream)
{
    if (a.nlanes == 0) return hipSuccess;
    if (nwords > kMaxReadWords) return a.filter ? hipErrorInvalidValue : launch_scan_reads_banded_long(nwords, syms, a, stream);
    switch (syms) {
        case 4: return launch_scan_reads_banded_s<4>(nwords, a, stream);
        case 8: return launch_scan_reads_banded_s<8>(nwords, a, stream);
        case 16: return launch_scan_reads_banded_s<16>(nwords, a, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------- filter candidates

__global__ void __launch_bounds__(256)
collect_candidates_kernel(const int* __restrict__ segCnt, const int* __restrict__ segPos, int S, int cap, int nlanes,
                          int* __restrict__ out, int maxOut, int* __restrict__ counter, int* __restrict__ overflow)
{
    const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // (lane, segment) record
    if (item >= (long long)nlanes * S) return;
    const int c = segCnt[item];
    if (c <= 0) return;
    const int lane = (int)(item / S);
    if (c > cap) overflow[lane] = 1;
    const int take = c < cap ? c : cap;
    const int at = atomicAdd(counter, take);
    for (int i = 0; i < take; ++i)
        if (at + i < maxOut) { out[2 * (at + i)] = lane; out[2 * (at + i) + 1] = segPos[item * cap + i]; }
}

hipError_t launch_collect_candidates(const int* segCnt, const int* segPos, int numSegments, int cap, int nlanes,
                                     int* out, int maxOut, int* counter, int* overflow, hipStream_t stream)
{
    const long long items = (long long)nlanes * numSegments;
    if (items == 0) return hipSuccess;
    hipLaunchKernelGGL(collect_candidates_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                       segCnt, segPos, numSegments, cap, nlanes, out, maxOut, counter, overflow);
    return hipGetLastError();
}

// ---------------------------------------------------------------- the merge

// Joins the per-segment records of a slot: global best, number of columns
// attaining it, and the first capFinal positions in ascending order
// (reference semantics of positions_, edlib.cpp:662-671).  flags bit0 = the
// list is incomplete (a segment or the final list overflowed): the host runs
// the exact second pass for that slot.
__global__ void __launch_bounds__(256)
merge_segments_kernel(const int* __restrict__ segBest, const int* __restrict__ segCnt,
                      const int* __restrict__ segPos, int S, int cap, int nlanes,
                      const int* __restrict__ slotmap, int capFinal,
                      int* __restrict__ best, int* __restrict__ total, int* __restrict__ pos,
                      int* __restrict__ flags)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // lane index of the scan launch
    if (idx >= nlanes) return;
    const int slot = slotmap ? slotmap[idx] : idx;
    const size_t base = (size_t)idx * S;
    int b = 0x7fffffff;
    for (int s = 0; s < S; ++s)
        if (segCnt[base + s] > 0 && segBest[base + s] < b) b = segBest[base + s];
    int n = 0, ovf = 0;
    if (b != 0x7fffffff) {
        for (int s = 0; s < S; ++s) {
            const int c = segCnt[base + s];
            if (c <= 0 || segBest[base + s] != b) continue;
            if (c > cap) ovf = 1;
            const int take = c < cap ? c : cap;
            for (int i = 0; i < take; ++i)
                if (n + i < capFinal) pos[(size_t)slot * capFinal + n + i] = segPos[(base + s) * cap + i];
            n += c;
        }
        if (n > capFinal) ovf = 1;
    }
    best[slot] = (b == 0x7fffffff) ? -1 : b;
    total[slot] = n;
    flags[slot] = ovf;
}

// The same merge with a wave per lane of the scan launch, its lanes striding over the segments: small batches cut the target
// into hundreds of segments (851 leftovers of a 16,384-read batch x 1,216 segments: 0.80 ms with a thread per slot walking
// its 1,216 entries one dependent load after the other).
__global__ void __launch_bounds__(64)
merge_segments_wave_kernel(const int* __restrict__ segBest, const int* __restrict__ segCnt,
                           const int* __restrict__ segPos, int S, int cap, int nlanes,
                           const int* __restrict__ slotmap, int capFinal,
                           int* __restrict__ best, int* __restrict__ total, int* __restrict__ pos,
                           int* __restrict__ flags)
{
    const int idx = blockIdx.x, lane = threadIdx.x;
    const int slot = slotmap ? slotmap[idx] : idx;
    const size_t base = (size_t)idx * S;
    int b = 0x7fffffff;
    for (int s = lane; s < S; s += 64)
        if (segCnt[base + s] > 0 && segBest[base + s] < b) b = segBest[base + s];
    for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(b, off); b = o < b ? o : b; }
    int n = 0, ovf = 0;
    if (b != 0x7fffffff) {
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            int c = s < S ? segCnt[base + s] : 0;
            if (c <= 0 || segBest[base + s] != b) c = 0;
            if (c > cap) ovf = 1;
            int incl = c;                                   // inclusive prefix sum over the wave: where this segment's hits go
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            const int at = n + incl - c, take = c < cap ? c : cap;
            for (int i = 0; i < take; ++i)
                if (at + i < capFinal) pos[(size_t)slot * capFinal + at + i] = segPos[(base + s) * cap + i];
            n += __shfl(incl, 63);
        }
        ovf = __any(ovf) ? 1 : 0;
        if (n > capFinal) ovf = 1;
    }
    if (lane == 0) {
        best[slot] = (b == 0x7fffffff) ? -1 : b;
        total[slot] = n;
        flags[slot] = ovf;
    }
}

hipError_t launch_merge_segments(const int* segBest, const int* segCnt, const int* segPos, int S,
                                 int cap, int nlanes, const int* slotmap, int capFinal, int* best,
                                 int* total, int* pos, int* flags, hipStream_t stream)
{
    if (nlanes == 0) return hipSuccess;
    if (S >= 64) {
        hipLaunchKernelGGL(merge_segments_wave_kernel, dim3(nlanes), dim3(64), 0, stream,
                           segBest, segCnt, segPos, S, cap, nlanes, slotmap, capFinal, best, total, pos, flags);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(merge_segments_kernel, dim3((nlanes + 255) / 256), dim3(256), 0, stream,
                       segBest, segCnt, segPos, S, cap, nlanes, slotmap, capFinal, best, total, pos, flags);
    return hipGetLastError();
}

}  // namespace edlib_amd
