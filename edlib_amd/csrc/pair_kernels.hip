// pair_kernels.hip -- gfx950 kernels for independent (query, target) units of any
// length, alphabet and mode: the general engine behind edlibAlign().
//
// Replaces myersCalcEditDistanceNW / myersCalcEditDistanceSemiGlobal
// (edlib.cpp:730-928, 550-704), buildPeq (:358-384), calculateBlock (:412-447),
// the AlignmentData column store (:22-47, 883-893) and obtainAlignmentTraceback
// (:942-1141).  Design:
//
//   * one wave64 owns one unit.  Lane l holds one 64-row block of the query
//     column (Pv, Mv as a 64-bit pair in VGPRs).  The only dependency inside a
//     column is the horizontal delta hout -> hin between vertically adjacent
//     blocks (edlib.cpp:781-785), so the wave runs the anti-diagonal schedule:
//     at step t lane l works on column t-l, and {hout, target symbol} move one
//     lane down per step in ONE v_mov_b32_dpp wave_shr:1 (lane 0 is fed the
//     next target symbol and the row -1 boundary by the DPP `old` operand).
//   * queries of more than 64 blocks are processed in strips of 64 blocks; the
//     bottom row of a strip hands its horizontal deltas to the next strip
//     through a per-unit buffer in HBM (written by lane 63, read back 64
//     columns at a time, coalesced).
//   * Peq rows live in LDS as [symbol][lane] so that a lane's 8-byte read hits a
//     bank pair that depends on the lane only: conflict-free whatever symbols
//     the 64 lanes are looking at.  The slice is refilled per strip from the
//     HBM Peq pool with coalesced loads; the target is read 64 symbols per
//     coalesced load and handed to lane 0 with v_readlane.
//   * this strip kernel computes every block of every column (all outputs are functions of the full
//     DP matrix); the banded lane-ring kernel further down (scan_pairs_ring_kernel)
//     is what NW pairs, PATH leaves and short semi-global units actually run on, the strips take
//     what no ring holds.
//   * PATH: each block-step also stores (Pv, Mv, block score) in anti-diagonal
//     order ([step][lane], one coalesced line per step); a second kernel walks
//     back with the reference's candidate order up > left > diagonal.
#include "pair_kernels.hpp"
#include "block64.hpp"
#include "lds_check.hpp"

namespace edlib_amd {

typedef unsigned long long u64;
typedef uint32_t u32;

__host__ __device__ static inline int num_blocks(int m) { return (m + 63) >> 6; }

// Column store geometry: strip s (64 blocks, the last one maybe fewer) runs
// T + nbS - 1 steps and writes nbS entries per step.
__host__ __device__ static inline long long strip_base(int strip, int T) {
    return (long long)strip * ((long long)T + 63) * 64;
}
__host__ __device__ static inline long long store_index(int T, int nb, int c, int b) {
    const int strip = b >> 6, l = b & 63;
    const int nbS = (nb - strip * 64) < 64 ? (nb - strip * 64) : 64;
    return strip_base(strip, T) + (long long)(c + l) * nbS + l;
}
long long pair_store_entries(int qlen, int tlen) {
    const int nb = num_blocks(qlen);
    const int full = nb / 64, restBlocks = nb - full * 64;
    long long n = strip_base(full, tlen);
    if (restBlocks) n += ((long long)tlen + restBlocks - 1) * restBlocks;
    return n;
}

// ------------------------------------------------------------------ buildPeq

// The equality relation of 32 target symbols at a time is folded into a 256-entry LDS table (query byte -> bit s set
// iff it equals symbol s).  A wave then builds 64 rows at once: the lanes load 64 consecutive query bytes (one
// coalesced line), look their symbol sets up, and a ballot per symbol IS the 64-bit Peq word of that block; lane j of
// the wave keeps the words of block j of a tile of 64 blocks, so the tile goes out as whole lines per symbol row.
// (Round 1 / 2 gave every thread a block: 64 dependent byte loads each from its own line, and a 256-thread workgroup
// per unit of which three threads had work when the units were 150-base reads.)
// bit r of Peq[sym][b] = eq8[query[64b+r]][byte of sym]; rows past the query end are 0
// (the kernels follow row m-1 explicitly, so the reference's wildcard padding,
// edlib.cpp:373-375, is not needed).
// perWave = 1: a wave per unit (many short units); 0: a workgroup per unit, its four waves taking tiles in turn.
__global__ void __launch_bounds__(256)
build_peq_pairs_kernel(const PairDesc* __restrict__ descs, int numUnits, int perWave, const uint8_t* __restrict__ qpool,
                       const uint8_t* __restrict__ eq8, const uint8_t* __restrict__ idToByte,
                       int sigmaT, u64* __restrict__ peq)
{
    __shared__ u32 s_mask[256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int u0 = perWave ? blockIdx.x * 4 + wv : blockIdx.x, ustep = perWave ? gridDim.x * 4 : gridDim.x;
    const int tile0 = perWave ? 0 : wv * 64, tstep = perWave ? 64 : 256;
    for (int g0 = 0; g0 < sigmaT; g0 += 32) {
        const int ns = (sigmaT - g0) < 32 ? (sigmaT - g0) : 32;
        __syncthreads();
        {
            u32 mk = 0;
            for (int sy = 0; sy < ns; ++sy) {
                const u32 tb = idToByte[g0 + sy];
                mk |= (u32)(eq8 ? (eq8[threadIdx.x * 256 + tb] & 1) : (tb == threadIdx.x)) << sy;      // no matrix: identity
            }
            s_mask[threadIdx.x] = mk;
        }
        __syncthreads();
        for (int u = u0; u < numUnits; u += ustep) {                       // wave-uniform
            const long long qoff = descs[u].qoff, peqOff = descs[u].peqOff;
            const int qlen = descs[u].qlen, qstep = descs[u].qstep;
            const int nb = num_blocks(qlen);
            for (int t0 = tile0; t0 < nb; t0 += tstep) {
                const int tn = (nb - t0) < 64 ? (nb - t0) : 64;
                for (int c0 = 0; c0 < ns; c0 += 4) {                       // four symbols per trip over the tile's bytes
                    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                    // four blocks per trip: their four loads are in flight together (one load per trip left a wave
                    // waiting out a full memory latency per block: 1.1 ms for the 500 MB of config 4's Peq; four per trip:
                    // 0.75 ms; sixteen per trip, round 5: 0.72 -- the rest is the ballots and the stores)
                    constexpr int TR = 4;
                    for (int j0 = 0; j0 < tn; j0 += TR) {
                        u32 by[TR];
#pragma unroll
                        for (int q = 0; q < TR; ++q) {
                            const int row = (t0 + j0 + q) * 64 + lane;
                            by[q] = (j0 + q < tn && row < qlen) ? (u32)qpool[qoff + (long long)row * qstep] : 0xffffffffu;
                        }
#pragma unroll
                        for (int q = 0; q < TR; ++q) {
                            if (j0 + q >= tn) break;                       // (wave-uniform)
                            const u32 mk = by[q] != 0xffffffffu ? s_mask[by[q]] >> c0 : 0u;
                            const u64 b0 = __builtin_amdgcn_ballot_w64((mk & 1u) != 0), b1 = __builtin_amdgcn_ballot_w64((mk & 2u) != 0);
                            const u64 b2 = __builtin_amdgcn_ballot_w64((mk & 4u) != 0), b3 = __builtin_amdgcn_ballot_w64((mk & 8u) != 0);
                            const bool mine = lane == j0 + q;
                            w0 = mine ? b0 : w0; w1 = mine ? b1 : w1; w2 = mine ? b2 : w2; w3 = mine ? b3 : w3;
                        }
                    }
                    if (lane < tn) {
                        u64* out = peq + peqOff + (long long)(g0 + c0) * nb + t0 + lane;
                        out[0] = w0;
                        if (c0 + 1 < ns) out[nb] = w1;
                        if (c0 + 2 < ns) out[2LL * nb] = w2;
                        if (c0 + 3 < ns) out[3LL * nb] = w3;
                    }
                }
            }
        }
    }
}

hipError_t launch_build_peq_pairs(const PairDesc* descs, int numUnits, const uint8_t* qpool,
                                  const uint8_t* eq8, const uint8_t* idToByte, int sigmaT,
                                  u64* peq, hipStream_t stream)
{
    if (numUnits == 0) return hipSuccess;
    // a handful of (possibly very long) units: a workgroup each; many units: a wave each, enough workgroups to
    // fill the chip several times
    const int perWave = numUnits >= 2048 ? 1 : 0;
    const int wgs = perWave ? (numUnits + 3) / 4 : numUnits;
    hipLaunchKernelGGL(build_peq_pairs_kernel, dim3(wgs < 16384 ? wgs : 16384), dim3(256), 0, stream,
                       descs, numUnits, perWave, qpool, eq8, idToByte, sigmaT, peq);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
fill_level_descs_kernel(const LevelSpec* __restrict__ specs, int numUnits, int kcap, int ringBlocks, int cap, int ring,
                        PairDesc* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numUnits) return;
    const LevelSpec v = specs[i];
    PairDesc d;
    d.qoff = v.qoff; d.toff = v.toff; d.peqOff = v.peqOff; d.storeOff = 0; d.auxOff = 0;
    d.qlen = v.qlen; d.tlen = v.tlen; d.qstep = 1; d.tstep = 1;
    const int whole = v.qlen > v.tlen ? v.qlen : v.tlen;
    const int k = num_blocks(v.qlen) <= ringBlocks ? whole : cap;
    d.kinit = k < kcap ? k : kcap;
    d.posCap = 0; d.posOff = 0; d.colOff = -1; d.bandT = 0; d.skip = 0; d.ring = ring;
    out[i] = d;
}

hipError_t launch_fill_level_descs(const LevelSpec* specs, int numUnits, int kcap, int ringBlocks, int cap, int ring,
                                   PairDesc* out, hipStream_t stream)
{
    if (numUnits == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_level_descs_kernel, dim3((numUnits + 255) / 256), dim3(256), 0, stream,
                       specs, numUnits, kcap, ringBlocks, cap, ring, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------- the scan

// MODE 0 NW, 1 SHW, 2 HW.  STORE: keep the column store.  LDSPEQ: Peq slice in LDS
// (sigmaT <= 32), else gathered from the HBM pool.
//
// Pipeline of one strip.  At step t lane l updates column t-l.  The packed word that travels one
// lane down per step (one v_mov_b32_dpp wave_shr:1) carries {hout of the sender's column, the
// sender's NEXT symbol}: the symbol stream runs one step ahead of the DP, so every lane knows the
// symbol of its next column one step early and fetches that Peq word (LDS or HBM) while it computes
// the current column.  Lane 0 is fed {row -1 delta of column t, target[t+1]} through the DPP `old`
// operand.  The loop starts at t = -1 (symbols only).
template <int MODE, bool STORE, bool LDSPEQ>
__global__ void __launch_bounds__(64)
scan_pairs_kernel(const PairScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u64 s_peq[];     // [sigmaT][64]
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const PairDesc d = a.descs[unit];
    const int m = d.qlen, T = d.tlen;
    const int nb = num_blocks(m);
    const int nstrips = (nb + 63) >> 6;
    const u32 sh = (u32)(m - 1) & 63u;                               // row m-1 inside the last block
    const int topCode = (MODE == 2) ? 0 : 1;                         // {hneg,hpos} bits at row -1: HW 0, SHW/NW +1

    int sc = m;                                                      // D[m][-1] (edlib.cpp:575-579)
    int best = d.kinit, cnt = 0, lastCol = -1;
    int* pos = a.posPool + d.posOff;
    const bool dumpCol = a.colP != nullptr && d.colOff >= 0;

    for (int strip = 0; strip < nstrips; ++strip) {
        const int nbS = (nb - strip * 64) < 64 ? (nb - strip * 64) : 64;
        const int b = strip * 64 + lane;
        const bool laneOn = lane < nbS;
        const bool lastStrip = strip == nstrips - 1;
        const bool tracker = lastStrip && (b == nb - 1);             // lane that owns row m-1

        if (LDSPEQ) {
            __syncthreads();                                         // previous strip done with s_peq
            for (int sy = 0; sy < a.sigmaT; ++sy)
                s_peq[sy * 64 + lane] = laneOn ? a.peq[d.peqOff + (long long)sy * nb + b] : 0ull;
            __syncthreads();
        }
        Block64 B{~0u, ~0u, 0u, 0u};                                 // column -1
        int bscore = (b + 1) * 64;                                   // block bottom score (edlib.cpp:576)
        int carry = 0;
        int tchunk = 0, hchunk = topCode;
        const long long sbase = STORE ? d.storeOff + strip_base(strip, T) : 0;
        const int nsteps = T + nbS - 1;
        const unsigned long long* peqRow = a.peq + d.peqOff + b;     // HBM Peq of this lane's block

        auto step = [&](const int t, const u64 eqCur, u64& eqNxt) {
            const int tc = t + 1;                                    // column injected at lane 0 now
            if ((tc & 63) == 0) {                                    // refill 64 columns of symbols
                const int c = tc + lane;
                tchunk = (c < T) ? a.tlut[a.tpool[d.toff + (long long)c * d.tstep]] : 0;
            }
            if (strip > 0 && t >= 0 && (t & 63) == 0) {              // and of the previous strip's bottom deltas
                const int c = t + lane;
                hchunk = (c < T) ? __hip_atomic_load(&a.aux[d.auxOff + c], __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
            const int in0 = (__builtin_amdgcn_readlane(tchunk, tc & 63) << 2)
                          | __builtin_amdgcn_readlane(hchunk, t & 63);
            const int x = __builtin_amdgcn_update_dpp(in0, carry, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
            const u32 symN = (u32)x >> 2;
            eqNxt = LDSPEQ ? s_peq[symN * 64 + lane] : (laneOn ? peqRow[(long long)symN * nb] : 0ull);
            const int col = t - lane;
            u32 hp = 0, hn = 0;
            if (laneOn && col >= 0 && col < T) {
                u32 ph0, ph1, mh0, mh1, xh0, xh1;
                advance_block64(B, (u32)eqCur, (u32)(eqCur >> 32), (u32)x & 1u, ((u32)x >> 1) & 1u, ph0, ph1, mh0, mh1, xh0, xh1);
                hp = ph1 >> 31; hn = mh1 >> 31;
                bscore += (int)hp - (int)hn;
                if (STORE) {
                    const long long si = sbase + (long long)t * nbS + lane;
                    StoreEntry e;
                    store_planes(B, ph0, ph1, xh0, xh1, e.x, e.y);
                    a.store[si] = e;
                }
                if (!lastStrip) {
                    if (lane == 63) a.aux[d.auxOff + col] = (int)(hp | (hn << 1));
                } else {
                    if (tracker) {
                        const u64 ph = ((u64)ph1 << 32) | ph0, mh = ((u64)mh1 << 32) | mh0;
                        sc += (int)((ph >> sh) & 1ull) - (int)((mh >> sh) & 1ull);
                        if (MODE != 0 && sc <= best && col >= d.skip) {  // edlib.cpp:658-673
                            if (sc < best) { best = sc; cnt = 0; }
                            if (cnt < d.posCap) pos[cnt] = col;
                            ++cnt;
                            lastCol = col;
                        }
                    }
                }
                if (dumpCol && col == T - 1) {                           // last column (Hirschberg)
                    a.colP[d.colOff + b] = ((u64)B.p1 << 32) | B.p0; a.colM[d.colOff + b] = ((u64)B.m1 << 32) | B.m0;
                    a.colS[d.colOff + b] = bscore;
                }
            }
            carry = (int)((symN << 2) | hp | (hn << 1));             // every lane forwards the symbol stream
        };
        u64 eqA = 0, eqB = 0;
        for (int t = -1; t < nsteps; t += 2) {                       // two steps per trip: Peq registers ping-pong
            step(t, eqA, eqB);
            if (t + 1 < nsteps) step(t + 1, eqB, eqA);
        }
        if (tracker) {
            a.outScore[unit] = (MODE == 0) ? sc : (cnt > 0 ? best : -1);
            a.outCount[unit] = (MODE == 0) ? 1 : cnt;
            a.outLast[unit] = (MODE == 0) ? T - 1 : lastCol;
        }
        if (!lastStrip) __threadfence();                             // hand-off buffer visible to our own reloads
    }
}

template <int MODE, bool STORE>
static hipError_t launch_scan_pairs_t(const PairScanArgs& a, hipStream_t stream)
{
    if (a.sigmaT <= 32) {
        const size_t lds = (size_t)a.sigmaT * 64 * sizeof(u64);
        hipLaunchKernelGGL((scan_pairs_kernel<MODE, STORE, true>), dim3(a.numUnits), dim3(64), lds, stream, a);
    } else {
        hipLaunchKernelGGL((scan_pairs_kernel<MODE, STORE, false>), dim3(a.numUnits), dim3(64), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_scan_pairs(int mode, bool store, const PairScanArgs& a, hipStream_t stream)
{
    if (a.numUnits == 0) return hipSuccess;
    switch (mode * 2 + (store ? 1 : 0)) {
        case 0: return launch_scan_pairs_t<0, false>(a, stream);
        case 1: return launch_scan_pairs_t<0, true>(a, stream);
        case 2: return launch_scan_pairs_t<1, false>(a, stream);
        case 3: return launch_scan_pairs_t<1, true>(a, stream);
        case 4: return launch_scan_pairs_t<2, false>(a, stream);
        case 5: return launch_scan_pairs_t<2, true>(a, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------- banded NW scan

// NW inside Ukkonen's diagonal band for a fixed threshold K = desc.kinit (what the reference's
// first/lastBlock bookkeeping converges to, edlib.cpp:744-830): a path of cost <= K only visits
// diagonals d = j - i in [dmin, dmax] = [min(0,D) - p, max(0,D) + p], D = T - m, p = (K - |D|) / 2.
// Block b therefore lives for columns [64b + dmin, 64b + 63 + dmax] -- steps [65b + dmin, 65b + 63 + dmax] of the
// anti-diagonal schedule below -- and at most G consecutive blocks are at work in a step when K <= ring_max_k(G) =
// 65 G - 64 (rounds 1-4 kept one lane idle between the bottom and the top of the band: 64 (G - 2); now a block stops
// listening to the lane above when the block above leaves the band).  A RING of G lanes (G = 4, 8, 16, 21, 32 or 64)
// therefore covers a query of any length, and a wave carries 64 / G independent units:
//   * blocks are mapped to the ring's lanes round-robin (block b -> ring lane b % G).  When a lane's
//     block leaves the band it re-arms for block b + G: state "+1 per row" below the upstream block's
//     bottom score, exactly the reference's new block (edlib.cpp:803-808); cells outside the band only
//     ever enter as such upper bounds, so values <= K stay exact (Ukkonen);
//   * same anti-diagonal schedule as scan_pairs_kernel (block b updates column t - b at step t); the
//     carry moves one lane up the ring per step -- one DPP move for rings of 4, 16 and 64 lanes
//     (quad_perm / row_ror / wave_ror), ds_bpermute_b32 for 8, 21 and 32 (any ring size works: 21 lanes
//     = three units per wave is what 10 kb pairs at ONT-like divergence run on);
//   * THE STEP IS BRANCH-LIGHT (round 2; round 1 spent 56 % of its instructions on glue).  Every lane runs
//     the block update every step, whether its block is alive or not: a lane outside its block's life
//     computes garbage on state that the next (re)arm overwrites.  What keeps that harmless:
//       - the SENDER neutralises its carry: a lane that is not inside its block's life emits hout = +1,
//         which is what a block takes from an upstream outside the band (edlib.cpp:779), so the receiver
//         needs no "is my upstream alive" test;
//       - the only per-step bookkeeping is one countdown per lane (steps to the lane's next event: its block
//         starts, or its block has finished); the wave leaves the straight-line step only when some lane's
//         countdown is at zero (about one step in eight at config 4's shape);
//       - the NW score is decoded from the last block's final state (bottom score minus the vertical deltas
//         below row m-1, like the reference does at edlib.cpp:914-917) instead of being followed per column;
//   * target symbols sit in a 256-entry LDS ring per unit as the BYTE OFFSET of their Peq row (one v_add
//     away from the address of the lane's Peq word), filled 64 columns at a time; each lane reads the
//     offset of its next-but-one column and the Peq word of its next column while it computes the current one;
//   * steps: T + numBlocks - 1 instead of (T + 63) per 64-block strip;
//   * STORE: every block-step also writes (Pv, Mv, block score) at [ring lane][column] for
//     traceback_kernel -- the reference's AlignmentData restricted to first..lastBlock (edlib.cpp:883-893).
// A unit whose blocks all fit the ring (numBlocks <= G) may use any K: with K = max(m, T) the band is
// the whole matrix.
template <int G> __device__ __forceinline__ int ring_ror(const int v, const int srcAddr)
{
    // every lane has a source lane: no `old` operand to initialise (v_mov_b32_dpp ... bound_ctrl:1)
    if constexpr (G == 64) return __builtin_amdgcn_mov_dpp(v, 0x13C /*wave_ror:1*/, 0xf, 0xf, true);
    else if constexpr (G == 16) return __builtin_amdgcn_mov_dpp(v, 0x121 /*row_ror:1*/, 0xf, 0xf, true);
    else if constexpr (G == 4) return __builtin_amdgcn_mov_dpp(v, 0x93 /*quad_perm:[3,0,1,2]*/, 0xf, 0xf, true);
    // (rings that do not divide a DPP row -- 8 is packed two to a row, 21 and 32 cross rows: a whole-wave rotation plus
    // v_readlane / v_writelane fix-ups of the rings' first lanes was measured in round 3: config 4's scans 19.6 -> 22.6 ms,
    // the SGPR round trips stall the VALU longer than the LDS round trip they replace)
    else return __builtin_amdgcn_ds_bpermute(srcAddr, v);              // srcAddr = 4 * (lane of the ring's previous lane)
}

// Ring layout of the column store: step-major, G entries per step, block b in slot b % G (blocks b and b + G are never
// alive in the same column); block b meets column c at step c + b.  The lanes of a unit write G consecutive entries
// per step (round 3, first layout: one row of T entries per ring lane -- every lane of the wave stored to a line of
// its own, and the storing scan of config 5 took 0.55 ms against 0.24 ms without the store: the address coalescer
// handles a line per cycle); the traceback walking left through a block row reads with stride G.
__host__ __device__ static inline long long ring_index(int G, int c, int b) {
    return (long long)(c + b) * G + (b % G);
}
long long ring_store_entries(int G, int qlen, int tlen) {
    return ((long long)tlen + num_blocks(qlen)) * G;
}

// PEQ: where a lane finds the Peq word of (symbol, its block):
//   0  HBM pool (more than 32 target symbols)
//   1  LDS slice [symbol][lane], refilled from the pool whenever the lane re-arms for a new block
//   2  the unit's whole Peq in LDS ([unit in wave][symbol][block], a.peqFullStride words per unit, rows
//      of a.peqRowStride words):
//      no refills, chosen by the launcher for packed rings when a wave needs at most 8 KB
// MODE 0 NW as described; MODE 1 SHW / 2 HW: no band (the launcher only sends units whose blocks all fit
// the ring), kinit is the end-location threshold as in scan_pairs_kernel, the lane of the last block
// follows row m-1 and records best / count / positions (edlib.cpp:658-673).
// H: 64-row blocks per ring lane ("superblock", round 3).  A lane holds H vertically adjacent blocks of the same column
// and updates them one after the other inside a step (the carry between them never leaves the lane), so everything a
// step does once per lane -- the carry exchange with the ring neighbour, the countdown, the fetch of the next column's
// row offset and Peq address, the neutral-carry selects -- is paid once per H blocks: ~20 of the 42 VALU instructions
// of the H = 1 step are such glue (profiles/README.md), against 22 per block update.  A ring of G lanes then holds
// bands of up to 64 H (G - 2) rows: 16 lanes x 2 blocks cover what 21- and 32-lane rings did, with four units per wave
// instead of three or two and the carry on one DPP row rotation instead of ds_bpermute.  Band, block life and the "+1 per
// row" start are the formulas above with 64 H rows per block.  H > 1: distance scans only (no column store).
template <int G, int MODE, bool STORE, int PEQ, int H>
__global__ void __launch_bounds__(64)
scan_pairs_ring_kernel(const PairScanArgs a)
{
    static_assert(H == 1 || !STORE, "the column store is laid out per 64-row block");
    // A ring scan is a chain of dependent steps: next to a bandwidth kernel on another stream (the alphabet count, the Peq
    // build of the whole batch beside the 16-wave divergence probe) its waves issue first.
    __builtin_amdgcn_s_setprio(2);
    extern __shared__ __attribute__((aligned(16))) u64 s_dyn[];      // Peq words, then the target rings
    constexpr int U = 64 / G;                                         // units per wave (lanes past U * G idle)
    constexpr int RH = 64 * H;                                        // rows per ring lane
    const int lane = threadIdx.x;
    const int rl = lane % G, uwRaw = lane / G;                        // lane within the ring, ring within the wave
    const bool inRing = uwRaw < U;
    const int uw = inRing ? uwRaw : U - 1;                            // idle lanes alias the last ring's LDS (reads only)
    const int srcAddr = 4 * (lane - rl + (rl == 0 ? G - 1 : rl - 1));
    const int peqWords = PEQ == 1 ? a.sigmaT * 64 * H : (PEQ == 2 ? ((U * a.peqFullStride + 63) & ~63) : 0);   // (a multiple of 512 bytes)
    u64* s_peq = s_dyn + (PEQ == 2 ? uw * a.peqFullStride : 0);
    unsigned short* s_tgt = reinterpret_cast<unsigned short*>(s_dyn + peqWords) + uw * 256;
    // LDS byte address of the unit's target ring: 512-byte aligned, so that slot (c & 255) is tbase | (2 c & 511) -- one
    // v_and_or_b32 from a counter that advances by 2 per step
    const u32 tbase = (u32)(size_t)(__attribute__((address_space(3))) unsigned short*)s_tgt;
    // (the rings start 512-byte aligned: no static LDS in front of the dynamic block -- checked on the host, lds_check.hpp)
    auto tgt_at = [&](const u32 c2) -> int {
        return *(const __attribute__((address_space(3))) unsigned short*)(size_t)(tbase | (c2 & 511u));
    };
    const int unit = blockIdx.x * U + uwRaw;
    const bool have = inRing && unit < a.numUnits;
    const PairDesc* dp = a.descs + (have ? unit : 0);
    // a whole-wave ring keeps its descriptor in SGPRs
    auto uni = [](int v) { return G == 64 ? __builtin_amdgcn_readfirstlane(v) : v; };
    auto uni64 = [&](long long v) {
        return G == 64 ? (long long)(((u64)(u32)uni((int)((u64)v >> 32)) << 32) | (u32)uni((int)(u32)(u64)v)) : v;
    };
    const int m = uni(dp->qlen), T = uni(dp->tlen), K = uni(dp->kinit);   // T: columns to process (the scan stops at column T-1)
    const int bandT = uni(dp->bandT);
    // rarely used descriptor fields: SGPRs for a whole-wave ring, re-read from the descriptor (L1/L2) by the
    // packed rings, whose VGPR budget decides the occupancy
    const int tstepU = G == 64 ? uni(dp->tstep) : 0;
    const long long toffU = G == 64 ? uni64(dp->toff) : 0, peqOffU = G == 64 ? uni64(dp->peqOff) : 0;
    const long long colOffU = G == 64 ? uni64(dp->colOff) : 0;
    auto tstep_ = [&]() { return G == 64 ? tstepU : dp->tstep; };
    auto toff_ = [&]() { return G == 64 ? toffU : dp->toff; };
    auto peqOff_ = [&]() { return G == 64 ? peqOffU : dp->peqOff; };
    const long long storeOff = STORE ? uni64(dp->storeOff) : 0;
    const int nb = num_blocks(m);                                     // 64-row blocks
    const int nsb = (nb + H - 1) / H;                                 // ring-lane blocks of RH rows
    // MODE 1 with bandT < 0: SHW inside the band of threshold K -- a cell (i, j) with D <= K has |i - j| <= K, so the band
    // is the diagonals [-K, K] whatever the lengths are (the reference's SHW band, edlib.cpp:562, 602-630, for a fixed k);
    // the host cuts the target at column m + K and answers "none" itself when T < m - K
    const bool shwBand = MODE == 1 && bandT < 0;
    // MODE 2 with bandT < 0: HW inside the band of threshold K -- an alignment of the whole query with cost <= K starts at a
    // column j0 in [0, T - m + K] and stays within K diagonals of j0: the diagonals [-K, (T - m) + 2 K].  What the
    // reference's first..lastBlock bookkeeping (edlib.cpp:562, 602-630) saves on a query in a window not much longer than
    // itself, as a static band: the smallest ring that holds (T - m) + 3 K rows instead of the whole query.
    const bool hwBand = MODE == 2 && bandT < 0;
    const int D = (shwBand || hwBand) ? 0 : (bandT > 0 ? bandT : T) - m, absD = D < 0 ? -D : D;     // the band is that of the whole problem
    // last-column dump of Hirschberg halves: the packed rings look their slot up when they get there
    const bool dumpCol = a.colP != nullptr && (G != 64 || colOffU >= 0);
    const bool active = have && (MODE != 0 || K >= absD);
    if (have && !active && rl == 0) { a.outScore[unit] = 0x3fffffff; a.outCount[unit] = 0; a.outLast[unit] = -1; }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;
    const int p = MODE != 0 ? ((shwBand || hwBand) ? K : (1 << 28)) : (K - absD) >> 1;            // semi-global: the whole matrix unless banded
    const int dmin = (D < 0 ? D : 0) - p, dmax = hwBand ? (T > m ? T - m : 0) + 2 * K : (D > 0 ? D : 0) + p;
    int best = K, cnt = 0, lastCol = -1;                              // MODE != 0: columns scoring <= best qualify
    const u32 sh = (u32)(m - 1) & 63u;                                // row m-1 inside its 64-row block ...
    const int hb = (nb - 1) % H;                                      // ... which is this block of the last ring-lane block
    const int lastRows = m - RH * (nsb - 1);                          // query rows in the last ring-lane block
    const int nbA = active ? nsb : 0;                                 // idle rings own no block

    // ---- target ring: the Peq row offsets of columns [0, loaded) are in s_tgt[col & 255]
    // symbol -> byte offset of its Peq row as seen from the lane's base address (PEQ 0: the symbol itself)
    const int rowStride = PEQ == 2 ? a.peqRowStride : 0;
    const int symScale = PEQ == 1 ? 512 * H : (PEQ == 2 ? 8 * rowStride : 1);
    int loaded = 0;
    auto refill = [&]() {                                             // 64 more columns per ring
        const long long toff = toff_(); const int tstep = tstep_();
        for (int i = rl; i < 64; i += G) {
            const int c = loaded + i;
            s_tgt[c & 255] = (unsigned short)((c < T) ? a.tlut[a.tpool[toff + (long long)c * tstep]] * symScale : 0);
        }
        loaded += 64;
    };
    if (inRing) for (int i = rl; i < 256; i += G) s_tgt[i] = 0;       // never index Peq with an unwritten slot
    if (active) { refill(); refill(); refill(); }                     // 192 columns ahead of column 0
    // rows padded to a.peqRowStride words (a multiple of 32 for long queries): a ring's lanes, which hold
    // consecutive blocks, then hit distinct bank pairs whatever symbols they look up
    if (PEQ == 2 && active) {
        const long long peqOff = peqOff_();
        for (int sy = 0; sy < a.sigmaT; ++sy)
            for (int i = rl; i < rowStride; i += G) s_peq[sy * rowStride + i] = i < nb ? a.peq[peqOff + (long long)sy * nb + i] : 0ull;
    }
    int b = rl;                                                       // current (or next) ring-lane block of this lane
    // Peq words of the column whose row offset is `off` (the H blocks of this lane): one add from the lane's base
    const char* peqBase = reinterpret_cast<const char*>(s_peq) + (PEQ == 1 ? 8 * H * lane : 0);
    auto peq_words = [&](const int off, u64 (&w)[H]) {
        if (PEQ == 2) {
            const u64* q = reinterpret_cast<const u64*>(peqBase + off + 8 * H * b);
#pragma unroll
            for (int h = 0; h < H; ++h) w[h] = q[h];
        } else if (PEQ == 1) {
            const u64* q = reinterpret_cast<const u64*>(peqBase + off);
#pragma unroll
            for (int h = 0; h < H; ++h) w[h] = q[h];
        } else {
#pragma unroll
            for (int h = 0; h < H; ++h) w[h] = (b * H + h < nb) ? a.peq[peqOff_() + (long long)off * nb + b * H + h] : 0ull;
        }
    };

    // ---- per-lane block bookkeeping.  Block b is updated at steps tstart .. tstart + span
    auto first_col = [&](int blk) { const int c = RH * blk + dmin; return c < 0 ? 0 : c; };
    auto last_col = [&](int blk) { const int c = RH * blk + RH - 1 + dmax; return c > T - 1 ? T - 1 : c; };
    const int never = 0x3fffffff;
    int ev, span = 0;                                                 // steps until this lane's next event; life of its block
    auto arm = [&](const int t) {                                     // countdown to the start of block b
        const bool ok = b < nbA && first_col(b) <= last_col(b);       // blocks below the band never get a column
        const int tstart = ok ? first_col(b) + b : never;
        span = ok ? last_col(b) + b - tstart : 0;
        ev = ok ? tstart - t : never;
    };
    arm(0);
    u32 actm = 0u;                                                    // all ones while the lane is inside its block's life
    u32 xmask = (b == 0) ? 0u : ~0u, xfix = (b == 0) ? ((MODE == 2) ? 0u : 1u) : 0u;   // what block 0 takes instead of x

    Block64 B[H];
#pragma unroll
    for (int h = 0; h < H; ++h) B[h] = Block64{~0u, ~0u, 0u, 0u};
    // carry word: bit 0 = hout is +1, bit 16 = hout is -1.  The block's bottom score is bscore + (low half of csum) - (high
    // half): csum simply adds up the carries the lane sends (one add per step), folded into bscore where it is read
    int bscore = 0, sc = 0, carry = 1;
    u32 csum = 0;
    auto fold = [&]() { bscore += (int)(csum & 0xffffu) - (int)(csum >> 16); csum = 0; };
    int nsteps = active ? T + nsb : 0;                                // one step past the last block's last: its closing event
    if constexpr (G == 64) nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    else {                                          // the wave runs for its longest unit
        int w = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) { const int v = __builtin_amdgcn_readlane(nsteps, u * G); w = v > w ? v : w; }
        nsteps = w;
    }
    const int nstepsU = __builtin_amdgcn_readfirstlane(nsteps);       // (the loop bound in an SGPR)
    u32 c2 = 0;                                                       // twice the column of the NEXT row-offset fetch (col + 2)
    // vertical deltas summed over the blocks below block h of this lane (score of block h's bottom row = bscore - that)
    auto below_blocks = [&](const int h) {
        int d = 0;
#pragma unroll
        for (int q = 1; q < H; ++q)
            if (q > h) d += __popcll(((u64)B[q].p1 << 32) | B[q].p0) - __popcll(((u64)B[q].m1 << 32) | B[q].m0);
        return d;
    };

    // The rare part of a step: some lane's block has just finished (its last update was step t - 1) or starts now.
    auto events = [&](const int t, const int x, u64 (&eqCur)[H], int& offCur) {
        if constexpr (!STORE) fold();
        const int upScore = ring_ror<G>(bscore, srcAddr);             // upstream's bottom score after step t - 1
        // ---- the block above has sent its last carry (taken in step t - 1): from here on block b is the top of the band
        // and takes +1 whatever arrives -- the lane above may already hold its next tenant, block b - 1 + G, when the
        // band fills the ring (K up to ring_max_k(G) = 65 G - 64; with the lane above kept idle instead it was 64 (G - 2))
        // (the block above is last updated at column RH b - 1 + dmax, unless the target ends first; its carry of that step
        // is taken at step (RH + 1) b + dmax - 1)
        if (ev == 0 && actm && xmask != 0u && b > 0 && t == (RH + 1) * b + dmax && RH * b + dmax < T) {
            xmask = 0u; xfix = 1u;
            ev = last_col(b) + b + 1 - t;
        }
        if (ev == 0 && actm) {                                        // ---- closing block b
            const int colLast = t - 1 - b;
            if (colLast == T - 1) {                                   // it was alive at the stop column
                if (b == nsb - 1) {
                    if (MODE == 0) {
                        // D[m][T] from the bottom score of row m-1's block and the vertical deltas below row m-1 (edlib.cpp:914-917)
                        u64 P = 0, M = 0;
#pragma unroll
                        for (int h = 0; h < H; ++h) if (h == hb) { P = ((u64)B[h].p1 << 32) | B[h].p0; M = ((u64)B[h].m1 << 32) | B[h].m0; }
                        const u64 below = (sh == 63u) ? 0ull : (~0ull << (sh + 1));
                        a.outScore[unit] = bscore - below_blocks(hb) - __popcll(P & below) + __popcll(M & below);
                        a.outCount[unit] = 1; a.outLast[unit] = T - 1;
                    } else {
                        a.outScore[unit] = cnt > 0 ? best : -1; a.outCount[unit] = cnt; a.outLast[unit] = lastCol;
                    }
                }
                if (dumpCol) {                                        // stop column of a Hirschberg half
                    const long long co = G == 64 ? colOffU : dp->colOff;
                    if (co >= 0) {
#pragma unroll
                        for (int h = 0; h < H; ++h)
                            if (b * H + h < nb) {
                                a.colP[co + b * H + h] = ((u64)B[h].p1 << 32) | B[h].p0; a.colM[co + b * H + h] = ((u64)B[h].m1 << 32) | B[h].m0;
                                a.colS[co + b * H + h] = bscore - below_blocks(h);
                            }
                    }
                }
            }
            actm = 0u;
            b += G;
            xmask = ~0u; xfix = 0u;
            arm(t);                                                   // ev == 0 again when block b + G starts right now
        }
        if (ev == 0 && !actm) {                                       // ---- block b starts with this step
            const int col = t - b;
            if (PEQ == 1) {
                const long long peqOff = peqOff_();
                for (int sy = 0; sy < a.sigmaT; ++sy)
#pragma unroll
                    for (int h = 0; h < H; ++h)
                        s_peq[(sy * 64 + lane) * H + h] = (b * H + h < nb) ? a.peq[peqOff + (long long)sy * nb + b * H + h] : 0ull;
            }
#pragma unroll
            for (int h = 0; h < H; ++h) B[h] = Block64{~0u, ~0u, 0u, 0u};    // "+1 per row" (edlib.cpp:759-763, 803-808)
            // bottom of the block above at column col - 1: upstream's bottom after its step minus its delta at `col`
            // (a sender outside its block's life extrapolates by +1 per column, the value its receivers assume)
            const int above = (col == 0) ? RH * b : upScore - ((x & 1) - ((x >> 16) & 1));
            bscore = above + RH;
            if (MODE != 0 && b == nsb - 1) sc = above + lastRows;
            peq_words(s_tgt[col & 255], eqCur);
            offCur = s_tgt[(col + 1) & 255];
            c2 = 2u * (u32)(col + 2);
            actm = ~0u;
            ev = span + 1;                                            // closes at the top of the step after its last
            if (b > 0 && RH * b + dmax < T) {                         // ... unless the block above leaves the band first
                const int w = (RH + 1) * b + dmax - t;
                if (w > 0) ev = w;
                else { xmask = 0u; xfix = 1u; }                       // (a band of one diagonal: it already has)
            }
        }
    };

    // One step.  eqCur / offCur: Peq words of this step's column and row offset of the next column (fetched by
    // the previous step); eqNxt / offNxt are fetched here for the next step -- the caller swaps the two
    // register sets every step instead of moving them.
    auto step = [&](const int t, u64 (&eqCur)[H], u64 (&eqNxt)[H], int& offCur, int& offNxt) {
        const int x = ring_ror<G>(carry, srcAddr);
        if (__builtin_amdgcn_ballot_w64(ev == 0) != 0ull) events(t, x, eqCur, offCur);
        --ev;
        peq_words(offCur, eqNxt);
        offNxt = tgt_at(c2);
        c2 += 2u;
        // block 0 takes row -1 (+1 per column, 0 for HW: edlib.cpp:584, 779) whatever its ring neighbour sends (in a
        // ring that holds all blocks of its unit the last block's lane feeds lane 0); every other block takes what
        // arrives: its upstream's delta, or the +1 a sender outside its block's life emits
        const u32 xx = __builtin_amdgcn_bitop3_b32((u32)x, xmask, xfix, 0xea /* (a & b) | c */);
        u32 hp = xx & 1u, hn = xx >> 16;
        u32 phT0 = 0, phT1 = 0, mhT0 = 0, mhT1 = 0;                   // horizontal deltas of row m-1's block (MODE != 0)
        StoreEntry ent{0, 0};
#pragma unroll
        for (int h = 0; h < H; ++h) {                                 // top to bottom: the carry stays in the lane
            u32 ph0, ph1, mh0, mh1, xh0, xh1;
            advance_block64(B[h], (u32)eqCur[h], (u32)(eqCur[h] >> 32), hp, hn, ph0, ph1, mh0, mh1, xh0, xh1);
            hp = ph1 >> 31; hn = mh1 >> 31;
            if constexpr (STORE) store_planes(B[h], ph0, ph1, xh0, xh1, ent.x, ent.y);    // (STORE: H == 1)
            if (MODE != 0) { const bool tr = h == hb; phT0 = tr ? ph0 : phT0; phT1 = tr ? ph1 : phT1; mhT0 = tr ? mh0 : mhT0; mhT1 = tr ? mh1 : mhT1; }
        }
        // carry (and with it the block score) of a lane outside its block's life: +1 per step
        carry = (int)__builtin_amdgcn_bitop3_b32(hp | (hn << 16), 1u, actm, 0xe4 /* c ? a : b */);
        if constexpr (STORE) bscore += (int)((u32)carry & 1u) - (int)((u32)carry >> 16);
        else csum += (u32)carry;
        if (STORE || MODE != 0) {
            if (actm) {
                const int col = t - b;
                if constexpr (STORE) a.store[storeOff + (long long)t * G + rl] = ent;     // ring_index(G, col, b)
                if (MODE != 0 && b == nsb - 1) {
                    const u64 ph = ((u64)phT1 << 32) | phT0, mh = ((u64)mhT1 << 32) | mhT0;
                    sc += (int)((ph >> sh) & 1ull) - (int)((mh >> sh) & 1ull);
                    if (sc <= best && col >= dp->skip) {              // edlib.cpp:658-673
                        if (sc < best) { best = sc; cnt = 0; }
                        if (cnt < dp->posCap) a.posPool[dp->posOff + cnt] = col;
                        ++cnt;
                        lastCol = col;
                    }
                }
            }
        }
    };

    u64 eqA[H], eqB[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { eqA[h] = 0; eqB[h] = 0; }
    int offA = 0, offB = 0;
    for (int t = 0; t <= nstepsU; t += 2) {
        if ((t & 63) == 0) {                                          // pace the ring by the largest column in use
            if constexpr (!STORE) fold();                             // (the halves of csum hold 64 steps with room to spare)
            int bt = t - (RH - 1) - dmax; bt = bt <= 0 ? 0 : (bt + RH) / (RH + 1);
            if (t - T + 1 > bt) bt = t - T + 1;
            const int jmax = t - bt;
            while (active && loaded < T && loaded < jmax + 64 + 67) refill();
        }
        step(t, eqA, eqB, offA, offB);
        step(t + 1, eqB, eqA, offB, offA);
    }
}

// Word-steps inside the band of a ring launch (EdlibAmdBatchStats.word_steps, bench.py's valu_roofline): the block lives of
// scan_pairs_ring_kernel recounted from the band geometry, one thread per unit.  (Round 3 first counted them inside the
// scan; the counter carried through the loop was the 65th and 66th VGPR of the Peq-in-LDS variant -- 7 waves per SIMD
// instead of 8 -- and neither an LDS slot nor a recount after the loop got the allocator below 64.)
__global__ void __launch_bounds__(256)
count_ring_steps_kernel(const PairDesc* __restrict__ descs, const int n, const int mode, const int H, unsigned long long* out)
{
    const int unit = blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = 0;
    if (unit < n) {
        const PairDesc d = descs[unit];
        const int m = d.qlen, T = d.tlen, K = d.kinit, RH = 64 * H;
        const int nsb = (num_blocks(m) + H - 1) / H;
        const bool shwBand = mode == 1 && d.bandT < 0, hwBand = mode == 2 && d.bandT < 0;
        const int D = (shwBand || hwBand) ? 0 : (d.bandT > 0 ? d.bandT : T) - m, absD = D < 0 ? -D : D;
        if (mode != 0 || K >= absD) {
            const int p = mode != 0 ? ((shwBand || hwBand) ? K : (1 << 28)) : (K - absD) >> 1;
            const int dmin = (D < 0 ? D : 0) - p, dmax = hwBand ? (T > m ? T - m : 0) + 2 * K : (D > 0 ? D : 0) + p;
            for (int b = 0; b < nsb; ++b) {
                int f = RH * b + dmin, l = RH * b + RH - 1 + dmax;
                f = f < 0 ? 0 : f; l = l > T - 1 ? T - 1 : l;
                if (f <= l) v += (unsigned long long)(l - f + 1);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v * (2ull * H));
}

hipError_t launch_count_ring_steps(const PairDesc* descs, int numUnits, int mode, int H, unsigned long long* out, hipStream_t stream)
{
    if (numUnits == 0) return hipSuccess;
    hipLaunchKernelGGL(count_ring_steps_kernel, dim3((numUnits + 255) / 256), dim3(256), 0, stream, descs, numUnits, mode, H, out);
    return hipGetLastError();
}

template <int G, int MODE, bool STORE, int H = 1>
static hipError_t launch_scan_pairs_ring_t(const PairScanArgs& a, hipStream_t stream)
{
    constexpr int U = 64 / G;
    const dim3 grid((a.numUnits + U - 1) / U);
    const size_t full = (((size_t)U * a.peqFullStride + 63) & ~(size_t)63) * sizeof(u64);     // (the target rings start 512-byte aligned)
    const size_t tgt = 512 * U;                                       // 256 row offsets (u16) per unit
    // the whole-Peq mode saves the refills of packed rings, but only pays while LDS does not cap the
    // occupancy (measured: 10 KB per wave costs config 4 a third of its rate); row offsets are 16 bits
    if (G < 64 && a.peqFullStride > 0 && full + tgt <= 8192 && 8LL * a.peqRowStride * a.sigmaT < 65536 && a.peqRowStride % H == 0) {
        EDLIB_AMD_CHECK_STATIC_LDS((scan_pairs_ring_kernel<G, MODE, STORE, 2, H>), 0);
        hipLaunchKernelGGL((scan_pairs_ring_kernel<G, MODE, STORE, 2, H>), grid, dim3(64), full + tgt, stream, a);
    } else if (a.sigmaT <= 32) {
        const size_t lds = (size_t)a.sigmaT * 64 * H * sizeof(u64) + tgt;
        EDLIB_AMD_CHECK_STATIC_LDS((scan_pairs_ring_kernel<G, MODE, STORE, 1, H>), 0);
        hipLaunchKernelGGL((scan_pairs_ring_kernel<G, MODE, STORE, 1, H>), grid, dim3(64), lds, stream, a);
    } else {
        EDLIB_AMD_CHECK_STATIC_LDS((scan_pairs_ring_kernel<G, MODE, STORE, 0, H>), 0);
        hipLaunchKernelGGL((scan_pairs_ring_kernel<G, MODE, STORE, 0, H>), grid, dim3(64), tgt, stream, a);
    }
    if (a.wordSteps) return launch_count_ring_steps(a.descs, a.numUnits, MODE, H, a.wordSteps, stream);
    return hipGetLastError();
}

hipError_t launch_scan_pairs_ring(int G, int mode, bool store, const PairScanArgs& a, hipStream_t stream, int H)
{
    if (a.numUnits == 0) return hipSuccess;
    if (H != 1) {                                                     // ring-lane blocks of 128 / 256 rows: 16-lane rings, distance only
        if (G != 16 || store || (H != 2 && H != 4)) return hipErrorInvalidValue;
        switch (mode * 8 + H) {
            case 2: return launch_scan_pairs_ring_t<16, 0, false, 2>(a, stream);
            case 4: return launch_scan_pairs_ring_t<16, 0, false, 4>(a, stream);
            case 10: return launch_scan_pairs_ring_t<16, 1, false, 2>(a, stream);
            case 12: return launch_scan_pairs_ring_t<16, 1, false, 4>(a, stream);
            case 18: return launch_scan_pairs_ring_t<16, 2, false, 2>(a, stream);
            case 20: return launch_scan_pairs_ring_t<16, 2, false, 4>(a, stream);
        }
        return hipErrorInvalidValue;
    }
    if (mode != 0 && (store || (G != 4 && G != 8 && G != 16))) return hipErrorInvalidValue;   // semi-global rings: 4, 8 or 16 lanes, distance only
    switch (G * 8 + mode * 2 + (store ? 1 : 0)) {
        case 32: return launch_scan_pairs_ring_t<4, 0, false>(a, stream);
        case 33: return launch_scan_pairs_ring_t<4, 0, true>(a, stream);
        case 34: return launch_scan_pairs_ring_t<4, 1, false>(a, stream);
        case 36: return launch_scan_pairs_ring_t<4, 2, false>(a, stream);
        case 64: return launch_scan_pairs_ring_t<8, 0, false>(a, stream);
        case 65: return launch_scan_pairs_ring_t<8, 0, true>(a, stream);
        case 66: return launch_scan_pairs_ring_t<8, 1, false>(a, stream);
        case 68: return launch_scan_pairs_ring_t<8, 2, false>(a, stream);
        case 128: return launch_scan_pairs_ring_t<16, 0, false>(a, stream);
        case 129: return launch_scan_pairs_ring_t<16, 0, true>(a, stream);
        case 130: return launch_scan_pairs_ring_t<16, 1, false>(a, stream);
        case 132: return launch_scan_pairs_ring_t<16, 2, false>(a, stream);
        case 168: return launch_scan_pairs_ring_t<21, 0, false>(a, stream);
        case 169: return launch_scan_pairs_ring_t<21, 0, true>(a, stream);
        case 256: return launch_scan_pairs_ring_t<32, 0, false>(a, stream);
        case 257: return launch_scan_pairs_ring_t<32, 0, true>(a, stream);
        case 512: return launch_scan_pairs_ring_t<64, 0, false>(a, stream);
        case 513: return launch_scan_pairs_ring_t<64, 0, true>(a, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------- Hirschberg split

// value of the cell at row r of a dumped column (same decoding as the traceback's left neighbour)
__device__ __forceinline__ int column_cell(const u64* P, const u64* M, const int* S, long long off, int r)
{
    const int b = r >> 6, bit = r & 63;
    const u64 above = (bit == 63) ? 0ull : (~0ull << (bit + 1));
    return S[off + b] - __popcll(P[off + b] & above) + __popcll(M[off + b] & above);
}

// reference edlib.cpp:1314-1353 ("find the best move").  One workgroup per piece.
__global__ void __launch_bounds__(256)
hirschberg_split_kernel(const SplitArgs a)
{
    __shared__ int s_first;
    const int p = blockIdx.x;
    const PairDesc f = a.descs[2 * p], r = a.descs[2 * p + 1];
    const int m = f.qlen, best = a.best[p];
    const int lw = f.tlen, rw = r.tlen;
    if (threadIdx.x == 0) s_first = 0x7fffffff;
    __syncthreads();
    // L[i]   = D(query[0..i], left half)            = forward column cell i
    // R[i+1] = D(query[i+1..m-1], right half)       = reverse column cell (m-1) - (i+1)
    for (int i = threadIdx.x; i <= m - 2; i += blockDim.x) {
        const int L = column_cell(a.colP, a.colM, a.colS, f.colOff, i);
        const int R = column_cell(a.colP, a.colM, a.colS, r.colOff, m - 2 - i);
        if (L + R == best) atomicMin(&s_first, i);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int i = s_first, ls = -1, rs = -1;
        if (i != 0x7fffffff) {
            ls = column_cell(a.colP, a.colM, a.colS, f.colOff, i);
            rs = column_cell(a.colP, a.colM, a.colS, r.colOff, m - 2 - i);
        } else {
            const int R0 = column_cell(a.colP, a.colM, a.colS, r.colOff, m - 1);     // whole query vs right half
            const int Lm = column_cell(a.colP, a.colM, a.colS, f.colOff, m - 1);     // whole query vs left half
            if (lw + R0 == best) { i = -1; ls = lw; rs = R0; }                       // :1337-1344
            else if (Lm + rw == best) { i = m - 1; ls = Lm; rs = rw; }               // :1345-1353
            else i = -2;
        }
        a.out[3 * p] = i; a.out[3 * p + 1] = ls; a.out[3 * p + 2] = rs;
    }
}

hipError_t launch_hirschberg_split(const SplitArgs& a, hipStream_t stream)
{
    if (a.numPieces == 0) return hipSuccess;
    hipLaunchKernelGGL(hirschberg_split_kernel, dim3(a.numPieces), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// The same pair of scans finds a DISTANCE: every alignment crosses from the left half of the target into the right half at
// some query row i, so D[m][T] = min over i in [-1, m-1] of L[i] + R[i+1] (L, R as above; L[-1] = lw, R[m] = rw).  Two
// half scans of T / 2 columns run side by side where one scan of T columns is T dependent steps.  Cells
// outside the band are upper bounds, so the minimum is exact iff it is <= the scans' threshold.  Pass 1: packed
// (sum << 32 | i) minima over the interior rows; pass 2: the boundary cases and the reference's preference among equal
// sums (edlib.cpp:1321-1353: first interior row, then i = -1, then i = m-1) -- so the result is also the level-0 split of
// the piece's path.  out[4p..4p+3] = {best, i, leftScore, rightScore}.
__global__ void __launch_bounds__(256)
split_min_rows_kernel(const SplitArgs a, unsigned long long* packed)
{
    const int p = blockIdx.y;
    const PairDesc f = a.descs[2 * p], r = a.descs[2 * p + 1];
    const int m = f.qlen;
    unsigned long long mine = ~0ull;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= (long long)m - 2; i += (long long)gridDim.x * blockDim.x) {
        const long long L = column_cell(a.colP, a.colM, a.colS, f.colOff, (int)i);
        const long long R = column_cell(a.colP, a.colM, a.colS, r.colOff, m - 2 - (int)i);
        const unsigned long long v = ((unsigned long long)(L + R) << 32) | (unsigned long long)i;
        mine = v < mine ? v : mine;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(mine, off, 64);
        mine = o < mine ? o : mine;
    }
    if ((threadIdx.x & 63) == 0 && mine != ~0ull) atomicMin(&packed[p], mine);
}
__global__ void __launch_bounds__(64)
split_min_finish_kernel(const SplitArgs a, const unsigned long long* packed)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= a.numPieces) return;
    const PairDesc f = a.descs[2 * p], r = a.descs[2 * p + 1];
    const int m = f.qlen, lw = f.tlen, rw = r.tlen;
    const long long R0 = column_cell(a.colP, a.colM, a.colS, r.colOff, m - 1);        // whole query vs right half
    const long long Lm = column_cell(a.colP, a.colM, a.colS, f.colOff, m - 1);        // whole query vs left half
    const unsigned long long pk = packed[p];
    long long best = pk == ~0ull ? (1LL << 40) : (long long)(pk >> 32);
    int row = pk == ~0ull ? -2 : (int)(unsigned)(pk & 0xffffffffu);
    if (lw + R0 < best) { best = lw + R0; row = -1; }
    if (Lm + rw < best) { best = Lm + rw; row = m - 1; }
    int ls, rs;
    if (row == -1) { ls = lw; rs = (int)R0; }
    else if (row == m - 1) { ls = (int)Lm; rs = rw; }
    else { ls = column_cell(a.colP, a.colM, a.colS, f.colOff, row); rs = column_cell(a.colP, a.colM, a.colS, r.colOff, m - 2 - row); }
    a.out[4 * p] = best > 0x3fffffff ? 0x3fffffff : (int)best; a.out[4 * p + 1] = row; a.out[4 * p + 2] = ls; a.out[4 * p + 3] = rs;
}

hipError_t launch_split_min(const SplitArgs& a, unsigned long long* packed, int maxRows, hipStream_t stream)
{
    if (a.numPieces == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(packed, 0xff, (size_t)a.numPieces * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    int chunks = (maxRows + 256 * 16 - 1) / (256 * 16);
    chunks = chunks < 1 ? 1 : (chunks > 1024 ? 1024 : chunks);
    hipLaunchKernelGGL(split_min_rows_kernel, dim3(chunks, a.numPieces), dim3(256), 0, stream, a, packed);
    hipLaunchKernelGGL(split_min_finish_kernel, dim3((a.numPieces + 63) / 64), dim3(64), 0, stream, a, packed);
    return hipGetLastError();
}

// --------------------------------------------------------------- traceback

// reference obtainAlignmentTraceback (edlib.cpp:942-1141), one lane per unit.
// Walks from (m-1, T-1) to the origin on the stored (P, M, score) columns.  A
// cell's neighbours are decoded on demand: up from the vertical delta bit of
// the current column (:1007-1013), left from the left column's block score by
// peeling the rows below it (:986-993, here a popcount), upper-left from left
// and its vertical delta (:996-1000).  Candidate order up > left > diagonal
// (:1020, 1054, 1085).  Ops are written back-to-front into the END of the
// unit's range, so no reversal pass (:1138-1139) is needed.
// One unit per LANE: 64 independent walks per wave keep 64 dependent loads in flight.  Two round-2 alternatives
// with one WAVE per unit and the walk's window of the store kept on chip were both slower at config 5's shape
// (10,000 x 1 kb: 0.89 ms here): window in LDS 1.31 ms (three LDS round trips per step), window in registers with
// the whole walk in the scalar unit 1.66 ms (39 waves per CU share one scalar ALU).
// The walk through the column store, one unit per lane (reference: obtainAlignmentTraceback, edlib.cpp:942-1141; the
// move preference is the reference's: up, then left, then diagonal).  The store answers the three questions of a cell
// directly (StoreEntry, pair_kernels.hpp), so the walk needs neither scores nor the neighbours' values.
//
// Round 2 walked cell by cell over (Pv, Mv, score) entries: a dependent 32-byte load, two popcounts and ~250 issued
// instructions of divergent control flow per column, 0.9 us per column with one wave per SIMD and nothing to overlap
// with.  Round 3 found by measurement that the load latency is NOT what bounds it (a per-lane line cache in LDS 1.37 ms,
// an 8-column x 2-block LDS window 1.18 ms, a cyclic register ring 1.5 ms, and a clean 8-column register batch 0.93 ms,
// against 0.88 ms for the plain walk at config 5's shape): it is the instruction stream of a lone wave, and the byte
// store per op under every s_waitcnt vmcnt (gfx9 counts loads and stores in one counter).  So:
//   * the columns of a block row are consumed strictly right to left: each lane fetches the next kBatch columns of its
//     block row in one go and finds the left neighbour of sub-iteration j at the static batch index j;
//   * a run of up-moves is one count-leading-ones on the "up" plane instead of a loop of cell steps;
//   * left / diagonal is predicated straight-line code on two bit tests; the rare events (first row / column reached,
//     leaving the block row) set flags that are resolved outside the hot sequence: the boundary tails are written after
//     the loop, a lane that left its block row waits for the next batch;
//   * ops leave through a shift register, eight at a time.
// 0.88 -> 0.44 ms with the old entries, -> see profiles/README.md with the planes.
constexpr int kBatch = 8;
__global__ void __launch_bounds__(64)
traceback_kernel(const TracebackArgs a, const int lanesPerWave)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x * lanesPerWave + lane;
    const bool have = lane < lanesPerWave && unit < a.numUnits;
    const PairDesc d = a.descs[have ? unit : 0];
    const int m = d.qlen, T = d.tlen, nb = num_blocks(m);
    const long long slotAt = a.opsOff[have ? unit : 0];
    uint8_t* ops = a.ops + slotAt;
    const int slot = (int)(a.opsOff[have ? unit + 1 : 1] - slotAt);   // the unit's op slot (>= any alignment the walk can produce)
    int w = slot;                                   // next write index is --w
    int r = m - 1, c = T - 1;
    const StoreEntry* S = a.store + d.storeOff;
    // ring layout (scan_pairs_ring_kernel): only the blocks inside the band of threshold kinit exist.
    // The walk stays on cells of optimal paths, which are inside the band and exact; a neighbour outside
    // the band can never be "one less than here", so it is simply not a candidate (the reference's
    // stored band behaves the same, edlib.cpp:996-1016).  A block's first column inside the band is computed
    // against "+1 per row" to its left: Ph = 0 there, so the planes never offer the move either.
    const int G = d.ring;
    int dmin = 0;
    if (G) {
        const int D = T - m, absD = D < 0 ? -D : D, p = (d.kinit - absD) >> 1;
        dmin = (D < 0 ? D : 0) - p;
    }
    // a unit whose scan ended above its threshold has no exact cells to walk on (it is rescanned at the next level)
    const bool skip = !have || (G && a.score[unit] > d.kinit);
    bool done = skip;
    // entries of one block row lie at a fixed stride: index(c, b) = row_base(b) + c * stride
    auto row_base = [&](int blk) -> long long {
        if (G) return (long long)blk * G + (blk % G);
        const int strip = blk >> 6, l = blk & 63;
        const int nbS = (nb - strip * 64) < 64 ? (nb - strip * 64) : 64;
        return strip_base(strip, T) + (long long)l * nbS + l;
    };
    const int strideLast = (nb & 63) ? (nb & 63) : 64;      // entries per step of the last strip of the plain layout
    u64 Xc = 0, Yc = 0;                              // planes of column c, block r >> 6 (valid unless `fetch`)
    bool fetch = true;
    int tailOp = 0, tailCnt = 0;                     // the run along the matrix boundary, written after the loop
    u64 acc = 0; int cnt = 0;
    auto push = [&](int op) {
        acc = (acc << 8) | (u64)op;
        if (++cnt == 8) { w -= 8; *reinterpret_cast<u64*>(ops + w) = acc; cnt = 0; }
    };
    while (__builtin_amdgcn_ballot_w64(!done) != 0ull) {
        // ---- the batch: columns c-1 .. c-kBatch of block row b (columns left of 0 read column 0 and are never used)
        const int b = r >> 6, cb = c;
        uint4 bxy[kBatch];
        const int cs = G ? G : ((b >> 6) == ((nb - 1) >> 6) ? strideLast : 64);
        const StoreEntry* row = S + row_base(b);
        if (!done) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int col = cb - 1 - j;
                bxy[j] = *reinterpret_cast<const uint4*>(row + (long long)(col < 0 ? 0 : col) * cs);
            }
            if (fetch) { const uint4 v = *reinterpret_cast<const uint4*>(row + (long long)c * cs); Xc = ((u64)v.y << 32) | v.x; Yc = ((u64)v.w << 32) | v.z; fetch = false; }
        }
        bool live = !done;                           // still inside block row b with this batch
        const int bandLeft = G ? 64 * b + dmin : -0x40000000;   // block b exists in column x iff x >= bandLeft
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            // -- up-moves: the set bits of the "up" plane from row r upwards are one INSERT each
            {
                const int bit = r & 63;
                const u64 nx = ~((Xc & ~Yc) << (63 - bit));                        // row r at bit 63; the zeros shifted in end the run
                const int ups = live ? (nx ? __builtin_clzll(nx) : 64) : 0;        // leading ones, <= bit + 1
                if (ups) { for (int i = 0; i < ups; ++i) push(1); }
                r -= ups;
                const bool top = ups > bit;                                         // ran through the top of the block
                const bool endRow = top && r < 0;                                   // INSERT taken in row 0 (:1040-1046)
                tailOp = endRow ? 2 : tailOp; tailCnt = endRow ? c + 1 : tailCnt;
                done = done || endRow; fetch = fetch || (top && !endRow); live = live && !top;
            }
            // -- left or diagonal
            {
                const int bit = r & 63;
                const bool lbit = ((Xc & Yc) >> bit) & 1ull, ybit = (Yc >> bit) & 1ull;
                const bool left = lbit && (c == 0 || c - 1 >= bandLeft);            // (column -1 is the boundary: :976-980)
                if (live) push(left ? 2 : (ybit ? 0 : 3));
                c -= live ? 1 : 0;
                const u64 Xl = ((u64)bxy[j].y << 32) | bxy[j].x, Yl = ((u64)bxy[j].w << 32) | bxy[j].z;
                Xc = live ? Xl : Xc; Yc = live ? Yl : Yc;                           // block b of the column the walk is in now
                const bool endC = live && c < 0;
                const bool endR = live && !endC && !left && r == 0;
                tailOp = endC ? 1 : (endR ? 2 : tailOp);
                tailCnt = endC ? (left ? r + 1 : r) : (endR ? c + 1 : tailCnt);
                const bool dn = live && !left && !endC && !endR;
                r -= dn ? 1 : 0;
                const bool cross = dn && (r & 63) == 63;
                done = done || endC || endR; fetch = fetch || cross; live = live && !(endC || endR || cross);
            }
        }
        // a lane that used up its batch inside the block row keeps its planes; one that left it re-reads (fetch)
    }
    for (int i = 0; i < cnt; ++i) ops[w - cnt + i] = (uint8_t)(acc >> (8 * i));
    w -= cnt;
    for (int i = 0; i < tailCnt; ++i) ops[--w] = (uint8_t)tailOp;
    if (have) a.opsLen[unit] = skip ? 0 : slot - w;
}

hipError_t launch_traceback(const TracebackArgs& a, hipStream_t stream)
{
    if (a.numUnits == 0) return hipSuccess;
    // lanes per wave: a wave steps at the pace of its slowest lane (up-moves, block-row changes), and a batch of a few
    // thousand units leaves most of the 1024 SIMDs idle at 64 units per wave, so small batches spread out
    int lanes = 64;
    while (lanes > 16 && (a.numUnits + lanes - 1) / lanes < 2048) lanes >>= 1;
    hipLaunchKernelGGL(traceback_kernel, dim3((a.numUnits + lanes - 1) / lanes), dim3(64), 0, stream, a, lanes);
    return hipGetLastError();
}

}  // namespace edlib_amd
