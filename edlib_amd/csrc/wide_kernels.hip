// wide_kernels.hip -- one (query, target) unit on MANY waves: the band of a long NW pair beyond every lane ring, and the
// pipelined form of the 64-block strips for long SHW / HW queries.
//
// Replaces, for units of more than 64 blocks, myersCalcEditDistanceNW with a fixed k (reference edlib.cpp:730-928; the
// band is Ukkonen's, what the first/lastBlock bookkeeping of :744-830 converges to) and the column loop of
// myersCalcEditDistanceSemiGlobal (:550-704).  The lane rings of pair_kernels.hip hold bands up to K = 3968 on one wave;
// above that scan_pairs_kernel walked every block of every column with ONE wave, strip after strip (round 3: tens of
// seconds for the reference's 1 Mb x 1 Mb Chromosome pairs, test_data/perf_tests.sh:180-191).  Here:
//
//   * the query is cut into STRIPS of 64 blocks (4096 rows); strip s is one wave's work, lane = block, anti-diagonal
//     schedule inside the wave exactly as in scan_pairs_kernel (lane l updates column j - l when lane 0 is at column j);
//   * a strip only exists for the columns its blocks have inside the band: with d in [dmin, dmax] the diagonals a path of
//     cost <= K can visit (:744-830), strip s runs columns [4096 s + dmin, 4096 s + 4095 + dmax] clipped to the target.
//     All 64 lanes run that whole range (the blocks of a strip enter and leave the band within 4096 columns of each
//     other: (bw + 4096) / (bw + 64) of the minimal work at band width bw, 1.24x at K = 16k, 1.03x at 128k);
//   * the strips of a unit form a PIPELINE over `slots` resident waves (workgroup = one wave; slot j runs strips j,
//     j + slots, ...): strip s + 1 consumes the horizontal deltas of strip s's bottom row (hout -> hin, :781-785) about
//     100 columns behind it.  The deltas travel through HBM as 8-byte granules {tag = strip + 1, 16 columns x 2 bits},
//     written by ONE agent-scope store of lane 63 every 16 steps and polled with agent-scope loads: the data is the
//     flag (no fence, no separate counter; per-XCD L2s are not coherent, so both sides bypass them).  A consumer
//     prefetches the granules of its next 64 columns while it works on the current ones;
//   * a strip that starts after column 0 takes over the absolute score of the row above it from one more granule
//     (bottom score of strip s at column c0(s + 1) - 1) and starts "+1 per row" below it, the reference's new block
//     (:803-808); beyond the last column of the strip above, the row above delivers +1 per column (:779).  Cells outside
//     the band only ever enter as such upper bounds, so values <= K stay exact (Ukkonen);
//   * block scores are not followed per step: the 2-bit codes a lane emits are kept for 16 steps in the register that
//     becomes the granule, and folded into the score by two popcounts when it is flushed;
//   * NW: the last block's final state gives D[m][T] (:914-917); Hirschberg halves (bandT) stop at their column and dump
//     (Pv, Mv, score) of the blocks alive there (:1252-1260); SHW / HW: the lane of row m-1 follows its score and records
//     best / count / positions (:658-673), no band.
//
// With slots >= (bw + 4096) / 4224 + 2 no wave ever waits for a free slot and a unit takes about T + bw dependent steps
// whatever K is -- the time of ONE strip of the old kernel instead of nstrips of them.  Every spin is bounded (wall clock)
// and a launch-wide abort word turns a stuck hand-off into EDLIB_STATUS_ERROR instead of a hung queue.
#include "pair_kernels.hpp"
#include "block64.hpp"
#include <type_traits>

namespace edlib_amd {

typedef unsigned long long u64;
typedef uint32_t u32;

__host__ __device__ static inline int num_blocks(int m) { return (m + 63) >> 6; }

constexpr long long kStripRows = 64 * 64;
constexpr long long kWideInf = 1LL << 40;

struct WideGeom { long long dmin, dmax; };
// diagonals j - i a path of cost <= K can visit (NW); semi-global modes: the whole matrix
__host__ __device__ static inline WideGeom wide_geom(int mode, int m, int T, int bandT, int K)
{
    if (mode != 0) return WideGeom{-kWideInf, kWideInf};
    const long long D = (long long)(bandT ? bandT : T) - m, absD = D < 0 ? -D : D;
    const long long p = ((long long)K - absD) >> 1;
    return WideGeom{(D < 0 ? D : 0) - p, (D > 0 ? D : 0) + p};
}
__host__ __device__ static inline long long wide_per_slot(int T) { return ((long long)T + 15) / 16 + 10; }

long long wide_stream_words(int tlen, int slots) { return (long long)slots * wide_per_slot(tlen); }

int wide_slots_wanted(int mode, int qlen, int tlen, int bandT, int K)
{
    const int nstrips = (num_blocks(qlen) + 63) / 64;
    const WideGeom g = wide_geom(mode, qlen, tlen, bandT, K);
    long long bw = g.dmax - g.dmin + 1;
    if (bw > tlen) bw = tlen;
    if (bw < 0) bw = 0;
    const long long live = (bw + 4096 + 63) / (4096 + 128) + 2;
    return (int)(live < nstrips ? live : nstrips);
}

long long wide_word_steps(int mode, int qlen, int tlen, int bandT, int K)
{
    const int nb = num_blocks(qlen), nstrips = (nb + 63) / 64;
    const WideGeom g = wide_geom(mode, qlen, tlen, bandT, K);
    long long v = 0;
    for (int s = 0; s < nstrips; ++s) {
        const long long r0 = kStripRows * s;
        long long c0 = r0 + g.dmin, c1 = r0 + kStripRows - 1 + g.dmax;
        if (c0 < 0) c0 = 0;
        if (c1 > tlen - 1) c1 = tlen - 1;
        if (c0 > c1) break;
        const int nbS = (nb - s * 64) < 64 ? (nb - s * 64) : 64;
        v += 2LL * nbS * (c1 - c0 + 1);
    }
    return v;
}

__device__ __forceinline__ u64 ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One poll failed: sleep; every so often look at the launch's abort word and at the wall clock (100 MHz).  True = give up.
__device__ __forceinline__ bool wide_spin_fail(unsigned& spins, const long long t0, unsigned* abortWord)
{
    __builtin_amdgcn_s_sleep(4);
    if ((++spins & 255u) != 0u) return false;
    if (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    if ((long long)wall_clock64() - t0 > 3000000000LL) {                // 30 s without the upstream strip moving
        __hip_atomic_store(abortWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

template <int MODE, bool LDSPEQ>
__global__ void __launch_bounds__(64)
scan_pairs_wide_kernel(const PairScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u64 s_peq[];      // [sigmaT][64]
    const int lane = threadIdx.x;
    const int slot = blockIdx.x, W = gridDim.x, unit = blockIdx.y;
    const PairDesc d = a.descs[unit];
    const int m = d.qlen, T = d.tlen, K = d.kinit;
    const int nb = num_blocks(m), nstrips = (nb + 63) >> 6;
    const u32 sh = (u32)(m - 1) & 63u;                               // row m-1 inside the last block
    if (MODE == 0) {
        const long long D = (long long)(d.bandT ? d.bandT : T) - m, absD = D < 0 ? -D : D;
        if ((long long)K < absD) {                                   // no path of cost <= K exists (edlib.cpp:749-754)
            if (slot == 0 && lane == 0) { a.outScore[unit] = 0x3fffffff; a.outCount[unit] = 0; a.outLast[unit] = -1; }
            return;
        }
    }
    const WideGeom g = wide_geom(MODE, m, T, d.bandT, K);
    const long long per = wide_per_slot(T);
    u64* const sbase = a.wstream + d.auxOff;
    int best = d.kinit, cnt = 0, lastCol = -1;                       // MODE != 0: columns scoring <= best qualify
    int* const pos = a.posPool + d.posOff;
    const bool dumpCol = a.colP != nullptr && d.colOff >= 0;
    const u32 rowAbove = (MODE == 2) ? 0u : 0x55555555u;             // 16 codes of row -1: HW 0, SHW / NW +1 (:584, 779)
    const long long clk0 = (long long)wall_clock64();
    unsigned spins = 0;

    for (int s = slot; s < nstrips; s += W) {
        const long long r0 = kStripRows * s;
        const long long c0l = r0 + g.dmin, c1l = r0 + kStripRows - 1 + g.dmax;
        const int c0 = c0l < 0 ? 0 : (c0l > T ? T : (int)c0l);
        const int c1 = c1l > T - 1 ? T - 1 : (int)c1l;
        if (c0 > c1) break;                                          // below the band at every column: so is everything after it
        const long long n0l = c0l + kStripRows, n1l = c1l + kStripRows;
        const int nextC0 = n0l < 0 ? 0 : (n0l > T ? T : (int)n0l);
        const bool nextLive = s + 1 < nstrips && nextC0 <= (n1l > T - 1 ? T - 1 : (int)n1l);
        const int startCol = nextC0 - 1;                             // the strip below starts from our bottom score at this column
        const bool hasUp = s > 0;
        const long long u1l = c1l - kStripRows;
        const int upC1 = u1l > T - 1 ? T - 1 : (int)u1l;             // last column of the strip above
        const u64* const upS = sbase + (long long)((s - 1 + W) % W) * per;
        u64* const myS = sbase + (long long)slot * per;
        const u32 tagUp = (u32)s, myTag = (u32)s + 1u;

        const int nbS = (nb - s * 64) < 64 ? (nb - s * 64) : 64;
        const int b = s * 64 + lane;
        const bool laneOn = lane < nbS;
        const bool tracker = b == nb - 1;                            // lane that owns row m-1
        const unsigned long long* const peqRow = a.peq + d.peqOff + b;

        if (LDSPEQ) {
            __syncthreads();                                         // previous strip done with s_peq
            for (int sy = 0; sy < a.sigmaT; ++sy)
                s_peq[sy * 64 + lane] = laneOn ? a.peq[d.peqOff + (long long)sy * nb + b] : 0ull;
            __syncthreads();
        }

        // ---- score of the row above the strip at column c0 - 1
        int top = (int)r0;                                           // D[r0 - 1][-1] = r0 (edlib.cpp:575-579)
        if (c0 > 0) {
            u64 v = ld_agent(upS);
            while ((u32)(v >> 32) != tagUp) {
                if (wide_spin_fail(spins, clk0, a.wabort)) return;
                v = ld_agent(upS);
            }
            top = (int)(u32)v;
        }
        Block64 B{~0u, ~0u, 0u, 0u};                                 // "+1 per row" (:759-763, 803-808)
        int bscore = top + 64 * (lane + 1);                          // bottom of this lane's block at column c0 - 1
        int sc = top + (m - (int)r0);                                // row m-1 at column c0 - 1 (tracker)
        u32 acc = 0;                                                 // the codes this lane emitted since the last fold, newest on top

        // ---- target symbols (as LDS row offsets, symbol * 512) 64 columns per load, one chunk ahead
        auto load_t = [&](const int base) -> u32 {
            const int c = base + lane;
            return (c < T) ? (u32)a.tlut[a.tpool[d.toff + (long long)c * d.tstep]] << 9 : 0u;
        };
        const int base0 = c0 & ~63;
        u32 tcur = load_t(base0), tnext = load_t(base0 + 64);
        // ---- deltas of the row above: four granules (64 columns) per load, one chunk ahead
        auto load_h = [&](const int base) -> u64 { return ld_agent(upS + 1 + (base >> 4) + (lane & 3)); };
        auto h_ok = [&](const int base, const u64 v) -> bool {
            const int gc = base + 16 * (lane & 3);
            const bool needed = gc + 15 >= c0 && gc <= upC1;
            return !needed || (u32)(v >> 32) == tagUp;
        };
        auto h_data = [&](const int base, const u64 v) -> u32 {
            const int gc = base + 16 * (lane & 3);
            if (gc > upC1) return 0x55555555u;                       // beyond the life of the strip above: +1 per column
            u32 w = (u32)v;
            if (gc + 15 > upC1) {                                    // its last granule: the columns behind upC1 likewise
                const u32 mask = (1u << (2 * (upC1 - gc + 1))) - 1u;
                w = (w & mask) | (0x55555555u & ~mask);
            }
            return w;
        };
        u32 hcur = rowAbove;
        u64 hnext = 0;
        if (hasUp) {
            u64 v = load_h(base0);
            while (__builtin_amdgcn_ballot_w64(!h_ok(base0, v)) != 0ull) {
                if (wide_spin_fail(spins, clk0, a.wabort)) return;
                v = load_h(base0);
            }
            hcur = h_data(base0, v);
            hnext = load_h(base0 + 64);
        }

        const int span = c1 - c0;                                    // lane l is active at steps l .. l + span
        const int nsteps = span + 64;
        u32 carry = 0;
        bool bail = false;

        // One step: lane 0 is at column j = c0 + t.  The word that travels one lane down per step is
        // {LDS offset of the receiver's Peq word of its NEXT column, code of the sender's hout}: offset = symbol * 512
        // + lane * 8 grows by 8 per hop, the code sits in its low bits.  Lane 0 is fed {symbol of column j + 1, row
        // above at column j} through the DPP `old` operand.
        auto step = [&](auto fullTag, const int t, const u64 eqCur, u64& eqNxt) {
            constexpr bool FULL = decltype(fullTag)::value;
            const int j = c0 + t, jn = j + 1;
            if (t >= 0 && (jn & 63) == 0) { tcur = tnext; tnext = load_t(jn + 64); }
            if (hasUp && t > 0 && (j & 63) == 0) {
                u64 v = hnext;
                while (__builtin_amdgcn_ballot_w64(!h_ok(j, v)) != 0ull) {
                    if (wide_spin_fail(spins, clk0, a.wabort)) { bail = true; break; }
                    v = load_h(j);
                }
                hcur = h_data(j, v);
                hnext = load_h(j + 64);
            }
            const u32 symw = (u32)__builtin_amdgcn_readlane((int)tcur, jn & 63);
            const u32 hw = (u32)__builtin_amdgcn_readlane((int)hcur, (j >> 4) & 3);
            const u32 in0 = symw | ((hw >> (2 * (j & 15))) & 3u);
            const u32 x = (u32)__builtin_amdgcn_update_dpp((int)in0, (int)carry, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
            const u32 addr = x & ~7u;
            if (LDSPEQ) eqNxt = *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(s_peq) + addr);
            else eqNxt = laneOn ? peqRow[(long long)(x >> 9) * nb] : 0ull;
            u32 code = 0;
            const int rel = t - lane;
            if (FULL || (laneOn && (unsigned)rel <= (unsigned)span)) {
                u32 ph0, ph1, mh0, mh1;
                advance_block64(B, (u32)eqCur, (u32)(eqCur >> 32), x & 1u, (x >> 1) & 1u, ph0, ph1, mh0, mh1);
                code = (ph1 >> 31) | ((mh1 >> 31) << 1);
                acc = (acc >> 2) | (code << 30);
                if (MODE != 0) {
                    if (tracker) {
                        const u64 ph = ((u64)ph1 << 32) | ph0, mh = ((u64)mh1 << 32) | mh0;
                        const int col = c0 + rel;
                        sc += (int)((ph >> sh) & 1ull) - (int)((mh >> sh) & 1ull);
                        if (sc <= best && col >= d.skip) {           // edlib.cpp:658-673
                            if (sc < best) { best = sc; cnt = 0; }
                            if (cnt < d.posCap) pos[cnt] = col;
                            ++cnt;
                            lastCol = col;
                        }
                    }
                }
            }
            carry = addr + 8u + code;                                // every lane forwards the symbol stream
            // ---- every 16 steps (uniform in t) the codes are folded into the block score; the 16 codes of lane 63 are the
            // granule of its columns c63 - 15 .. c63 (its last granule holds fewer: moved down to their columns' bits)
            const int c63 = j - 63;
            const bool live63 = nextLive && c63 >= c0 && c63 <= c1;
            if (live63 && c63 == startCol) {
                const int now = bscore + __popc(acc & 0x55555555u) - __popc(acc & 0xaaaaaaaau);
                if (lane == 63) st_agent(myS, ((u64)myTag << 32) | (u32)now);
            }
            if ((c63 & 15) == 15 || c63 == c1) {
                if (live63 && lane == 63) st_agent(myS + 1 + (c63 >> 4), ((u64)myTag << 32) | (acc >> (2 * (15 - (c63 & 15)))));
                bscore += __popc(acc & 0x55555555u) - __popc(acc & 0xaaaaaaaau);
                acc = 0;
            }
        };
        u64 eqA = 0, eqB = 0;
        for (int t = -1; t < nsteps && !bail; t += 2) {
            if (t >= 63 && t + 1 <= span) {                          // every lane inside its columns
                step(std::true_type{}, t, eqA, eqB);
                step(std::true_type{}, t + 1, eqB, eqA);
            } else {
                step(std::false_type{}, t, eqA, eqB);
                if (t + 1 < nsteps) step(std::false_type{}, t + 1, eqB, eqA);
            }
        }
        if (bail) return;
        bscore += __popc(acc & 0x55555555u) - __popc(acc & 0xaaaaaaaau);       // (lanes above 63's last flush)
        if (laneOn && c1 == T - 1 && dumpCol) {                      // stop column of a Hirschberg half
            a.colP[d.colOff + b] = ((u64)B.p1 << 32) | B.p0; a.colM[d.colOff + b] = ((u64)B.m1 << 32) | B.m0;
            a.colS[d.colOff + b] = bscore;
        }
        if (tracker) {
            if (MODE == 0) {
                if (c1 == T - 1) {
                    // D[m][T] from the bottom score of row m-1's block and the vertical deltas below row m-1 (edlib.cpp:914-917)
                    const u64 P = ((u64)B.p1 << 32) | B.p0, M = ((u64)B.m1 << 32) | B.m0;
                    const u64 below = (sh == 63u) ? 0ull : (~0ull << (sh + 1));
                    a.outScore[unit] = bscore - __popcll(P & below) + __popcll(M & below);
                    a.outCount[unit] = 1; a.outLast[unit] = T - 1;
                }
            } else {
                a.outScore[unit] = cnt > 0 ? best : -1; a.outCount[unit] = cnt; a.outLast[unit] = lastCol;
            }
        }
    }
}

template <int MODE>
static hipError_t launch_wide_t(const PairScanArgs& a, int slots, hipStream_t stream)
{
    const dim3 grid(slots, a.numUnits);
    if (a.sigmaT <= 32) {
        const size_t lds = (size_t)a.sigmaT * 64 * sizeof(u64);
        hipLaunchKernelGGL((scan_pairs_wide_kernel<MODE, true>), grid, dim3(64), lds, stream, a);
    } else {
        hipLaunchKernelGGL((scan_pairs_wide_kernel<MODE, false>), grid, dim3(64), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_scan_pairs_wide(int mode, const PairScanArgs& a, int slots, hipStream_t stream)
{
    if (a.numUnits == 0) return hipSuccess;
    if (slots < 1 || !a.wstream || !a.wabort) return hipErrorInvalidValue;
    switch (mode) {
        case 0: return launch_wide_t<0>(a, slots, stream);
        case 1: return launch_wide_t<1>(a, slots, stream);
        case 2: return launch_wide_t<2>(a, slots, stream);
    }
    return hipErrorInvalidValue;
}

// waves of this kernel the device holds at once (the pipeline's hand-offs spin: every workgroup of a launch must be resident)
int wide_resident_waves(int sigmaT)
{
    int dev = 0, cus = 0, perCu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    hipError_t e;
    if (sigmaT <= 32) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, scan_pairs_wide_kernel<0, true>, 64, (size_t)sigmaT * 512);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, scan_pairs_wide_kernel<0, false>, 64, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (perCu > 8) perCu = 8;                                        // two waves per SIMD: more only share its issue slots
    if (perCu > 1) perCu -= 1;                                       // margin (the occupancy query can be one block per CU high)
    return cus * perCu;
}

}  // namespace edlib_amd
