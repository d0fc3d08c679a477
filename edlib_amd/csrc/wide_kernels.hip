// wide_kernels.hip -- one (query, target) unit on MANY waves: the band of a long NW pair beyond every lane ring, and the
// pipelined form of the strips for long SHW / HW queries.
//
// Replaces, for long units, myersCalcEditDistanceNW with a fixed k (reference edlib.cpp:730-928; the band is Ukkonen's,
// what the first/lastBlock bookkeeping of :744-830 converges to) and the column loop of myersCalcEditDistanceSemiGlobal
// (:550-704).  The lane rings of pair_kernels.hip hold bands up to K = 4031 on one wave; above that scan_pairs_kernel
// walked every block of every column with ONE wave, strip after strip (round 3: tens of seconds for the reference's
// 1 Mb x 1 Mb Chromosome pairs, test_data/perf_tests.sh:180-191).  Here:
//
//   * the query is cut into STRIPS of 64 lanes x one 32-ROW WORD (2048 rows); strip s is one wave's work, anti-diagonal
//     schedule inside the wave (lane l updates column j - l when lane 0 is at column j).  32-row words, not the
//     reference's 64-row blocks: a unit on this kernel is a chain of dependent steps on waves that have their SIMD to
//     themselves, where every issued instruction costs 2.2-3.5 ns whatever it is (tools/lone_wave_ubench.hip), and the
//     block update (calculateBlock, :412-447) is 13 VALU instructions on a 32-bit word against 22 on a 64-bit pair;
//   * a strip only exists for the columns its rows have inside the band: with d in [dmin, dmax] the diagonals a path of
//     cost <= K can visit (:744-830), strip s runs columns [2048 s + dmin, 2048 s + 2047 + dmax] clipped to the target.
//     All 64 lanes run that whole range ((bw + 2048) / (bw + 32) of the minimal work at band width bw);
//   * the strips of a unit form a PIPELINE over `slots` resident waves (workgroup = one wave; slot j runs strips j,
//     j + slots, ...): strip s + 1 consumes the horizontal deltas of strip s's bottom row (hout -> hin, :781-785) about
//     100 columns behind it.  The deltas travel through HBM as 8-byte granules {tag = strip + 1, 16 columns x (+1 bit,
//     -1 bit)}, written by ONE agent-scope store of lane 63 every 16 steps and polled with agent-scope loads: the data
//     is the flag (no fence, no separate counter; per-XCD L2s are not coherent, so both sides bypass them);
//   * what lane 0 is fed per step -- the delta of the row above and the next target symbol -- is prepared 64 columns at
//     a time by all lanes and parked in an LDS ring: one broadcast LDS read per step instead of scalar-unit glue;
//   * sixteen steps at a time run as one straight-line block (no branch, no rare event inside); everything rare -- chunk
//     rotation, polling, publishing, lanes outside their columns -- lives in a generic step between blocks;
//   * a strip that starts after column 0 takes over the absolute score of the row above it from one more granule
//     (bottom score of strip s at column c0(s + 1) - 1) and starts "+1 per row" below it, the reference's new block
//     (:803-808); beyond the last column of the strip above, the row above delivers +1 per column (:779).  Cells outside
//     the band only ever enter as such upper bounds, so values <= K stay exact (Ukkonen);
//   * word scores are not followed per step: the hout bits a lane emits are shifted into two registers that become the
//     granule, and folded into the score by two popcounts when it is flushed;
//   * NW: the last word's final state gives D[m][T] (:914-917); Hirschberg halves (bandT) stop at their column and dump
//     (Pv, Mv, score) of the blocks alive there (:1252-1260); SHW / HW: the lane of row m-1 follows its score and records
//     best / count / positions (:658-673), no band.
//
// With slots >= (bw + 2048) / 2176 + 2 no wave ever waits for a free slot and a unit takes about T + bw dependent steps
// whatever K is (0.074 us each on an MI355X).  The workgroups of a launch check at entry that they are all on the device
// together (wide_all_resident); every later spin is bounded by the wall clock PER WAIT; either failure sets the launch's
// abort word, every wave leaves, and the host runs the same units with one slot each -- a wave that only reads granules it
// wrote itself never waits -- so a call still returns its answer (the reference always does: edlib.cpp:197-217).
#include "pair_kernels.hpp"
#include "block64.hpp"
#include "lds_check.hpp"
#include <type_traits>

namespace edlib_amd {

typedef unsigned long long u64;
typedef uint32_t u32;

__host__ __device__ static inline int num_blocks(int m) { return (m + 63) >> 6; }

constexpr long long kStripRows = 64 * 32;                              // a strip: 64 lanes x one 32-row word
__host__ __device__ static inline int num_words(int m) { return (m + 31) >> 5; }
constexpr long long kWideInf = 1LL << 40;

struct WideGeom { long long dmin, dmax; };
// diagonals j - i a path of cost <= K can visit (NW); semi-global modes: the whole matrix
__host__ __device__ static inline WideGeom wide_geom(int mode, int m, int T, int bandT, int K)
{
    if (mode == 1 && bandT < 0) return WideGeom{-(long long)K, (long long)K};     // SHW inside the band of threshold K: |i - j| <= K
    if (mode == 2 && bandT < 0) return WideGeom{-(long long)K, (long long)(T > m ? T - m : 0) + 2LL * K};   // HW: starts in [0, T - m + K], K diagonals either side
    if (mode != 0) return WideGeom{-kWideInf, kWideInf};
    const long long D = (long long)(bandT > 0 ? bandT : T) - m, absD = D < 0 ? -D : D;
    const long long p = ((long long)K - absD) >> 1;
    return WideGeom{(D < 0 ? D : 0) - p, (D > 0 ? D : 0) + p};
}
__host__ __device__ static inline long long wide_per_slot(int T) { return ((long long)T + 15) / 16 + 10; }

long long wide_stream_words(int tlen, int slots) { return (long long)slots * wide_per_slot(tlen); }

int wide_slots_wanted(int mode, int qlen, int tlen, int bandT, int K)
{
    const int nstrips = (num_words(qlen) + 63) / 64;
    const WideGeom g = wide_geom(mode, qlen, tlen, bandT, K);
    long long bw = g.dmax - g.dmin + 1;
    if (bw > tlen) bw = tlen;
    if (bw < 0) bw = 0;
    const long long live = (bw + kStripRows + 63) / (kStripRows + 128) + 2;
    return (int)(live < nstrips ? live : nstrips);
}

long long wide_word_steps(int mode, int qlen, int tlen, int bandT, int K)
{
    const int nb = num_words(qlen), nstrips = (nb + 63) / 64;
    const WideGeom g = wide_geom(mode, qlen, tlen, bandT, K);
    long long v = 0;
    for (int s = 0; s < nstrips; ++s) {
        const long long r0 = kStripRows * s;
        long long c0 = r0 + g.dmin, c1 = r0 + kStripRows - 1 + g.dmax;
        if (c0 < 0) c0 = 0;
        if (c1 > tlen - 1) c1 = tlen - 1;
        if (c0 > c1) break;
        const int nbS = (nb - s * 64) < 64 ? (nb - s * 64) : 64;
        v += (long long)nbS * (c1 - c0 + 1);
    }
    return v;
}

__device__ __forceinline__ u64 ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// abort word of a launch: 0 = fine, kWideAbortStalled = a hand-off made no progress, kWideAbortNotResident = the launch's
// workgroups were not all on the device together.  Either way every wave leaves at its next poll and the host runs the
// units again with ONE slot each (a wave then only reads granules it wrote itself: nothing to wait for, edlib.cpp:197-217
// always returns).
constexpr unsigned kWideAbortStalled = 1u, kWideAbortNotResident = 2u;

// One poll failed: sleep; every so often look at the launch's abort word and at the wall clock (100 MHz).  True = give up.
// t0 belongs to ONE wait (the caller zeroes it before the wait's first poll): the limit measures how long THIS hand-off
// has made no progress, not how long the launch has been running (a launch of many strips per slot may run for minutes).
__device__ __forceinline__ bool wide_spin_fail(unsigned& spins, long long& t0, unsigned* abortWord)
{
    __builtin_amdgcn_s_sleep(4);
    if ((++spins & 255u) != 0u) return false;
    if (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) t0 = now;
    else if (now - t0 > 1000000000LL) {                                  // 10 s without the upstream strip moving
        __hip_atomic_store(abortWord, kWideAbortStalled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

// Residency by contract: the strips of a unit wait for each other across workgroups, which only ends if every workgroup of
// the launch is on the device at the same time.  The host sizes a launch for that (wide_resident_waves), but it cannot see
// another process on the device or a tool that holds workgroups back -- so the kernel checks: every workgroup counts
// itself in and waits (0.2 s at most) until all have; one that gives up marks the launch "not resident".  A launch with
// one slot per unit has no cross-workgroup hand-off and skips this.
__device__ __forceinline__ bool wide_all_resident(unsigned* abortWord, const unsigned total)
{
    unsigned ok = 1u;
    if (threadIdx.x == 0) {
        unsigned* arrived = abortWord + 1;
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long t0 = 0; unsigned spins = 0;
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 63u) != 0u) continue;
            if (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0u; break; }
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 20000000LL) {                            // 0.2 s: a launch that fits is resident within microseconds
                __hip_atomic_store(abortWord, kWideAbortNotResident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0u; break;
            }
        }
    }
    return __builtin_amdgcn_readfirstlane(ok) != 0u;
}

#define BITOP3_OR_NOR32(a, b, c)  ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xf1))   /* a | ~(b | c)  */
#define BITOP3_XOR_OR32(a, b, c)  ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xde))   /* (a ^ c) | b   */
#define BITOP3_BFI(m, a, b)       ((u32)__builtin_amdgcn_bitop3_b32((m), (a), (b), 0xca))   /* m ? a : b     */

template <int MODE, bool LDSPEQ>
__global__ void __launch_bounds__(64)
scan_pairs_wide_kernel(const PairScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_peq32[];    // [sigmaT][64] words of 32 rows
    const int lane = threadIdx.x;
    const int slot = blockIdx.x, W = gridDim.x, unit = blockIdx.y;
    if (W > 1 && !wide_all_resident(a.wabort, a.wideExpect ? a.wideExpect : gridDim.x * gridDim.y)) return;     // (before any other exit: everybody counts)
    const PairDesc d = a.descs[unit];
    const int m = d.qlen, T = d.tlen, K = d.kinit;
    const int nw = num_words(m), nb64 = num_blocks(m), nstrips = (nw + 63) >> 6;
    const u32 sh = (u32)(m - 1) & 31u;                               // row m-1 inside the last word
    if (MODE == 0) {
        const long long D = (long long)(d.bandT > 0 ? d.bandT : T) - m, absD = D < 0 ? -D : D;
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
    // a granule's data word: bit 15 - (c & 15) = "hout of column c is +1", bit 31 - (c & 15) = "is -1"
    const u32 rowAbove = (MODE == 2) ? 0u : 0x0000ffffu;             // 16 columns of row -1: HW 0, SHW / NW +1 (:584, 779)
    long long waitT0 = 0;                                            // start of the current wait (wide_spin_fail)
    unsigned spins = 0;
    const u32 laneOff = 4u * (u32)lane;

    for (int s = slot; s < nstrips; s += W) {
        const long long r0 = kStripRows * s;
        const long long c0l = r0 + g.dmin, c1l = r0 + kStripRows - 1 + g.dmax;
        const int c0 = c0l < 0 ? 0 : (c0l > T ? T : (int)c0l);
        const int c1 = c1l > T - 1 ? T - 1 : (int)c1l;
        if (c0 > c1) break;                                          // below the band at every column: so is everything after it
        const long long n0l = c0l + kStripRows, n1l = c1l + kStripRows;
        const int nextC0 = n0l < 0 ? 0 : (n0l > T ? T : (int)n0l);
        const bool nextLive = s + 1 < nstrips && nextC0 <= (n1l > T - 1 ? T - 1 : (int)n1l);
        const int startCol = nextLive ? nextC0 - 1 : -0x40000000;    // the strip below starts from our bottom score at this column
        const bool hasUp = s > 0;
        const long long u1l = c1l - kStripRows;
        const int upC1 = u1l > T - 1 ? T - 1 : (int)u1l;             // last column of the strip above
        const u64* const upS = sbase + (long long)((s - 1 + W) % W) * per;
        u64* const myS = sbase + (long long)slot * per;
        const u32 tagUp = (u32)s, myTag = (u32)s + 1u;

        const int nwS = (nw - s * 64) < 64 ? (nw - s * 64) : 64;
        const int w = s * 64 + lane;                                 // this lane's 32-row word of the query
        const bool laneOn = lane < nwS;
        const bool tracker = w == nw - 1;                            // lane that owns row m-1
        const u32* const peqRow = reinterpret_cast<const u32*>(a.peq + d.peqOff) + w;   // HBM Peq: [symbol][64-row block] u64

        if (LDSPEQ) {
            __syncthreads();                                         // previous strip done with the slice
            for (int sy = 0; sy < a.sigmaT; ++sy)
                s_peq32[sy * 64 + lane] = laneOn ? peqRow[2LL * sy * nb64] : 0u;
            __syncthreads();
        }

        // ---- score of the row above the strip at column c0 - 1
        int top = (int)r0;                                           // D[r0 - 1][-1] = r0 (edlib.cpp:575-579)
        if (c0 > 0) {
            u64 v = ld_agent(upS);
            waitT0 = 0;
            while ((u32)(v >> 32) != tagUp) {
                if (wide_spin_fail(spins, waitT0, a.wabort)) return;
                v = ld_agent(upS);
            }
            top = (int)(u32)v;
        }
        u32 Pv = ~0u, Mv = 0u;                                       // "+1 per row" (:759-763, 803-808)
        int bscore = top + 32 * (lane + 1);                          // bottom of this lane's word at column c0 - 1
        int sc = top + (m - (int)r0);                                // row m-1 at column c0 - 1 (tracker)
        u32 accP = 0, accM = 0;                                      // the houts of this lane since the last fold, newest at bit 0
        u32 PhOut = 0, Bout = 0;                                     // what travels one lane down per step

        // ---- what lane 0 is fed, 64 columns at a time.  For lane 0's column j the feed is the pair {A, B}: A bit 31 = "the
        // row above delivers +1 at column j", B = {bit 31: "delivers -1", bits 8..: LDS row offset (symbol * 256) of column
        // j + 1}.  A chunk of 64 feeds is built lane-parallel when lane 0 enters it (lane i: column base + i) from the
        // target bytes and the four granules of the strip above, both requested one chunk ahead, and written to an LDS
        // ring; a step then costs lane 0's feed ONE broadcast LDS read instead of readlanes, shifts and ors in the scalar
        // unit (a lone wave issues one instruction at a time, scalar or vector: they were a quarter of the step).
        auto load_t = [&](const int base) -> u32 {                   // lane i: symbol of column base + i + 1
            const int c = base + lane + 1;
            return (c >= 0 && c < T) ? (u32)a.tlut[a.tpool[d.toff + (long long)c * d.tstep]] << 8 : 0u;
        };
        auto h_needed = [&](const int base) -> bool {
            const int gc = base + 16 * (lane & 3);
            return hasUp && gc + 15 >= c0 && gc <= upC1;
        };
        auto load_h = [&](const int base) -> u64 {                   // lanes 4 g + {0..3}: granule of columns base + 16 g ..
            return h_needed(base) ? ld_agent(upS + 1 + (base >> 4) + (lane & 3)) : 0ull;
        };
        auto h_ok = [&](const int base, const u64 v) -> bool { return !h_needed(base) || (u32)(v >> 32) == tagUp; };
        auto h_data = [&](const int base, const u64 v) -> u32 {
            if (!hasUp) return rowAbove;
            const int gc = base + 16 * (lane & 3);
            if (gc > upC1) return 0x0000ffffu;                       // beyond the life of the strip above: +1 per column
            u32 x = (u32)v;
            if (gc + 15 > upC1) {                                    // its last granule: the columns behind upC1 likewise
                const u32 low = (1u << (15 - (upC1 - gc))) - 1u;     // their "+1" bits; the same bits 16 up are their "-1" bits
                x = (x & ~(low | (low << 16))) | low;
            }
            return x;
        };
        bool bail = false;
        u32 tnext = 0; u64 hnext = 0;
        const u32 feedBase = (u32)a.sigmaT * 256u * (LDSPEQ ? 1u : 0u);      // the feed ring follows the Peq slice
        auto feed_at = [&](const u32 byteAddr) -> u64 { return *(const __attribute__((address_space(3))) u64*)(size_t)byteAddr; };
        // lane 0 enters the 64 columns from `base` on (tnext / hnext hold what was requested for them)
        auto enter_chunk = [&](const int base) {
            u64 v = hnext;
            waitT0 = 0;
            while (__builtin_amdgcn_ballot_w64(!h_ok(base, v)) != 0ull) {
                if (wide_spin_fail(spins, waitT0, a.wabort)) { bail = true; break; }
                v = load_h(base);
            }
            const u32 words = h_data(base, v);                       // lanes 4 g + q: the word of group q
            const u32 wd = (u32)__builtin_amdgcn_ds_bpermute(4 * (lane >> 4), (int)words);   // lane i: the word of ITS group
            const u32 pj = (u32)lane & 15u;
            const u32 fa = wd << (16u + pj);
            const u32 fb = ((wd << pj) & 0x80000000u) | tnext;
            *(__attribute__((address_space(3))) u64*)(size_t)(feedBase + 8u * (u32)lane) = ((u64)fb << 32) | fa;
            tnext = load_t(base + 64);
            hnext = load_h(base + 64);
        };
        {
            const int base0 = (c0 - 1) & ~63;                        // the chunk of lane 0's first step (column c0 - 1: symbols only)
            tnext = load_t(base0);
            hnext = load_h(base0);
            enter_chunk(base0);
            if (bail) return;
        }

        const int span = c1 - c0;                                    // lane l is active at steps l .. l + span
        const int nsteps = span + 64;
        u32 eq = 0;                                                  // Peq word of this lane's current column

        // The fold: every 16 columns of lane 63 the houts of all lanes go into their word scores; lane 63's are the
        // granule of those columns (its last granule holds fewer: moved up to their columns' bits)
        auto fold = [&](const int c63) {
            if (nextLive && c63 >= c0 && c63 <= c1 && lane == 63) {
                const int up = 15 - (c63 & 15);
                st_agent(myS + 1 + (c63 >> 4), ((u64)myTag << 32) | ((accP << up) & 0xffffu) | ((accM << up) << 16));
            }
            bscore += __popc(accP) - __popc(accM);
            accP = 0; accM = 0;
        };

        // One step: lane 0 is at column j = c0 + t.  Two words travel one lane down per step: A = the sender's Ph (its
        // bit 31 is hout = +1) and B = {bit 31: the sender's Mh bit 31 (hout = -1), bits 8..: LDS row offset of the
        // receiver's NEXT column's symbol}; lane 0 keeps its feed (the DPP `old` operands).  GENERIC: any step (rotation,
        // publishing, lanes outside their columns); else a step inside a straight-line block.
        auto step = [&](auto genericTag, const int t, const u64 feed) {
            constexpr bool GENERIC = decltype(genericTag)::value;
            const u32 A = (u32)__builtin_amdgcn_update_dpp((int)(u32)feed, (int)PhOut, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
            const u32 Bv = (u32)__builtin_amdgcn_update_dpp((int)(u32)(feed >> 32), (int)Bout, 0x138, 0xf, 0xf, false);
            u32 eqNxt;
            // (the slice is the first LDS object: its address is 0, and an address_space(3) access keeps hipcc from adding it)
            if (LDSPEQ) eqNxt = *(const __attribute__((address_space(3))) u32*)(size_t)((Bv & 0x7fffff00u) | laneOff);
            else eqNxt = laneOn ? peqRow[2LL * ((Bv & 0x7fffffffu) >> 8) * nb64] : 0u;
            const int rel = t - lane;
            if (!GENERIC || (laneOn && (unsigned)rel <= (unsigned)span)) {
                // reference calculateBlock (edlib.cpp:412-447) on a 32-row word
                const u32 hneg = Bv >> 31;
                const u32 eqn = eq | hneg;                            // Eq |= hinIsNeg     (:423)
                const u32 xv = eq | Mv;                               // Xv = Eq | Mv       (:421)
                const u32 sum = (eqn & Pv) + Pv;
                const u32 xh = BITOP3_XOR_OR32(sum, eqn, Pv);         // (:424)
                const u32 ph = BITOP3_OR_NOR32(Mv, xh, Pv);           // (:426)
                const u32 mh = Pv & xh;                               // (:427)
                const u32 phs = __builtin_amdgcn_alignbit(ph, A, 31);     // (ph << 1) | hin > 0   (:435-441)
                const u32 mhs = __builtin_amdgcn_alignbit(mh, Bv, 31);    // (mh << 1) | hin < 0
                Pv = BITOP3_OR_NOR32(mhs, xv, phs);
                Mv = phs & xv;
                PhOut = ph;
                Bout = BITOP3_BFI(0x80000000u, mh, Bv);
                accP = __builtin_amdgcn_alignbit(accP, ph, 31);       // (acc << 1) | hout bit
                accM = __builtin_amdgcn_alignbit(accM, mh, 31);
                if (MODE != 0) {
                    if (tracker) {
                        const int col = c0 + rel;
                        sc += (int)((ph >> sh) & 1u) - (int)((mh >> sh) & 1u);
                        if (sc <= best && col >= d.skip) {           // edlib.cpp:658-673
                            if (sc < best) { best = sc; cnt = 0; }
                            if (cnt < d.posCap) pos[cnt] = col;
                            ++cnt;
                            lastCol = col;
                        }
                    }
                }
            } else {
                Bout = Bv;                                            // every lane forwards the symbol stream
            }
            eq = eqNxt;
            if (GENERIC) {                                            // what lane 63 has finished (uniform in t)
                const int c63 = c0 + t - 63;
                if (c63 == startCol && c63 >= c0 && c63 <= c1) {
                    const int now = bscore + __popc(accP) - __popc(accM);
                    if (lane == 63) st_agent(myS, ((u64)myTag << 32) | (u32)now);
                }
                if ((c63 & 15) == 15 || c63 == c1) fold(c63);
            }
        };
        for (int t = -1; t < nsteps && !bail;) {
            const int j0 = c0 + t;
            if (t >= 0 && (j0 & 63) == 0) { enter_chunk(j0); if (bail) break; }
            const u32 slot0 = feedBase + 8u * ((u32)j0 & 63u);
            // sixteen straight-line steps: every lane inside its columns, lane 63 finishing a granule with the fifteenth,
            // nothing to publish but that granule
            if (t >= 63 && t + 15 <= span && (j0 & 15) == 0 && !(startCol >= j0 - 63 && startCol <= j0 - 48)) {
                const u64 f0 = feed_at(slot0), f1 = feed_at(slot0 + 8), f2 = feed_at(slot0 + 16), f3 = feed_at(slot0 + 24);
                step(std::false_type{}, t, f0);
                const u64 f4 = feed_at(slot0 + 32);
                step(std::false_type{}, t + 1, f1);
                const u64 f5 = feed_at(slot0 + 40);
                step(std::false_type{}, t + 2, f2);
                const u64 f6 = feed_at(slot0 + 48);
                step(std::false_type{}, t + 3, f3);
                const u64 f7 = feed_at(slot0 + 56);
                step(std::false_type{}, t + 4, f4);
                const u64 f8 = feed_at(slot0 + 64);
                step(std::false_type{}, t + 5, f5);
                const u64 f9 = feed_at(slot0 + 72);
                step(std::false_type{}, t + 6, f6);
                const u64 f10 = feed_at(slot0 + 80);
                step(std::false_type{}, t + 7, f7);
                const u64 f11 = feed_at(slot0 + 88);
                step(std::false_type{}, t + 8, f8);
                const u64 f12 = feed_at(slot0 + 96);
                step(std::false_type{}, t + 9, f9);
                const u64 f13 = feed_at(slot0 + 104);
                step(std::false_type{}, t + 10, f10);
                const u64 f14 = feed_at(slot0 + 112);
                step(std::false_type{}, t + 11, f11);
                const u64 f15 = feed_at(slot0 + 120);
                step(std::false_type{}, t + 12, f12);
                step(std::false_type{}, t + 13, f13);
                step(std::false_type{}, t + 14, f14);
                fold(j0 + 14 - 63);
                step(std::false_type{}, t + 15, f15);
                t += 16;
            } else {
                step(std::true_type{}, t, feed_at(slot0));
                t += 1;
            }
        }
        if (bail) return;
        bscore += __popc(accP) - __popc(accM);                       // (what the last fold left)
        if (laneOn && c1 == T - 1 && dumpCol) {                      // stop column of a Hirschberg half: 64-row blocks from two words
            reinterpret_cast<u32*>(a.colP + d.colOff)[w] = Pv;
            reinterpret_cast<u32*>(a.colM + d.colOff)[w] = Mv;
            if ((w & 1) || w == nw - 1) a.colS[d.colOff + (w >> 1)] = bscore;      // (a missing upper half is all zeros)
        }
        if (tracker) {
            if (MODE == 0) {
                if (c1 == T - 1) {
                    // D[m][T] from the bottom score of row m-1's word and the vertical deltas below row m-1 (edlib.cpp:914-917)
                    const u32 below = (sh == 31u) ? 0u : (~0u << (sh + 1));
                    a.outScore[unit] = bscore - __popc(Pv & below) + __popc(Mv & below);
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
        const size_t lds = (size_t)a.sigmaT * 64 * sizeof(u32) + 512;     // Peq slice + the ring of lane 0's feeds
        EDLIB_AMD_CHECK_STATIC_LDS((scan_pairs_wide_kernel<MODE, true>), 0);        // the Peq slice starts at LDS address 0
        hipLaunchKernelGGL((scan_pairs_wide_kernel<MODE, true>), grid, dim3(64), lds, stream, a);
    } else {
        hipLaunchKernelGGL((scan_pairs_wide_kernel<MODE, false>), grid, dim3(64), 512, stream, a);
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
    // the smallest answer over the three instantiations a launch may be (MODE 1 / 2 carry tracker and position state)
    perCu = 1 << 20;
    for (int mode = 0; mode < 3; ++mode) {
        int v = 0;
        hipError_t e;
        if (sigmaT <= 32) {
            const size_t lds = (size_t)sigmaT * 256 + 512;
            e = mode == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, scan_pairs_wide_kernel<0, true>, 64, lds)
              : mode == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, scan_pairs_wide_kernel<1, true>, 64, lds)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, scan_pairs_wide_kernel<2, true>, 64, lds);
        } else {
            e = mode == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, scan_pairs_wide_kernel<0, false>, 64, 512)
              : mode == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, scan_pairs_wide_kernel<1, false>, 64, 512)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, scan_pairs_wide_kernel<2, false>, 64, 512);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
        perCu = v < perCu ? v : perCu;
    }
    if (perCu > 8) perCu = 8;                                        // two waves per SIMD: more only share its issue slots
    if (perCu > 1) perCu -= 1;                                       // margin (the occupancy query can be one block per CU high)
    return cus * perCu;
}

}  // namespace edlib_amd
