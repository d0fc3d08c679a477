// ring32_kernels.hip -- NW pairs on lane rings of 32-ROW WORDS, with the column store and a lane-parallel traceback:
// what batches of short pairs with TASK_PATH run on (BASELINE config 5: 10,000 x 1 kb, path + CIGAR).
//
// Replaces, for such batches, myersCalcEditDistanceNW with a fixed k (reference edlib.cpp:730-928: the band is Ukkonen's,
// what the first / lastBlock bookkeeping of :744-830 converges to), the AlignmentData column store (:22-47, 883-893),
// obtainAlignmentTraceback (:942-1141) and, for the caller-facing arrays, the run-length encoding of
// edlibAlignmentToCigar (:303-350).
//
// Why another kernel next to scan_pairs_ring_kernel (pair_kernels.hip).  A batch of 10,000 pairs is 625 waves of 4-lane
// rings on a chip of 1024 SIMDs: every wave has its SIMD to itself, and a lone wave is bound by the LATENCY of its own
// instruction stream -- about 3.4 ns per dependent instruction whatever it is (tools/lone_wave_ubench.hip) --
// not by issue slots.  Round 4's storing scan took 0.48 us per step there (64-bit blocks: 22 VALU for the update, a
// scalar branch on a ballot every step whose join made hipcc wait for every LDS read it had just issued, block events of
// sixteen units scattered over every second step) and its walk 0.39 us per COLUMN (one lane per unit, ~110 issued
// instructions per column).  Here:
//
//   * lane = one 32-row word (calculateBlock, :412-447, is 11 instructions on a 32-bit word against 22 on a 64-bit pair),
//     ring of G lanes per unit (word b -> ring lane b % G), 64 / G units per wave; the band of threshold K needs
//     K <= 32 (G - 2) (ring32_max_k), or all words on the ring (any K: the whole matrix);
//   * the two words a lane sends its ring neighbour are its raw Ph and Mh: bit 31 IS hout, and v_alignbit_b32 shifts it in
//     as the receiver's hin (no carry extraction, no carry word); a lane outside its word's life sends +1, which is what a
//     word takes from an upstream outside the band (:779) and what row -1 of NW delivers;
//   * the units of a wave share ONE band geometry when their bands fit the ring together (dmin / dmax = the extremes
//     over the wave: a wider band is always exact), so the words of all units start and end at the same steps and the
//     rare-event path of the step is taken for a handful of steps per word, not for every second step;
//   * target symbols come from a pre-mapped symbol pool (target_symbols_kernel) through a 256-slot LDS ring per unit
//     that is refilled 64 columns at a time: the loads are issued at one refill and committed to LDS at the NEXT one, so
//     no step ever waits for global memory;
//   * STORE: the two planes of a block-step (pair_kernels.hpp StoreEntry, on 32 rows: 8 bytes) stay in registers for eight
//     steps and go out as the lane's own 64-byte line of the group ([group of 8 steps][ring lane][step]), through a buffer
//     resource whose out-of-range offsets drop the lines of lanes that were outside their word's life for the whole group
//     (no exec juggling, only the band is written: ~5 of 8 lanes at K = 128) -- and the walk, which follows ONE word
//     through consecutive columns, finds eight of its entries per line (round 5's first layout, [step][lane], had it
//     fetch a 64-byte line per column for 8 bytes of it: 0.67 GB of reads for config 5);
//   * the walk (traceback32_kernel) takes 32 cells of the current DIAGONAL per trip: lane i looks at cell (r - i, c - i),
//     a ballot over "an indel move is possible here" finds the end of the run of diagonal moves, the lanes before it
//     write their MATCH / MISMATCH ops side by side, and the cell that stopped the run is resolved with the reference's
//     preference up > left > diagonal (:1020, 1054, 1085) -- a run of up-moves by one count-leading-ones.  About
//     T / 32 + 2 x (indels) trips instead of T + m cell steps.
#include "pair_kernels.hpp"
#include "lds_check.hpp"

namespace edlib_amd {

typedef unsigned long long u64;
typedef uint32_t u32;

__host__ __device__ static inline int num_blocks(int m) { return (m + 63) >> 6; }
__host__ __device__ static inline int num_words(int m) { return (m + 31) >> 5; }

#define R32_OR_NOR(a, b, c)   ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xf1))   /* a | ~(b | c)  */
#define R32_XOR_OR(a, b, c)   ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xde))   /* (a ^ c) | b   */
#define R32_SEL(m, a, b)      ((u32)__builtin_amdgcn_bitop3_b32((m), (a), (b), 0xca))   /* m ? a : b     */
#define R32_AND_OR(a, b, c)   ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xea))   /* (a & b) | c   */
#define R32_ANDN_OR(a, b, c)  ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0x0e))   /* ~a & (b | c)  */

// Column store of one unit: 8-byte entries in groups of 8 steps, [group][ring lane][step & 7]: a ring lane's eight entries of a
// group are one 64-byte line -- written by that lane alone (a lane outside its word's life for the whole group writes
// nothing), and what the walk reads when it follows a word through consecutive columns.
__host__ __device__ static inline long long ring32_entry(int G, long long t, int ringLane) { return (((t >> 3) * G + ringLane) << 3) + (t & 7); }
long long ring32_store_entries(int G, int qlen, int tlen) { return ((((long long)tlen + num_words(qlen) + 1) >> 3) + 2) * G * 8; }

// LDS words (u32) of one unit's Peq on this kernel: [symbol][rowStride] + a skew of G words, so that the rings of a wave
// (whose lanes hold the same word indices at the same time) start in different banks
static inline int ring32_row_stride(int G, int maxWords) {
    int s = G;
    while (s < maxWords && s < 32) s <<= 1;
    if (maxWords > 32) s = (maxWords + 31) / 32 * 32;
    return s;
}
int ring32_peq_stride(int G, int sigmaT, int maxWords) { return (sigmaT * ring32_row_stride(G, maxWords) + 63) / 64 * 64 + G; }
size_t ring32_lds_bytes(int G, int sigmaT, int maxWords) { return (size_t)(64 / G) * (512 + 4 * (size_t)ring32_peq_stride(G, sigmaT, maxWords)); }

// 32-row word-columns scan_pairs_ring32_kernel computes inside its bands for these units (host; the kernel's own rule:
// one band geometry per wave of 64 / G units when the union of their bands fits the ring)
long long ring32_word_steps(int G, const PairDesc* descs, int n)
{
    const int U = 64 / G, big = 1 << 28;
    long long v = 0;
    for (int w0 = 0; w0 < n; w0 += U) {
        const int w1 = w0 + U < n ? w0 + U : n;
        int lo = big, hi = -big;
        for (int u = w0; u < w1; ++u) {
            const int D = descs[u].tlen - descs[u].qlen, absD = D < 0 ? -D : D, K = descs[u].kinit;
            if (K < absD) continue;
            const int p = (K - absD) >> 1, dmin = (D < 0 ? D : 0) - p, dmax = (D > 0 ? D : 0) + p;
            lo = dmin < lo ? dmin : lo; hi = dmax > hi ? dmax : hi;
        }
        const bool uniform = hi - lo <= 32 * (G - 2);
        for (int u = w0; u < w1; ++u) {
            const int m = descs[u].qlen, T = descs[u].tlen, D = T - m, absD = D < 0 ? -D : D, K = descs[u].kinit;
            if (K < absD) continue;
            const int p = (K - absD) >> 1;
            const long long dmin = uniform ? lo : (D < 0 ? D : 0) - p, dmax = uniform ? hi : (D > 0 ? D : 0) + p;
            for (int b = 0; b < num_words(m); ++b) {
                long long f = 32LL * b + dmin, l = 32LL * b + 31 + dmax;
                f = f < 0 ? 0 : f; l = l > T - 1 ? T - 1 : l;
                if (f <= l) v += l - f + 1;
            }
        }
    }
    return v;
}

// ------------------------------------------------------------------ target symbols

// target byte -> symbol id (row of Peq), once per run over the whole target pool: the scan's refill then needs no lookup
__global__ void __launch_bounds__(256)
target_symbols_kernel(const uint8_t* __restrict__ tpool, const uint8_t* __restrict__ tlut, const long long n, uint8_t* __restrict__ tsym)
{
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = tlut[threadIdx.x];
    __syncthreads();
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 + 16 <= n) {                                              // (the pools are 16-byte aligned and padded)
        const uint4 v = *reinterpret_cast<const uint4*>(tpool + i0);
        const u32 in[4] = {v.x, v.y, v.z, v.w};
        u32 out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            out[k] = (u32)s_lut[in[k] & 0xffu] | ((u32)s_lut[(in[k] >> 8) & 0xffu] << 8) | ((u32)s_lut[(in[k] >> 16) & 0xffu] << 16) | ((u32)s_lut[in[k] >> 24] << 24);
        *reinterpret_cast<uint4*>(tsym + i0) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
        for (long long i = i0; i < n; ++i) tsym[i] = s_lut[tpool[i]];
    }
}

hipError_t launch_target_symbols(const uint8_t* tpool, const uint8_t* tlut, long long n, uint8_t* tsym, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const long long blocks = (n + 4095) / 4096;
    hipLaunchKernelGGL(target_symbols_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tpool, tlut, n, tsym);
    return hipGetLastError();
}

// ------------------------------------------------------------------ the scan

// ring neighbour's value: lane rl takes it from ring lane rl - 1, ring lane 0 from ring lane G - 1
template <int G> __device__ __forceinline__ u32 ring32_rot(const u32 v, const bool first)
{
    if constexpr (G == 16) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x121 /*row_ror:1*/, 0xf, 0xf, true);
    else if constexpr (G == 4) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x93 /*quad_perm:[3,0,1,2]*/, 0xf, 0xf, true);
    else {                                                           // G == 8: two rings per DPP row
        const u32 a = (u32)__builtin_amdgcn_mov_dpp((int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
        const u32 b = (u32)__builtin_amdgcn_mov_dpp((int)v, 0x107 /*row_shl:7*/, 0xf, 0xf, true);
        return first ? b : a;
    }
}

template <typename T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const T o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const T o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// NW inside Ukkonen's band of threshold K = desc.kinit (exact iff the result is <= K; any K when all words sit on the
// ring).  Forward units only (tstep = +1; the target is read through a.tsym).  STORE: the two planes of every word-step
// inside the band at entry storeOff + ring32_entry(G, t, ring lane) of the u64 view of a.store, t = column + word.
template <int G, bool STORE>
__global__ void __launch_bounds__(64)
scan_pairs_ring32_kernel(const PairScanArgs a, const int rowStride, const int peqStride)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_dyn32[];     // [U] target rings of 256 u16, then [U] Peq tables
    constexpr int U = 64 / G, NP = 64 / G;
    const int lane = threadIdx.x;
    const int rl = lane % G, uw = lane / G;
    const bool firstLane = rl == 0;
    unsigned short* const s_tgt = reinterpret_cast<unsigned short*>(s_dyn32) + uw * 256;
    u32* const s_peq = s_dyn32 + U * 128 + uw * peqStride;
    // LDS byte addresses (the rings start at LDS address 0: no static LDS in this kernel, checked on the host)
    const u32 tbase = (u32)uw * 512u;
    const u32 peqBase = (u32)(U * 512) + 4u * (u32)(uw * peqStride);
    auto lds_u16 = [](const u32 addr) -> u32 { return *(const __attribute__((address_space(3))) unsigned short*)(size_t)addr; };
    auto lds_u32 = [](const u32 addr) -> u32 { return *(const __attribute__((address_space(3))) u32*)(size_t)addr; };

    const int unit = blockIdx.x * U + uw;
    const bool have = unit < a.numUnits;
    const PairDesc* const dp = a.descs + (have ? unit : 0);
    const int m = dp->qlen, T = dp->tlen, K = dp->kinit;
    const int nw = num_words(m), nb64 = num_blocks(m);
    const int D = T - m, absD = D < 0 ? -D : D;
    const bool active = have && K >= absD;                           // else no path of cost <= K exists (edlib.cpp:749-754)
    if (have && !active && firstLane) { a.outScore[unit] = 0x3fffffff; a.outCount[unit] = 0; a.outLast[unit] = -1; }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;
    int dmin, dmax;
    {
        const int p = (K - absD) >> 1;
        dmin = (D < 0 ? D : 0) - p; dmax = (D > 0 ? D : 0) + p;
        // one geometry for the whole wave when the union of the units' bands still fits the ring: every unit then starts and
        // ends its words at the same steps (a wider band is always exact: cells outside only enter as upper bounds)
        const int big = 1 << 28;
        const int lo = wave_min(active ? dmin : big), hi = wave_max(active ? dmax : -big);
        if (hi - lo <= 32 * (G - 2)) { dmin = lo; dmax = hi; }
    }
    const u32 sh = (u32)(m - 1) & 31u;                               // row m-1 inside the last word
    const int nwA = active ? nw : 0;

    // ---- target ring: s_tgt[col & 255] = byte offset of the column's Peq row.  Loads are issued one refill ahead of their use.
    const u32 symScale = 4u * (u32)rowStride;
    const uint8_t* const tsym = a.tsym + dp->toff;
    int written = 0;                                                  // columns [0, written) have been committed to the ring
    u32 pend[NP];                                                     // symbols of columns [written, written + 64), in flight
    auto issue = [&]() {
#pragma unroll
        for (int k = 0; k < NP; ++k) { const int c = written + rl + G * k; pend[k] = (active && c < T) ? (u32)tsym[c] : 0u; }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NP; ++k) { const int c = written + rl + G * k; s_tgt[c & 255] = (unsigned short)(pend[k] * symScale); }
        written += 64;
    };
    for (int i = rl; i < 256; i += G) s_tgt[i] = 0;                   // never index Peq with an unwritten slot
    issue(); commit(); issue(); commit(); issue();                    // 128 columns written, 64 more on their way
    // the unit's whole Peq: [symbol][rowStride] words of 32 rows (the pool holds 64-bit blocks: two words each)
    {
        const u32* const pool = reinterpret_cast<const u32*>(a.peq + dp->peqOff);
        for (int sy = 0; sy < a.sigmaT; ++sy)
            for (int i = rl; i < rowStride; i += G)
                s_peq[sy * rowStride + i] = (active && i < 2 * nb64) ? pool[2LL * sy * nb64 + i] : 0u;
    }

    // ---- per-lane word bookkeeping: word b is updated at steps tstart .. tstart + span (column + word = step)
    int b = rl;
    auto first_col = [&](int w) { const int c = 32 * w + dmin; return c < 0 ? 0 : c; };
    auto last_col = [&](int w) { const int c = 32 * w + 31 + dmax; return c > T - 1 ? T - 1 : c; };
    const int never = 0x3fffffff;
    int ev, span = 0;                                                 // steps until this lane's next event; life of its word
    auto arm = [&](const int t) {
        const bool ok = b < nwA && first_col(b) <= last_col(b);       // words below the band never get a column
        const int tstart = ok ? first_col(b) + b : never;
        span = ok ? last_col(b) + b - tstart : 0;
        ev = ok ? tstart - t : never;
    };
    arm(0);
    u32 actm = 0u;                                                    // all ones while the lane is inside its word's life
    // word 0 takes row -1 (+1 per column: edlib.cpp:779) whatever its ring neighbour sends (a ring that holds all words of
    // its unit: the last word's lane feeds lane 0)
    u32 xmask = (b == 0) ? 0u : ~0u, xfix = (b == 0) ? 0x80000000u : 0u;
    u32 Pv = ~0u, Mv = 0u;
    u32 PhOut = 0x80000000u, MhOut = 0u;                              // what the ring neighbour takes next step: bit 31 = hout +1 / -1
    u32 accP = 0u, accM = 0u;                                         // the houts sent since the last fold, newest at bit 0
    int bscore = 0;                                                   // bottom score of the lane's word after the last fold
    auto fold = [&]() { bscore += __popc(accP) - __popc(accM); accP = 0u; accM = 0u; };
    u32 peqLane = peqBase + 4u * (u32)b;                              // LDS address of this lane's word in Peq row 0
    u32 c2 = 0;                                                       // twice the column of the NEXT row-offset fetch (col + 2)
    u32 eqA = 0, eqB = 0, offA = 0, offB = 0;

    int nsteps = active ? T + nw : 0;                                 // one step past the last word's last: its closing event
    nsteps = wave_max(nsteps);
    const int nstepsU = __builtin_amdgcn_readfirstlane(nsteps);

    // column store: 8 bytes per word-step through a buffer resource; a lane outside its word's life stores out of range
    // (the caller keeps the store of a launch below 4 GB - 16: a 32-bit byte offset per lane; 0xffffffff is out of range)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(a.store), 0, (int)0xfffffff0u, 0x00020000);
    u32 soff = STORE ? (u32)(8ull * ((u64)dp->storeOff + 8ull * (u64)rl)) : 0u, dead = ~0u;
    u32 gdead = ~0u;                                                  // all ones while the lane has been outside its word's life for the whole group
    u32 px[8], py[8];                                                 // the planes of the group's eight steps

    // The rare part of a step: some lane's word has just finished (its last update was step t - 1) or starts now.
    auto events = [&](const int t, const u32 A, const u32 Bm, u32& eqCur, u32& offCur) {
        fold();
        const int upScore = (int)ring32_rot<G>((u32)bscore, firstLane);   // upstream's bottom score after step t - 1
        if (ev == 0 && actm) {                                        // ---- closing word b
            const int colLast = t - 1 - b;
            if (colLast == T - 1 && b == nw - 1) {
                // D[m][T] from the bottom score of row m-1's word and the vertical deltas below row m-1 (edlib.cpp:914-917)
                const u32 below = (sh == 31u) ? 0u : (~0u << (sh + 1));
                a.outScore[unit] = bscore - __popc(Pv & below) + __popc(Mv & below);
                a.outCount[unit] = 1; a.outLast[unit] = T - 1;
            }
            actm = 0u; dead = ~0u;
            b += G;
            xmask = ~0u; xfix = 0u;
            arm(t);                                                   // ev == 0 again when word b + G starts right now
        }
        if (ev == 0 && !actm) {                                       // ---- word b starts with this step
            const int col = t - b;
            Pv = ~0u; Mv = 0u;                                        // "+1 per row" (edlib.cpp:759-763, 803-808)
            // bottom of the word above at column col - 1: upstream's bottom after its step minus its delta at `col`
            // (a sender outside its word's life extrapolates by +1 per column, the value its receivers assume)
            const int above = (col == 0) ? 32 * b : upScore - ((int)(A >> 31) - (int)(Bm >> 31));
            bscore = above + 32;
            peqLane = peqBase + 4u * (u32)b;
            eqCur = lds_u32(peqLane + (u32)s_tgt[col & 255]);
            offCur = s_tgt[(col + 1) & 255];
            c2 = 2u * (u32)(col + 2);
            actm = ~0u; dead = 0u;
            ev = span + 1;                                            // closes at the top of the step after its last
        }
    };

    // One step.  eqCur / offCur: Peq word of this step's column and row offset of the next column (fetched by the previous
    // step); eqNxt / offNxt are fetched here for the next step -- the caller swaps the two register sets every step.
    auto step = [&](const int t, u32& eqCur, u32& eqNxt, u32& offCur, u32& offNxt, u32& outX, u32& outY) {
        const u32 A = ring32_rot<G>(PhOut, firstLane), Bm = ring32_rot<G>(MhOut, firstLane);
        if (__builtin_amdgcn_ballot_w64(ev == 0) != 0ull) events(t, A, Bm, eqCur, offCur);
        --ev;
        eqNxt = lds_u32(peqLane + offCur);
        offNxt = lds_u16(tbase | (c2 & 511u));
        c2 += 2u;
        const u32 Ain = R32_AND_OR(A, xmask, xfix), Bin = Bm & xmask;
        // reference calculateBlock (edlib.cpp:412-447) on a 32-row word
        const u32 hneg = Bin >> 31;
        const u32 eqn = eqCur | hneg;                                 // Eq |= hinIsNeg     (:423)
        const u32 xv = eqCur | Mv;                                    // Xv = Eq | Mv       (:421)
        const u32 sum = (eqn & Pv) + Pv;
        const u32 xh = R32_XOR_OR(sum, eqn, Pv);                      // (:424)
        const u32 ph = R32_OR_NOR(Mv, xh, Pv);                        // (:426)
        const u32 mh = Pv & xh;                                       // (:427)
        const u32 phs = __builtin_amdgcn_alignbit(ph, Ain, 31);       // (ph << 1) | hin > 0   (:435-441)
        const u32 mhs = __builtin_amdgcn_alignbit(mh, Bin, 31);       // (mh << 1) | hin < 0
        Pv = R32_OR_NOR(mhs, xv, phs);
        Mv = phs & xv;
        PhOut = R32_SEL(actm, ph, 0x80000000u);                       // outside the word's life: hout = +1
        MhOut = mh & actm;
        accP = __builtin_amdgcn_alignbit(accP, PhOut, 31);            // (acc << 1) | hout bit
        accM = __builtin_amdgcn_alignbit(accM, MhOut, 31);
        if constexpr (STORE) {
            // the planes of pair_kernels.hpp StoreEntry on 32 rows: x = Pv | Ph, y = ~Pv & (Ph | Xh)
            outX = Pv | ph; outY = R32_ANDN_OR(Pv, ph, xh);
            gdead &= dead;
        }
    };
    // the lane's line of the group that has just ended: eight entries = 64 bytes, four 16-byte stores (the line of a lane
    // that was dead for the whole group has an out-of-range offset: dropped)
    auto flush = [&]() {
        if constexpr (STORE) {
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const int voff = (int)(soff | gdead);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32x4 v; v.x = px[2 * k]; v.y = py[2 * k]; v.z = px[2 * k + 1]; v.w = py[2 * k + 1];
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 16 * k, 0);
            }
            soff += 64u * (u32)G;
            gdead = ~0u;
        }
    };

    for (int t = 0; t <= nstepsU; t += 8) {
        if ((t & 31) == 0) {
            fold();                                                   // (the accumulators hold 32 steps)
            if ((t & 63) == 0 && t > 0) {
                // pace the ring by the largest column in use: lowest word alive at step t, its column
                int bt = t - 31 - dmax; bt = bt <= 0 ? 0 : (bt + 32) / 33;
                if (t - T + 1 > bt) bt = t - T + 1;
                const int jmax = t - bt;
                if (active && written < jmax + 66 + 64) { commit(); issue(); }
            }
        }
        step(t, eqA, eqB, offA, offB, px[0], py[0]);
        step(t + 1, eqB, eqA, offB, offA, px[1], py[1]);
        step(t + 2, eqA, eqB, offA, offB, px[2], py[2]);
        step(t + 3, eqB, eqA, offB, offA, px[3], py[3]);
        step(t + 4, eqA, eqB, offA, offB, px[4], py[4]);
        step(t + 5, eqB, eqA, offB, offA, px[5], py[5]);
        step(t + 6, eqA, eqB, offA, offB, px[6], py[6]);
        step(t + 7, eqB, eqA, offB, offA, px[7], py[7]);
        flush();
    }
}

template <int G>
static hipError_t launch_ring32_t(bool store, const PairScanArgs& a, int maxWords, hipStream_t stream)
{
    constexpr int U = 64 / G;
    const int rowStride = ring32_row_stride(G, maxWords), peqStride = ring32_peq_stride(G, a.sigmaT, maxWords);
    const size_t lds = ring32_lds_bytes(G, a.sigmaT, maxWords);
    const dim3 grid((a.numUnits + U - 1) / U);
    if (store) {
        EDLIB_AMD_CHECK_STATIC_LDS((scan_pairs_ring32_kernel<G, true>), 0);
        hipLaunchKernelGGL((scan_pairs_ring32_kernel<G, true>), grid, dim3(64), lds, stream, a, rowStride, peqStride);
    } else {
        EDLIB_AMD_CHECK_STATIC_LDS((scan_pairs_ring32_kernel<G, false>), 0);
        hipLaunchKernelGGL((scan_pairs_ring32_kernel<G, false>), grid, dim3(64), lds, stream, a, rowStride, peqStride);
    }
    return hipGetLastError();
}

hipError_t launch_scan_pairs_ring32(int G, bool store, const PairScanArgs& a, int maxWords, hipStream_t stream)
{
    if (a.numUnits == 0) return hipSuccess;
    if (!a.tsym || (store && !a.store) || ring32_lds_bytes(G, a.sigmaT, maxWords) > 48 * 1024) return hipErrorInvalidValue;
    switch (G) {
        case 4: return launch_ring32_t<4>(store, a, maxWords, stream);
        case 8: return launch_ring32_t<8>(store, a, maxWords, stream);
        case 16: return launch_ring32_t<16>(store, a, maxWords, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------ the walk

// reference obtainAlignmentTraceback (edlib.cpp:942-1141) on the store of scan_pairs_ring32_kernel, L lanes per unit.
// Position (r, c); lane i of the unit looks at cell (r - i, c - i) of the current diagonal.  "x" of a cell says that an
// indel move is possible there (up: D[r][c] = D[r-1][c] + 1, or left: D[r][c] = D[r][c-1] + 1; pair_kernels.hpp
// StoreEntry): the cells before the first such cell take the diagonal, MATCH iff their "y" bit; the cell that stopped the
// run takes up-moves while the "up" plane (x & ~y) says so -- one count-leading-ones -- else one left move (the
// reference's preference: :1020, 1054, 1085).  Reaching row -1 / column -1 leaves a tail of DELETEs / INSERTs
// (:1040-1046, 1070-1078).  Ops are written back to front into the END of the unit's slot, as traceback_kernel does.
template <int L>
__global__ void __launch_bounds__(64)
traceback32_kernel(const TracebackArgs a, const int G)
{
    constexpr int U = 64 / L;
    const int lane = threadIdx.x, sub = lane % L, base = lane - sub;
    const int unit = blockIdx.x * U + lane / L;
    const bool have = unit < a.numUnits;
    const PairDesc d = a.descs[have ? unit : 0];
    const int m = d.qlen, T = d.tlen;
    const long long slotAt = a.opsOff[have ? unit : 0];
    uint8_t* const ops = a.ops + slotAt;
    const int slot = (int)(a.opsOff[have ? unit + 1 : 1] - slotAt);
    const u64* const S = reinterpret_cast<const u64*>(a.store) + d.storeOff;
    // a unit whose scan ended above its threshold has no exact cells to walk on (it is rescanned at the next level)
    const bool skip = !have || a.score[unit] > d.kinit;
    int r = m - 1, c = T - 1, w = slot;
    bool done = skip;
    const u64 unitMask = L == 64 ? ~0ull : ((1ull << L) - 1ull);
    while (__builtin_amdgcn_ballot_w64(!done) != 0ull) {
        // ---- the diagonal from (r, c) upwards: lane i at (r - i, c - i)
        const int ri = r - sub, ci = c - sub;
        const bool valid = !done && ri >= 0 && ci >= 0;
        u64 e = 0;
        if (valid) { const int wi = ri >> 5; e = S[ring32_entry(G, (long long)ci + wi, wi & (G - 1))]; }
        const u32 xw = (u32)e, yw = (u32)(e >> 32);
        const u32 bit = (u32)ri & 31u;
        const bool xb = (xw >> bit) & 1u, yb = (yw >> bit) & 1u;
        const u64 stops = (__builtin_amdgcn_ballot_w64(!valid || xb) >> base) & unitMask;
        const int j = stops ? __builtin_ctzll(stops) : L;             // diagonal moves before the first stop
        if (!done && sub < j) ops[w - 1 - sub] = yb ? (uint8_t)0 : (uint8_t)3;      // MATCH / MISMATCH
        // the planes of the cell that stopped the run (lane j's word): everybody takes part in the exchange
        const int srcLane = base + (j < L ? j : 0);
        const u32 X = (u32)__shfl((int)xw, srcLane, 64), Y = (u32)__shfl((int)yw, srcLane, 64);
        if (!done) {
            r -= j; c -= j; w -= j;
            if (j < L) {
                if (r < 0 || c < 0) {                                 // the matrix boundary: the rest is one run
                    const int cnt = c < 0 ? r + 1 : c + 1;            // (both negative: nothing left)
                    const uint8_t op = c < 0 ? (uint8_t)1 : (uint8_t)2;       // INSERT / DELETE
                    for (int i = sub; i < cnt; i += L) ops[w - 1 - i] = op;
                    w -= cnt > 0 ? cnt : 0;
                    done = true;
                } else {
                    const u32 bb = (u32)r & 31u;
                    const u32 up = X & ~Y;
                    const u32 nx = ~(up << (31u - bb));                // row r at bit 31; the zeros shifted in end the run
                    const int ups = nx ? __builtin_clz(nx) : 32;      // leading ones: <= bb + 1
                    if (ups > 0) {                                    // INSERTs
                        if (sub < ups) ops[w - 1 - sub] = (uint8_t)1;
                        r -= ups; w -= ups;
                    } else {                                          // DELETE (x set, not up: Ph)
                        if (sub == 0) ops[w - 1] = (uint8_t)2;
                        c -= 1; w -= 1;
                    }
                    if (r < 0 || c < 0) {                             // (the same boundary rule, now)
                        const int cnt = (r < 0 && c < 0) ? 0 : (c < 0 ? r + 1 : c + 1);
                        const uint8_t op = c < 0 ? (uint8_t)1 : (uint8_t)2;
                        for (int i = sub; i < cnt; i += L) ops[w - 1 - i] = op;
                        w -= cnt;
                        done = true;
                    }
                }
            }
        }
    }
    if (have && sub == 0) a.opsLen[unit] = skip ? 0 : slot - w;
}

hipError_t launch_traceback32(const TracebackArgs& a, int G, hipStream_t stream)
{
    if (a.numUnits == 0) return hipSuccess;
    if (G != 4 && G != 8 && G != 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL((traceback32_kernel<32>), dim3((a.numUnits + 1) / 2), dim3(64), 0, stream, a, G);
    return hipGetLastError();
}

}  // namespace edlib_amd
