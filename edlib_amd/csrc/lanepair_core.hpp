// lanepair_core.hpp -- the lane-per-PAIR banded NW distance scan (round 6): what one LANE does.
//
// What it replaces: myersCalcEditDistanceNW (edlib.cpp:730-928) with a fixed threshold K -- the block band of
// edlib.cpp:744-830 (first/lastBlock follow the scores, :799-830), the carry chain of :781-785, the decode of the
// last block :914-917 -- for big batches of independent pairs over at most four target symbols (config 4).
//
// The lane rings of pair_kernels.hip spend a wave's lanes on the ROWS of a few units: a carry crosses lanes every step,
// every lane fetches its Peq word from LDS, a countdown finds block starts and ends (38.6 VALU per 64-row block-step).
// Here a lane owns a UNIT, like the lanes of kernel A own reads:
//   * the band of the column is a window of NA <= W words of 32 rows in the lane's registers (Pv, Mv and the query as
//     two bit planes Q0, Q1 per word), updated top-down with one add-carry chain (the whole window is one long Myers
//     word: no hin / hout between words, no cross-lane traffic);
//   * Eq is SYNTHESISED: the lane's target symbol of the column as two all-ones / all-zeros masks s0, s1 gives
//     ~Eq = (Q1 ^ s1) | (Q0 ^ s0) -- two v_bitop3: no Peq, no LDS;
//   * the window follows the diagonal: every 32 columns (at the same column for all lanes) the words move up one
//     register and a fresh word enters below as "+1 per row" (the reference's new block, edlib.cpp:803-808).  The row
//     the window starts on is PRIVATE to the lane (top = 32 * (c / 32) - dmax of ITS band): lanes never meet, so nothing
//     makes them share rows; only the slide's timing and the number of words NA are wave-uniform;
//   * rows above the matrix are virtual cells D[r][c] = c - r (vertical delta -1, horizontal +1): the recurrence keeps
//     them so whatever Eq says, and row -1 comes out as the NW boundary D[-1][c] = c + 1;
//   * the band NARROWS with the scores like the reference's (edlib.cpp:812-830): a cell is dead when its computed value
//     plus its diagonal distance to the end cell exceeds K; dead cells above the target diagonal stay dead along every
//     higher diagonal, below it along every lower one (a vertical / horizontal step costs 2 of slack).  Every 32 columns
//     the lane tests the bottom cell of word 1 and of word NA - 2; when EVERY lane of the wave agrees the window drops
//     its top word (an extra slide) or does not take the fresh bottom word.  At config 4 (K ~ 1280, distance ~ 1136) the
//     window falls from 41 words to ~8 over the scan: about half the word-steps of the static band.
//
// The result is exact iff it is <= K (Ukkonen: cells outside the window only enter as upper bounds).
//
// This header compiles for the device (lanepair_kernels.hip) AND for the host (tests/lanepair_host.cpp: the same code, one
// lane at a time, checked against the reference on the CPU: tests/test_lanepair_model.py).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include "lanepair_asm.hpp"
#endif

#if defined(__HIPCC__)
#define LP_FN __host__ __device__ __forceinline__
#else
#define LP_FN static inline __attribute__((always_inline))
#endif

namespace edlib_amd {
namespace lanepair {

typedef uint32_t u32;

#if !defined(LANEPAIR_HAVE_PLANES)
struct Plane2 { u32 q0, q1; };        // 32 query rows: bit i of q0 / q1 = low / high bit of the symbol code of row 32 w + i
struct Tgt2 { u32 t0, t1; };          // 32 target columns: bit j of t0 / t1 = low / high bit of the symbol code of column 32 b + j
#endif

// ({hi, lo} >> sh) & 0xffffffff, sh in 0..31
LP_FN u32 lp_alignbit(u32 hi, u32 lo, u32 sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (u32)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u));
#endif
}
LP_FN u32 lp_addc(u32 a, u32 b, u32 cin, u32& cout)
{
#if defined(__HIP_DEVICE_COMPILE__)
    u32 co;
    const u32 s = __builtin_addc(a, b, cin, &co);
    cout = co;
    return s;
#else
    const uint64_t s = (uint64_t)a + b + cin;
    cout = (u32)(s >> 32);
    return (u32)s;
#endif
}
LP_FN int lp_popc(u32 x) { return __builtin_popcount(x); }

// wave-uniform decisions: the device overrides these (ballot over the wave); the host model is one lane
#if defined(__HIP_DEVICE_COMPILE__)
LP_FN bool lp_all(bool p) { return __all(p) != 0; }
LP_FN int lp_wave_max(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; }
    return __builtin_amdgcn_readfirstlane(v);
}
LP_FN int lp_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
LP_FN bool lp_all(bool p) { return p; }
LP_FN int lp_wave_max(int v) { return v; }
LP_FN int lp_uniform(int v) { return v; }
#endif

template <int W>
struct Window {
    u32 Pv[W], Mv[W], Q0[W], Q1[W];
};

// v_bitop3_b32: bit (a*4 + b*2 + c) of the immediate is f(a, b, c)
#if defined(__HIP_DEVICE_COMPILE__)
#define LP_BITOP3(a, b, c, imm) __builtin_amdgcn_bitop3_b32((a), (b), (c), (imm))
// (the column itself: lanepair_asm.hpp -- one asm statement per window height, and why)
#define LP_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0f70)          /* s_waitcnt vmcnt(0) */
#define LP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
LP_FN u32 lp_bitop3_host(u32 a, u32 b, u32 c, u32 imm)
{
    u32 r = 0;
    for (int k = 0; k < 8; ++k)
        if ((imm >> k) & 1u) r |= ((k & 4) ? a : ~a) & ((k & 2) ? b : ~b) & ((k & 1) ? c : ~c);
    return r;
}
#define LP_BITOP3(a, b, c, imm) lp_bitop3_host((a), (b), (c), (imm))
#define LP_WAIT_LOADS() do { } while (0)
#define LP_SCHED_FENCE() do { } while (0)
#endif

// One column over the NA active words, top-down.  hin at the window's top is +1: row -1 of NW while the window still
// starts above the matrix, a cell outside the band afterwards (edlib.cpp:779).  12 VALU ops per word (9 full rate + the
// add-carry and two v_alignbit at half rate).  On the device the column is ONE asm statement per window height
// (lanepair_asm.hpp: VOP3 encodings behind an alignment fence, the words software-pipelined three deep -- a wave has one
// neighbour on its SIMD at most, so the ~7-cycle dependent-issue latency is its own to hide; word by word, as the recurrence
// reads and as the host path below spells it, the first build ran at 67-90 SIMD cycles per word-column against the 30 its
// instructions take: tools/lanepair_ubench.hip).
template <int W, int NA>
LP_FN void lp_column(Window<W>& L, const u32 t0, const u32 t1, int& sb)
{
    static_assert(NA >= 3 && NA <= 48, "the pipeline of lp_column is three words deep; lanepair_asm.hpp spells heights up to 48");
#if defined(__HIP_DEVICE_COMPILE__)
    u32 s0, s1, x, t, sv, xh, phs, mhs, xv, ne0, ne1, ne2, ph0, ph1, ph2, mh0, mh1, mh2;
    unsigned long long cy;
    LP_COLUMN_DISPATCH(NA)
#else
    // the same recurrence, word by word (the host model of tests/lanepair_host.cpp)
    const u32 s0 = 0u - (t0 & 1u), s1 = 0u - (t1 & 1u);
    u32 carry = 0, php = ~0u, mhp = 0u, Phl = 0, Mhl = 0;
    for (int i = 0; i < NA; ++i) {
        const u32 ne = (L.Q1[i] ^ s1) | (L.Q0[i] ^ s0);
        const u32 pv = L.Pv[i], mv = L.Mv[i];
        const u32 tt = pv & ~ne;
        const uint64_t w = (uint64_t)tt + pv + carry;
        const u32 s = (u32)w; carry = (u32)(w >> 32);
        const u32 Xh = (s ^ pv) | ~ne;
        const u32 Ph = mv | ~(Xh | pv), Mh = pv & Xh;
        const u32 ph = lp_alignbit(Ph, php, 31), mh = lp_alignbit(Mh, mhp, 31);       // the window's top takes hin = +1
        php = Ph; mhp = Mh; Phl = Ph; Mhl = Mh;
        const u32 Xv = ~ne | mv;
        L.Pv[i] = mh | ~(Xv | ph);
        L.Mv[i] = ph & Xv;
    }
    sb += (int)(Phl >> 31) - (int)(Mhl >> 31);         // the bottom row of the window moves with its horizontal delta
#endif
}

// the words move up one register (the top word leaves)
template <int W, int NA>
LP_FN void lp_shift_up(Window<W>& L)
{
#if defined(__HIP_DEVICE_COMPILE__)
    LP_SHIFT_DISPATCH(NA)
#else
    for (int i = 0; i + 1 < NA; ++i) { L.Pv[i] = L.Pv[i + 1]; L.Mv[i] = L.Mv[i + 1]; L.Q0[i] = L.Q0[i + 1]; L.Q1[i] = L.Q1[i + 1]; }
#endif
}

// What a lane carries besides its window.
struct LaneCtl {
    int m, T, K, delta;       // query / target length, threshold, T - m
    int top;                  // row of bit 0 of word 0
    int sb;                   // computed value of the window's bottom row (bit 31 of word na - 1) after the last column done
    int score;                // D[m-1][T-1] as computed, set in the block of the lane's last column
    int wi;                   // index of the plane word that holds the row below the window's bottom
    u32 sh;                   // bit offset of the window's rows inside the plane words (constant: top moves by 32)
    Plane2 lo;                // plane word wi
    u32 planeOff, tgtOff;     // the unit's first Plane2 / Tgt2 in the pools
};

LP_FN int lp_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// D[m-1][T-1] from the window of the lane's last column: the bottom value minus the vertical deltas below row m - 1
// (the reference decodes its last block the same way, edlib.cpp:914-917)
template <int W, int NA>
LP_FN int lp_final_score(const Window<W>& L, const LaneCtl& s)
{
    const int rel = (s.m - 1) - s.top;
    int score = s.sb;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int lowbit = rel - 32 * j;
        u32 mask;
        if (lowbit < 0) mask = ~0u; else if (lowbit >= 31) mask = 0u; else mask = ~0u << (u32)(lowbit + 1);
        score -= lp_popc(L.Pv[j] & mask) - lp_popc(L.Mv[j] & mask);
    }
    return score;
}

// 32 columns (c0 .. c0 + 31; a lane stops at its own T), then the dead tests, the slide and the trims.
// Returns the number of active words of the next block.  `more` = this lane has columns beyond the block.
template <int W, int NA>
LP_FN int lp_block(Window<W>& L, LaneCtl& s, Tgt2& tg, const Tgt2* tgt, const int b, const Plane2* planes, const u32 deny)
{
    const int c0 = 32 * b;
    // what the previous block requested has had 32 columns to arrive; waiting HERE keeps every s_waitcnt (4 bytes) out of
    // the column loop (lanepair_asm.hpp: the phase of the code).  Then this block's requests: the target
    // planes of the next 32 columns and the plane word the slide may take.
    LP_WAIT_LOADS();
    // (`planes` / `tgt` are the POOLS -- wave-uniform, scalar registers -- and the lane's own 32-bit offsets into them sit in
    // LaneCtl; the unit's word counts are recomputed from m and T: every register that lives through the scan counts)
    const int nblkOwn = s.T > 0 ? (s.T + 31) >> 5 : 1, nplanes = (s.m + 31) >> 5;
    const Tgt2 tnext = tgt[s.tgtOff + (u32)(b + 1 < nblkOwn ? b + 1 : nblkOwn - 1)];
    const Plane2 hiPre = planes[s.planeOff + (u32)lp_clampi(s.wi + 1, 0, nplanes - 1)];
    u32 t0 = tg.t0, t1 = tg.t1;
#pragma unroll 1
    for (int j = 0; j < 32; ++j) {
        if (c0 + j < s.T) lp_column<W, NA>(L, t0, t1, s.sb);
        t0 >>= 1; t1 >>= 1;
    }
    const int cnext = c0 + 32;
    const bool more = s.T > cnext;
    // a lane whose last column was in this block reads its answer off the window now: later stages hold fewer words
    // (and only the words of THIS stage are named here, so that a word dies with its stage)
    if (!more && s.T > c0) s.score = lp_final_score<W, NA>(L, s);
    // ---- dead tests on the state after column cnext - 1 (window not yet moved): the bottom cell of word j is on diagonal
    // dmax - 32 j, dmax = c0 - top.  S_j = its computed value = sb minus the vertical deltas of the words below.
    bool dropTop = false, dropBottom = false;
    if (NA >= 4) {
        int acc = s.sb - (lp_popc(L.Pv[NA - 1]) - lp_popc(L.Mv[NA - 1]));      // S_{NA-2}
        const int dmax = c0 - s.top;
        const int dl = dmax - 32 * (NA - 2);
        const bool lowDead = dl <= s.delta && acc + (s.delta - dl) > s.K;
#pragma unroll
        for (int j = NA - 2; j >= 2; --j) acc -= lp_popc(L.Pv[j]) - lp_popc(L.Mv[j]);   // S_1
        const int du = dmax - 32;
        const bool upDead = du >= s.delta && acc + (du - s.delta) > s.K;
        dropBottom = lp_all(!more || lowDead) && !(deny & 1u);
        dropTop = NA - (dropBottom ? 1 : 0) >= 4 && lp_all(!more || upDead) && !(deny & 2u);     // (never below three words)
    }
    // ---- the slide: words up one register, the fresh word below (edlib.cpp:803-808: P = ~0, M = 0) unless it is dead
    int na = NA;
    if (more) {
        lp_shift_up<W, NA>(L);
        s.top += 32;
    }
    if (!dropBottom) {
        if (more) {
            const Plane2 hi = hiPre;                     // plane word wi + 1, requested before the block
            L.Pv[NA - 1] = ~0u; L.Mv[NA - 1] = 0u;
            L.Q0[NA - 1] = lp_alignbit(hi.q0, s.lo.q0, s.sh);
            L.Q1[NA - 1] = lp_alignbit(hi.q1, s.lo.q1, s.sh);
            s.lo = hi; s.wi += 1;
            s.sb += 32;
        }
    } else {
        na -= 1;
    }
    if (dropTop) {
        if (more) {
            lp_shift_up<W, NA>(L);          // (word NA - 1 moves too: it is the fresh word when the bottom stayed)
            s.top += 32;
        }
        na -= 1;
    }
    tg = tnext;
    return na;
}

// the lane's band for threshold K (Ukkonen): diagonals [min(0, delta) - p, max(0, delta) + p], p = (K - |delta|) / 2
LP_FN int lp_band_words(int m, int T, int K)
{
    const int delta = T - m, ad = delta < 0 ? -delta : delta;
    if (ad > K || T <= 0 || m <= 0) return 0;
    const int p = (K - ad) >> 1;
    const int width = ad + 2 * p + 1;
    return (width + 62) / 32;                        // ceil((width + 31) / 32): the window covers the band at every c % 32
}

template <int W>
LP_FN void lp_init(Window<W>& L, LaneCtl& s, const int m, const int T, const int K, const int naInit, const Plane2* planes, const u32 planeOff, const u32 tgtOff)
{
    const int nplanes = (m + 31) >> 5;
    s.planeOff = planeOff; s.tgtOff = tgtOff;
    s.m = m; s.T = T; s.K = K; s.delta = T - m;
    const int ad = s.delta < 0 ? -s.delta : s.delta;
    const int p = (K - ad) >> 1;
    const int dmax = (s.delta > 0 ? s.delta : 0) + p;
    const int off = -dmax;
    s.top = off;
    s.sh = (u32)off & 31u;
    const int wi0 = off >> 5;                         // floor
    Plane2 prev = planes[planeOff + (u32)lp_clampi(wi0, 0, nplanes - 1)];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const Plane2 nx = planes[planeOff + (u32)lp_clampi(wi0 + j + 1, 0, nplanes - 1)];
        L.Q0[j] = lp_alignbit(nx.q0, prev.q0, s.sh);
        L.Q1[j] = lp_alignbit(nx.q1, prev.q1, s.sh);
        prev = nx;
        const int R = off + 32 * j;                  // first row of the word; rows < 0 are virtual (vertical delta -1)
        u32 mv;
        if (R >= 0) mv = 0u; else if (R <= -32) mv = ~0u; else mv = (1u << (u32)(-R)) - 1u;
        L.Mv[j] = mv; L.Pv[j] = ~mv;
        LP_SCHED_FENCE();                             // (word by word: 2 W loads in flight at once cost 2 W registers for nothing)
    }
    s.score = 0x3fffffff;
    s.wi = wi0 + naInit;
    s.lo = planes[planeOff + (u32)lp_clampi(s.wi, 0, nplanes - 1)];
    s.sb = off + 32 * naInit;                         // D[r][-1] = r + 1 at the bottom row r = off + 32 naInit - 1
}

// A LADDER, not a switch inside the block loop: the number of words only falls, so the scan is a descending chain of loops,
// one per window height.  (As `switch (na)` in one loop every word stayed live around it -- case 48 might run next -- and
// every case reloaded a few dozen registers from scratch at its entry; in the chain a word is dead once its stage is left.)
#define LP_STAGE(N)                                                                                         \
    if constexpr (N <= W && N >= 3) {                                                                       \
        if (na == N && b < nblkWave) {                                                                      \
            do {                                                                                            \
                u32 deny = 0;                                                                               \
                if (denySeed == 0xffffffffu) deny = 3u;                                                     \
                else if (denySeed) { denySeed = denySeed * 1664525u + 1013904223u; deny = (denySeed >> 28) & 3u; } \
                steps += N;                                                                                 \
                na = lp_uniform(lp_block<W, N>(L, s, tg, tgt, b, planes, deny));                                    \
                ++b;                                                                                        \
            } while (na == N && b < nblkWave);                                                              \
        }                                                                                                   \
    }

// The whole scan of one lane.  planes / tgt: the pools; planeOff / tgtOff: where the lane's query planes (ceil(m / 32) entries)
// and target planes (ceil(T / 32) entries) start in them.
// naWave / nblkWave: wave-uniform (the device passes the wave's maxima).  denySeed (tests): trims are refused in blocks
// where an LCG says so -- a lane must stay exact when the wave does not follow its vote; 0xffffffff = never trim.
template <int W>
LP_FN int lp_scan(const Plane2* planes, const u32 planeOff, const Tgt2* tgt, const u32 tgtOff, const int m, const int T, const int K,
                  const int naWave, const int nblkWave, u32 denySeed, int* wordSteps)
{
    Window<W> L;
    LaneCtl s;
    lp_init<W>(L, s, m, T, K, naWave, planes, planeOff, tgtOff);
    int na = naWave;
    long long steps = 0;
    Tgt2 tg = tgt[tgtOff];
    int b = 0;
    LP_STAGE(48) LP_STAGE(47) LP_STAGE(46) LP_STAGE(45) LP_STAGE(44) LP_STAGE(43) LP_STAGE(42) LP_STAGE(41) LP_STAGE(40) LP_STAGE(39)
    LP_STAGE(38) LP_STAGE(37) LP_STAGE(36) LP_STAGE(35) LP_STAGE(34) LP_STAGE(33) LP_STAGE(32) LP_STAGE(31) LP_STAGE(30) LP_STAGE(29)
    LP_STAGE(28) LP_STAGE(27) LP_STAGE(26) LP_STAGE(25) LP_STAGE(24) LP_STAGE(23) LP_STAGE(22) LP_STAGE(21) LP_STAGE(20) LP_STAGE(19)
    LP_STAGE(18) LP_STAGE(17) LP_STAGE(16) LP_STAGE(15) LP_STAGE(14) LP_STAGE(13) LP_STAGE(12) LP_STAGE(11) LP_STAGE(10) LP_STAGE(9)
    LP_STAGE(8) LP_STAGE(7) LP_STAGE(6) LP_STAGE(5) LP_STAGE(4) LP_STAGE(3)
    if (wordSteps) *wordSteps = (int)(steps > 0x7fffffff ? 0x7fffffff : steps);
    return s.score;
}

}  // namespace lanepair
}  // namespace edlib_amd
