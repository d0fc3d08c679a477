// one_pair.hip -- edlibAlign() on one small pair in ONE kernel launch.
//
// A call of edlibAlign() (edlib.cpp:146-301) is transform -> Peq -> scan -> (start locations) -> (path).  As a batch
// of one (engine.hip) that is several launches, an upload and host decisions between the phases: 90..900 us for pairs the
// reference's README quotes at 3.6 / 47 us (bindings/python/README-tmpl.rst:194-205).  For queries of up to 1024 rows
// against targets of up to 4096 columns (identity equality) the whole call runs here in one wave:
//
//   * the caller's bytes go into a pinned, device-visible mailbox (cached per host thread together with a stream); the
//     kernel reads them from there -- no upload command, no staging of tables;
//   * buildPeq (edlib.cpp:358-384): a 256-row table in LDS indexed by the raw target byte, filled with 64-bit LDS atomics
//     by the lanes walking the query (no alphabet transform at all: alphabetLength is counted on the host);
//   * the scan (myersCalcEditDistanceNW / SemiGlobal, :550-928) is the anti-diagonal schedule of kernel W on the first
//     ceil(m / 64) lanes, every row of every column (a pair this small needs no band; the user's k filters the answer),
//     target bytes and Peq words fetched from LDS one step ahead;
//   * HW start locations (:228-272): one reverse SHW scan per end location inside the same launch (the reversed query's
//     Peq is rebuilt in LDS);
//   * PATH (:276-289, 942-1141): the storing scan keeps the two planes of every block-step in LDS when the matrix of the
//     alignment window fits (ceil(m / 64) x T' x 16 bytes); the wave walks back 64 cells of the current diagonal per trip
//     (the reference's candidate order up > left > diagonal), the ops leave through the mailbox.
//
// What does not fit (more than 64 end locations, a store beyond the LDS budget, additional equalities, unknown modes)
// answers "not handled" and the call takes the general path; results are the same function of the DP matrix either way
// (tests/test_gpu_one_pair.py compares both paths with the reference).
#include "engine.hpp"
#include "block64.hpp"

#include <cstring>
#include <type_traits>

namespace edlib_amd {

namespace {

typedef unsigned long long u64;
typedef uint32_t u32;

const int kOneMaxQ = 1024, kOneMaxT = 4096, kOneMaxLoc = 64;
const int kOneLdsBudget = 150 * 1024;

struct OneHeader {            // mailbox, host -> device (followed by the query bytes, then the target bytes)
    int m, T, mode, task, k;
    int storeCap;             // entries the LDS column store may hold (0: PATH not requested)
    int pad[2];
};
struct OneResult {            // mailbox, device -> host (followed by kOneMaxLoc + 1 ends, as many starts, then the ops)
    int code;                 // 0 = done, 2 = not handled here
    int editDistance, numLocations, hasEnds, hasStarts, hasAlignment, alignmentLength, pad;
};

struct ScanOut { int best, cnt, last, score; };

// One scan of the query (Peq table in LDS) against tgt[toff + i * tstep], i < Tn: MODE 0 NW, 1 SHW, 2 HW.
// Lane l < nb owns block l; at step t it updates column t - l; {hout} moves one lane down per step by DPP.
template <bool STORE>
__device__ void scan_small(const u64* s_peq, const int nb, const uint8_t* s_t, const int toff, const int tstep, const int Tn,
                           const int m, const int mode, const int kthr, int* s_pos, u32* s_store, ScanOut& o)
{
    const int lane = threadIdx.x;
    const bool on = lane < nb;
    const u32 sh = (u32)(m - 1) & 63u;
    const bool tracker = lane == nb - 1;
    Block64 B{~0u, ~0u, 0u, 0u};                                     // column -1 (edlib.cpp:575-579)
    int sc = m, carry = 0;
    int best = kthr, cnt = 0, last = -1;
    const int top = mode == 2 ? 0 : 1;                               // row -1: HW 0, SHW / NW +1 (edlib.cpp:584, 779)
    const int nsteps = Tn + nb - 1;
    // The target byte of a column is fetched two steps ahead and its Peq word one step ahead: the two LDS reads a column needs
    // are then in different iterations, and neither sits on the step's dependent chain (a single wave has nobody to hide an
    // LDS round trip behind)
    auto byte_of = [&](int col) -> int { return (on && col >= 0 && col < Tn) ? (int)s_t[toff + col * tstep] : 0; };
    auto word_of = [&](int byte, int col) -> u64 { return (on && col >= 0 && col < Tn) ? s_peq[byte * nb + lane] : 0ull; };
    u64 eq = word_of(byte_of(-lane), -lane);
    int tbN = byte_of(1 - lane);
    for (int t = 0; t < nsteps; ++t) {
        const int col = t - lane;
        const u64 eqN = word_of(tbN, col + 1);
        const int tbNN = byte_of(col + 2);
        const int x = __builtin_amdgcn_update_dpp(top, carry, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
        if (on && col >= 0 && col < Tn) {
            u32 ph0, ph1, mh0, mh1, xh0, xh1;
            advance_block64(B, (u32)eq, (u32)(eq >> 32), (u32)x & 1u, ((u32)x >> 1) & 1u, ph0, ph1, mh0, mh1, xh0, xh1);
            const u32 hp = ph1 >> 31, hn = mh1 >> 31;
            carry = (int)(hp | (hn << 1));
            if (STORE) {                                             // the two planes of pair_kernels.hpp StoreEntry
                unsigned long long px, py;
                store_planes(B, ph0, ph1, xh0, xh1, px, py);
                *reinterpret_cast<uint4*>(s_store + (size_t)(col * nb + lane) * 4) = make_uint4((u32)px, (u32)(px >> 32), (u32)py, (u32)(py >> 32));
            }
            if (tracker) {
                const u64 ph = ((u64)ph1 << 32) | ph0, mh = ((u64)mh1 << 32) | mh0;
                sc += (int)((ph >> sh) & 1ull) - (int)((mh >> sh) & 1ull);
                if (mode != 0 && sc <= best) {                       // edlib.cpp:658-673
                    if (sc < best) { best = sc; cnt = 0; }
                    if (cnt < kOneMaxLoc) s_pos[cnt] = col;
                    ++cnt;
                    last = col;
                }
            }
        }
        eq = eqN; tbN = tbNN;
    }
    const int src = nb - 1;                                          // wave-uniform
    o.best = __builtin_amdgcn_readlane(cnt > 0 ? best : -1, src);
    o.cnt = __builtin_amdgcn_readlane(cnt, src);
    o.last = __builtin_amdgcn_readlane(last, src);
    o.score = __builtin_amdgcn_readlane(sc, src);
    __syncthreads();                                                 // positions / store visible to every lane
}

// reference buildPeq (edlib.cpp:358-384) keyed by the raw target byte: bit r of row[byte][block] = query[64 block + r] == byte
__device__ void build_peq_small(u64* s_peq, const int nb, const uint8_t* s_q, const int m, const bool reversed)
{
    for (int i = threadIdx.x; i < 256 * nb; i += 64) s_peq[i] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 64) {
        const int c = s_q[reversed ? m - 1 - i : i];
        atomicOr(&s_peq[c * nb + (i >> 6)], 1ull << (i & 63));
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64)
one_pair_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const OneHeader h = *reinterpret_cast<const OneHeader*>(in);
    const int m = h.m, T = h.T, nb = (m + 63) >> 6, lane = threadIdx.x;
    // ---- LDS: Peq table | query | target | end positions | ops | column store
    u64* s_peq = reinterpret_cast<u64*>(s_mem);
    uint8_t* s_q = s_mem + (size_t)256 * nb * 8;
    uint8_t* s_t = s_q + ((m + 15) & ~15);
    int* s_pos = reinterpret_cast<int*>(s_t + ((T + 15) & ~15));
    uint8_t* s_ops = reinterpret_cast<uint8_t*>(s_pos + kOneMaxLoc);
    u32* s_store = reinterpret_cast<u32*>(s_ops + ((m + T + 15) & ~15));
    {   // the caller's bytes: pinned host memory -> LDS, dwords (the mailbox pads to 16 bytes)
        const u32* qs = reinterpret_cast<const u32*>(in + sizeof(OneHeader));
        const u32* ts = reinterpret_cast<const u32*>(in + sizeof(OneHeader) + ((m + 15) & ~15));
        for (int i = lane; i < (m + 3) / 4; i += 64) reinterpret_cast<u32*>(s_q)[i] = qs[i];
        for (int i = lane; i < (T + 3) / 4; i += 64) reinterpret_cast<u32*>(s_t)[i] = ts[i];
    }
    __syncthreads();
    OneResult* res = reinterpret_cast<OneResult*>(out);
    int* ends = reinterpret_cast<int*>(out + sizeof(OneResult));
    int* starts = ends + kOneMaxLoc + 1;
    uint8_t* opsOut = reinterpret_cast<uint8_t*>(starts + kOneMaxLoc + 1);
    auto give_up = [&]() { if (lane == 0) { res->code = 2; __threadfence_system(); } };

    const int mode = h.mode, task = h.task, k = h.k;
    const int kthr = (k < 0 || k > m) ? m : k;                       // HW clamps k to m (edlib.cpp:566-568); SHW's best never exceeds m
    build_peq_small(s_peq, nb, s_q, m, false);
    // ---- phase 1: distance + end locations (NW with PATH: the storing scan right away)
    const bool nwStore = mode == 0 && task == 2 && nb * T <= h.storeCap;
    if (mode == 0 && task == 2 && !nwStore) { give_up(); return; }
    ScanOut o;
    if (nwStore) scan_small<true>(s_peq, nb, s_t, 0, 1, T, m, mode, kthr, s_pos, s_store, o);
    else scan_small<false>(s_peq, nb, s_t, 0, 1, T, m, mode, kthr, s_pos, s_store, o);
    int ed, nloc = 0; bool hasEnds = false;
    int first = -2;                                                   // (wave-uniform copy of the first end location)
    if (mode == 0) {
        const bool over = k >= 0 && o.score > k;                      // edlib.cpp:744-747, 917
        ed = over ? -1 : o.score;
        if (!over) { hasEnds = true; nloc = 1; first = T - 1; if (lane == 0) ends[0] = T - 1; }
    } else {
        // the empty prefix (position -1, score m) takes part exactly when the padded last block of the
        // reference would see it: W = 64 ceil(m / 64) - m > 0 (edlib.cpp:661, 670, 681-693)
        const int W = 64 * nb - m;
        const bool kAllowsM = k < 0 || k >= m;
        if (o.best < 0) {
            if (W > 0 && kAllowsM) { ed = m; hasEnds = true; nloc = 1; first = -1; if (lane == 0) ends[0] = -1; }
            else ed = -1;
        } else {
            const int lead = (W > 0 && o.best == m) ? 1 : 0;
            if (lead + o.cnt > kOneMaxLoc) { give_up(); return; }       // (lane i keeps location i for the reverse scans)
            ed = o.best; hasEnds = true;
            if (lead && lane == 0) ends[0] = -1;
            for (int i = lane; i < o.cnt; i += 64) ends[lead + i] = s_pos[i];
            nloc = lead + o.cnt;
            first = lead ? -1 : s_pos[0];
        }
    }
    bool hasStarts = false, hasAln = false; int alen = 0;
    int start0 = 0;
    if (ed >= 0 && task >= 1) {
        // ---- phase 2: start locations (edlib.cpp:228-272): 0 for NW / SHW; HW: reverse SHW per end location with k = ed
        hasStarts = true;
        if (mode != 2) { for (int i = lane; i < nloc; i += 64) starts[i] = 0; }
        else {
            __syncthreads();
            // the end locations are needed after s_pos is reused by the reverse scans: keep them in registers (lane i holds location i)
            int myEnd = -2;
            if (lane < nloc) myEnd = (lane == 0 && first == -1) ? -1 : s_pos[lane - ((first == -1) ? 1 : 0)];
            build_peq_small(s_peq, nb, s_q, m, true);
            for (int j = 0; j < nloc; ++j) {
                const int e = __builtin_amdgcn_readlane(myEnd, j);
                int st = 0;
                if (e != -1) {                                        // :237-249
                    // reverse query against the reversed prefix target[0..e], prefix mode, k = distance (:253-257); columns
                    // past m + distance cannot score <= distance, so the window stops there
                    const int win = (e + 1 < m + ed) ? e + 1 : m + ed;
                    ScanOut r;
                    scan_small<false>(s_peq, nb, s_t, e, -1, win, m, 1, ed, s_pos, s_store, r);
                    st = e - r.last;                                  // last reported position of the reverse scan (:260)
                }
                if (lane == 0) starts[j] = st;
                if (j == 0) start0 = st;
            }
            if (task == 2) build_peq_small(s_peq, nb, s_q, m, false);
        }
        // ---- phase 3: alignment path of the first location (edlib.cpp:276-289, 1161-1213)
        if (task == 2 && nloc > 0) {
            const int s0 = mode == 2 ? start0 : 0, e0 = first;
            const int len = e0 - s0 + 1;
            hasAln = true;
            if (len <= 0) {                                           // :1168-1175
                for (int i = lane; i < m; i += 64) s_ops[i] = 1;
                alen = m;
                __syncthreads();
                for (int i = lane; i < alen; i += 64) opsOut[i] = s_ops[i];
            } else {
                if (!nwStore) {
                    if (nb * len > h.storeCap) { give_up(); return; }
                    ScanOut r;
                    scan_small<true>(s_peq, nb, s_t, s0, 1, len, m, 0, m + len, s_pos, s_store, r);
                }
                // reference obtainAlignmentTraceback (edlib.cpp:942-1141) on the stored planes (up = x & ~y, left = x & y, diagonal = ~x
                // with MATCH iff y), the whole wave at once (round 5; lane 0 alone took 0.25 us per column: half of a 1 k x 1 k
                // call): lane i looks at cell (r - i, c - i) of the current diagonal; the run of diagonal moves ends at the first
                // cell whose x bit says an indel move is possible (or outside the matrix) -- one ballot --, the lanes before it
                // write their MATCH / MISMATCH ops side by side, the cell that stopped the run takes up-moves while the "up"
                // plane says so (one count-leading-ones inside its block), else one left move (the reference's preference up >
                // left > diagonal, :1020, 1054, 1085); row -1 / column -1 leave the tail (:1040-1046, 1070-1078).  Ops are written
                // back to front.  (ring32_kernels.hip: traceback32_kernel is the same walk over HBM, 32 lanes per unit.)
                int wpos = m + len;
                {
                    int r = m - 1, c = len - 1;
                    bool done = false;
                    while (!done) {                                   // (r, c, wpos, done are wave-uniform)
                        const int ri = r - lane, ci = c - lane;
                        const bool valid = ri >= 0 && ci >= 0;
                        u64 x = 0, y = 0;
                        if (valid) {
                            const uint4 e = *reinterpret_cast<const uint4*>(s_store + (size_t)(ci * nb + (ri >> 6)) * 4);
                            x = ((u64)e.y << 32) | e.x; y = ((u64)e.w << 32) | e.z;
                        }
                        const u32 bit = (u32)ri & 63u;
                        const bool xb = (x >> bit) & 1ull, yb = (y >> bit) & 1ull;
                        const u64 stops = __builtin_amdgcn_ballot_w64(!valid || xb);
                        const int j = stops ? __builtin_ctzll(stops) : 64;        // diagonal moves before the first stop
                        if (lane < j) s_ops[wpos - 1 - lane] = yb ? (uint8_t)0 : (uint8_t)3;
                        const int src = j < 64 ? j : 0;
                        const u32 X0 = (u32)__shfl((int)(u32)x, src, 64), X1 = (u32)__shfl((int)(u32)(x >> 32), src, 64);
                        const u32 Y0 = (u32)__shfl((int)(u32)y, src, 64), Y1 = (u32)__shfl((int)(u32)(y >> 32), src, 64);
                        r -= j; c -= j; wpos -= j;
                        if (j < 64) {
                            if (r >= 0 && c >= 0) {                   // the cell that stopped the run: x is set there
                                const u64 X = ((u64)X1 << 32) | X0, Y = ((u64)Y1 << 32) | Y0;
                                const u32 bb = (u32)r & 63u;
                                const u64 nx = ~((X & ~Y) << (63u - bb));        // row r at bit 63; the zeros shifted in end the run
                                const int ups = nx ? __builtin_clzll(nx) : 64;   // leading ones: <= bb + 1
                                if (ups > 0) {                        // INSERTs
                                    if (lane < ups) s_ops[wpos - 1 - lane] = 1;
                                    r -= ups; wpos -= ups;
                                } else {                              // DELETE (x set, not up: Ph)
                                    if (lane == 0) s_ops[wpos - 1] = 2;
                                    c -= 1; wpos -= 1;
                                }
                            }
                            if (r < 0 || c < 0) {                     // the matrix boundary: the rest is one run
                                const int cnt = (r < 0 && c < 0) ? 0 : (c < 0 ? r + 1 : c + 1);
                                const uint8_t op = c < 0 ? (uint8_t)1 : (uint8_t)2;      // INSERT / DELETE
                                for (int i = lane; i < cnt; i += 64) s_ops[wpos - 1 - i] = op;
                                wpos -= cnt;
                                done = true;
                            }
                        }
                    }
                }
                wpos = __builtin_amdgcn_readfirstlane(wpos);
                alen = m + len - wpos;
                __syncthreads();
                for (int i = lane; i < alen; i += 64) opsOut[i] = s_ops[wpos + i];
            }
        }
    }
    __syncthreads();
    if (lane == 0) {
        res->editDistance = ed; res->numLocations = nloc; res->hasEnds = hasEnds ? 1 : 0; res->hasStarts = hasStarts ? 1 : 0;
        res->hasAlignment = hasAln ? 1 : 0; res->alignmentLength = alen;
        res->code = 0;
        __threadfence_system();
    }
}

// ---------------------------------------------------------------- NW distance of one small pair, two waves
//
// A distance call is a chain of dependent steps on a wave that has its SIMD to itself (every issued
// instruction costs 2.2-3.5 ns there), and the two halves of the target are independent of each other.  So for NW
// without a path the call runs as TWO half scans on the two waves of one workgroup -- wave 0: query against the left
// half, wave 1: reversed query against the reversed right half (what the first Hirschberg level does,
// edlib.cpp:1246-1260) -- and D[m][T] = min over i of L[i] + R[i+1].  The scans are the wide kernel's step on 32-row
// words (wide_kernels.hip): lane = word, 18 issued instructions per step, sixteen steps per straight-line block; the
// feeds of lane 0 (row -1 is +1 per column, the next target byte as the LDS offset of its Peq row) are laid out for the
// whole half before the scan.  Nothing is accumulated per step: every value of the last column is the top boundary plus
// the vertical deltas above it, i.e. prefix sums of popcounts over the final Pv / Mv words.
#define OP3_OR_NOR(a, b, c)  ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xf1))   /* a | ~(b | c)  */
#define OP3_XOR_OR(a, b, c)  ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xde))   /* (a ^ c) | b   */
#define OP3_BFI(m, a, b)     ((u32)__builtin_amdgcn_bitop3_b32((m), (a), (b), 0xca))   /* m ? a : b     */

__global__ void __launch_bounds__(128)
one_pair_nw_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const OneHeader h = *reinterpret_cast<const OneHeader*>(in);
    const int m = h.m, T = h.T, k = h.k;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nw = (m + 31) >> 5;                                    // 32-row words: at most 32 lanes of a wave
    int shiftS = 2; while ((1 << shiftS) < 4 * nw) ++shiftS;         // Peq row stride in bytes: 4 nw rounded up to a power of two
    const u32 tableBytes = 256u << shiftS;
    const int lw = T / 2, rw = T - lw;                               // edlib.cpp:1247-1248
    const int myT = wv == 0 ? lw : rw;
    // ---- LDS: [Peq forward | Peq reverse] at address 0 | query | target | feeds of the two halves | final columns | minima
    u32* s_peq = reinterpret_cast<u32*>(s_mem);
    uint8_t* s_q = s_mem + 2 * tableBytes;
    uint8_t* s_t = s_q + ((m + 15) & ~15);
    u64* s_feed = reinterpret_cast<u64*>(s_t + ((T + 15) & ~15));    // [2][rw + 2]
    u32* s_col = reinterpret_cast<u32*>(s_feed + 2 * (rw + 2));      // [2][3][32]: Pv, Mv, score above the word
    int* s_min = reinterpret_cast<int*>(s_col + 2 * 3 * 32);
    {
        const u32* qs = reinterpret_cast<const u32*>(in + sizeof(OneHeader));
        const u32* ts = reinterpret_cast<const u32*>(in + sizeof(OneHeader) + ((m + 15) & ~15));
        for (int i = tid; i < (m + 3) / 4; i += 128) reinterpret_cast<u32*>(s_q)[i] = qs[i];
        for (int i = tid; i < (T + 3) / 4; i += 128) reinterpret_cast<u32*>(s_t)[i] = ts[i];
        for (u32 i = tid; i < 2 * tableBytes / 16; i += 128) reinterpret_cast<uint4*>(s_peq)[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // buildPeq (edlib.cpp:358-384) keyed by the raw byte, forward and reversed: bit r of row[byte][word] = query[32 word + r] == byte
    for (int i = tid; i < m; i += 128) {
        const u32 c = s_q[i];
        atomicOr(&s_peq[((c << shiftS) >> 2) + (i >> 5)], 1u << (i & 31));
        const int ir = m - 1 - i;
        atomicOr(&s_peq[((tableBytes + (c << shiftS)) >> 2) + (ir >> 5)], 1u << (ir & 31));
    }
    // feeds of lane 0: slot j + 1 = column j of the half: {A: row -1 delivers +1 (edlib.cpp:779), B: Peq row offset of column j + 1}
    {
        u64* f = s_feed + wv * (rw + 2);
        const u32 base = wv == 0 ? 0u : tableBytes;
        for (int j = lane - 1; j < myT; j += 64) {
            const int c = j + 1;                                      // column of the half whose byte this feed carries
            u32 off = base;
            if (c < myT) off += (u32)s_t[wv == 0 ? c : T - 1 - c] << shiftS;
            f[j + 1] = ((u64)off << 32) | 0x80000000u;
        }
    }
    __syncthreads();
    // ---- the half scan of this wave: lane l < nw owns word l, at step t it updates column t - l
    u32 Pv = ~0u, Mv = 0u;                                           // column -1 (edlib.cpp:575-579)
    if (myT > 0) {
        const u32 laneOff = 4u * (u32)lane;
        const u32 feedAddr = (u32)(size_t)(__attribute__((address_space(3))) u64*)(s_feed + wv * (rw + 2));
        const u32 rowMask = ~((1u << shiftS) - 1u) & 0x7fffffffu;
        auto feed_at = [&](const int slot) -> u64 { return *(const __attribute__((address_space(3))) u64*)(size_t)(feedAddr + 8u * (u32)slot); };
        u32 PhOut = 0, Bout = 0, eq = 0;
        auto step = [&](auto genericTag, const int t, const u64 feed) {
            constexpr bool GENERIC = decltype(genericTag)::value;
            const u32 A = (u32)__builtin_amdgcn_update_dpp((int)(u32)feed, (int)PhOut, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
            const u32 Bv = (u32)__builtin_amdgcn_update_dpp((int)(u32)(feed >> 32), (int)Bout, 0x138, 0xf, 0xf, false);
            const u32 eqNxt = *(const __attribute__((address_space(3))) u32*)(size_t)((Bv & rowMask) | laneOff);
            if (!GENERIC || (lane < nw && (unsigned)(t - lane) < (unsigned)myT)) {
                const u32 hneg = Bv >> 31;                             // reference calculateBlock (edlib.cpp:412-447) on a 32-row word
                const u32 eqn = eq | hneg;
                const u32 xv = eq | Mv;
                const u32 sum = (eqn & Pv) + Pv;
                const u32 xh = OP3_XOR_OR(sum, eqn, Pv);
                const u32 ph = OP3_OR_NOR(Mv, xh, Pv);
                const u32 mh = Pv & xh;
                const u32 phs = __builtin_amdgcn_alignbit(ph, A, 31);
                const u32 mhs = __builtin_amdgcn_alignbit(mh, Bv, 31);
                Pv = OP3_OR_NOR(mhs, xv, phs);
                Mv = phs & xv;
                PhOut = ph;
                Bout = OP3_BFI(0x80000000u, mh, Bv);
            } else {
                Bout = Bv;
            }
            eq = eqNxt;
        };
        const int nsteps = myT + nw - 1;
        // (the feed of step t is slot t + 1: column t of lane 0; step -1 only hands the first symbol down)
        int t = -1;
        while (t < nsteps) {
            if (t >= nw - 1 && t + 15 <= myT - 1) {                    // every lane < nw inside its columns: sixteen straight-line steps
                const u64 f0 = feed_at(t + 1), f1 = feed_at(t + 2), f2 = feed_at(t + 3), f3 = feed_at(t + 4);
                step(std::false_type{}, t, f0);
                const u64 f4 = feed_at(t + 5);
                step(std::false_type{}, t + 1, f1);
                const u64 f5 = feed_at(t + 6);
                step(std::false_type{}, t + 2, f2);
                const u64 f6 = feed_at(t + 7);
                step(std::false_type{}, t + 3, f3);
                const u64 f7 = feed_at(t + 8);
                step(std::false_type{}, t + 4, f4);
                const u64 f8 = feed_at(t + 9);
                step(std::false_type{}, t + 5, f5);
                const u64 f9 = feed_at(t + 10);
                step(std::false_type{}, t + 6, f6);
                const u64 f10 = feed_at(t + 11);
                step(std::false_type{}, t + 7, f7);
                const u64 f11 = feed_at(t + 12);
                step(std::false_type{}, t + 8, f8);
                const u64 f12 = feed_at(t + 13);
                step(std::false_type{}, t + 9, f9);
                const u64 f13 = feed_at(t + 14);
                step(std::false_type{}, t + 10, f10);
                const u64 f14 = feed_at(t + 15);
                step(std::false_type{}, t + 11, f11);
                const u64 f15 = feed_at(t + 16);
                step(std::false_type{}, t + 12, f12);
                step(std::false_type{}, t + 13, f13);
                step(std::false_type{}, t + 14, f14);
                step(std::false_type{}, t + 15, f15);
                t += 16;
            } else {
                step(std::true_type{}, t, (t + 1 <= myT) ? feed_at(t + 1) : 0ull);
                t += 1;
            }
        }
    }
    // ---- the last column of the half: score above each word = (columns of the half) + vertical deltas of the words above
    {
        const int d = (lane < nw) ? __popc(Pv) - __popc(Mv) : 0;       // (an empty half: Pv = ~0: +1 per row, D[i][-1] = i + 1)
        int incl = d;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (lane < 32) {
            u32* c = s_col + wv * 96;
            c[lane] = Pv; c[32 + lane] = Mv; c[64 + lane] = (u32)(myT + incl - d);
        }
    }
    __syncthreads();
    // ---- D[m][T] = min over i in [-1, m-1] of L[i] + R[i+1]   (L[-1] = lw, R[m] = rw; edlib.cpp:1314-1353)
    auto cell = [&](const int half, const int r) -> int {             // value of row r in the half's last column
        const u32* c = s_col + half * 96;
        const int wd = r >> 5, bit = r & 31;
        const u32 upto = (bit == 31) ? ~0u : ((2u << bit) - 1u);
        return (int)c[64 + wd] + __popc(c[wd] & upto) - __popc(c[32 + wd] & upto);
    };
    int best = 0x3fffffff;
    for (int i = tid - 1; i <= m - 1; i += 128) {
        const int L = (i < 0) ? lw : cell(0, i);
        const int R = (i == m - 1) ? rw : cell(1, m - 2 - i);
        best = (L + R < best) ? L + R : best;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(best, off, 64); best = o < best ? o : best; }
    if (lane == 0) s_min[wv] = best;
    __syncthreads();
    if (tid == 0) {
        OneResult* res = reinterpret_cast<OneResult*>(out);
        int* ends = reinterpret_cast<int*>(out + sizeof(OneResult));
        int* starts = ends + kOneMaxLoc + 1;
        const int ed0 = s_min[0] < s_min[1] ? s_min[0] : s_min[1];
        const bool over = k >= 0 && ed0 > k;                          // edlib.cpp:744-747, 917
        res->editDistance = over ? -1 : ed0;
        res->numLocations = over ? 0 : 1; res->hasEnds = over ? 0 : 1;
        res->hasStarts = (!over && h.task >= 1) ? 1 : 0;              // NW: start 0 (:267-271)
        res->hasAlignment = 0; res->alignmentLength = 0;
        if (!over) { ends[0] = T - 1; starts[0] = 0; }
        res->code = 0;
        __threadfence_system();
    }
}

// per-thread context: stream + mailboxes (cached for the life of the thread)
struct OneCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    PinBuf in, out;
    int attrDevice = -1;                    // device the kernel's dynamic-LDS limit was raised on
    bool nwKernelOk = false;                // one_pair_nw_kernel has no static LDS in front of its tables
    // (runs at thread exit, for the main thread at process exit: no HIP call here; every call ended with a stream synchronisation)
    ~OneCtx() { if (stream) pool_stream_put(device, stream); }
};

}  // namespace

// 0 = answered in *out, 1 = error (last_error set), 2 = not handled here (the caller takes the general path)
int align_one_fused(const char* q, int m, const char* t, int T, EdlibAlignConfig cfg, EdlibAlignResult* out)
{
    static const bool enabled = !(getenv("EDLIB_AMD_ONEPAIR") && getenv("EDLIB_AMD_ONEPAIR")[0] == '0');
    if (!enabled || m < 1 || T < 1 || m > kOneMaxQ || T > kOneMaxT) return 2;
    if (cfg.additionalEqualities && cfg.additionalEqualitiesLength > 0) return 2;
    const int mode = (int)cfg.mode, task = (int)cfg.task;
    if (mode < 0 || mode > 2 || task < 0 || task > 2) return 2;
    const int nb = (m + 63) / 64;
    // NW without a path: two half scans on two waves (one_pair_nw_kernel), any size this file takes
    const bool nwKernel = mode == 0 && task != 2 && T >= 2;
    // One wave walks T + nb dependent steps of ~0.2 us here; beyond ~400 steps the batch-of-one path (ring kernel: ~0.12 us
    // per step behind ~45 us of launches and copies) is the faster one for distances.  PATH is faster here whenever its
    // store fits (1 k x 1 k: the general path takes 900 us).
    if (!nwKernel && task != 2 && T + nb > 400) return 2;
    // LDS: Peq table + sequences + positions + ops, the rest is the column store of a PATH call
    const size_t fixed = (size_t)256 * nb * 8 + ((m + 15) & ~15) + ((T + 15) & ~15) + kOneMaxLoc * sizeof(int) + ((m + T + 15) & ~15);
    if (fixed > (size_t)kOneLdsBudget) return 2;
    size_t storeCap = 0;
    if (task == 2) {
        storeCap = ((size_t)kOneLdsBudget - fixed) / 16;
        // the window of the alignment: the whole target (NW), at most m + distance <= 2m columns (SHW / HW)
        const long long cols = mode == 0 ? T : std::min<long long>(T, 2LL * m);
        if ((long long)nb * cols > (long long)storeCap) return 2;
    }
    const int dev = default_device();
    if (device_count() == 0) { set_error("no usable HIP device (this library has no CPU fallback)"); return 1; }
    static thread_local OneCtx ctx;
    pool_quarantine(false);
    DeviceGuard guard(dev);
    EDLIB_AMD_HIP(guard.status);
    if (ctx.device != dev || !ctx.stream || !ctx.in.p || !ctx.out.p) {
        // (Re)initialisation is transactional: the context names a device only while its stream and both mailboxes exist,
        // so a failure part-way leaves "no context" and the next call starts over (never a null mailbox behind a valid
        // device).  The old device's stream goes back under THAT device (pool_stream_release files under the current one).
        if (ctx.stream) { (void)hipStreamSynchronize(ctx.stream); pool_stream_put(ctx.device, ctx.stream); ctx.stream = nullptr; }
        ctx.device = -1; ctx.attrDevice = -1;
        ctx.in.release(); ctx.out.release();
        hipStream_t st = nullptr;
        EDLIB_AMD_HIP(pool_stream(&st));
        if (ctx.in.alloc(sizeof(OneHeader) + kOneMaxQ + kOneMaxT + 64) != hipSuccess ||
            ctx.out.alloc(sizeof(OneResult) + 2 * (kOneMaxLoc + 1) * sizeof(int) + kOneMaxQ + kOneMaxT + 64) != hipSuccess) {
            ctx.in.release(); ctx.out.release();
            pool_stream_put(dev, st);
            set_error("single-pair context: pinned mailbox allocation failed");
            pool_quarantine(true);
            return 1;
        }
        ctx.stream = st;
        ctx.device = dev;
    }
    if (ctx.attrDevice != dev) {            // (a function attribute is per device)
        EDLIB_AMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(one_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kOneLdsBudget + 1024));
        EDLIB_AMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(one_pair_nw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kOneLdsBudget + 1024));
        hipFuncAttributes fa;               // the Peq tables of the two-wave kernel are addressed from LDS address 0
        EDLIB_AMD_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(one_pair_nw_kernel)));
        ctx.nwKernelOk = fa.sharedSizeBytes == 0;
        ctx.attrDevice = dev;
    }
    OneHeader* h = reinterpret_cast<OneHeader*>(ctx.in.p);
    h->m = m; h->T = T; h->mode = mode; h->task = task; h->k = cfg.k; h->storeCap = (int)std::min<size_t>(storeCap, 0x7fffffff);
    memcpy(ctx.in.p + sizeof(OneHeader), q, (size_t)m);
    memcpy(ctx.in.p + sizeof(OneHeader) + ((m + 15) & ~15), t, (size_t)T);
    OneResult* r = reinterpret_cast<OneResult*>(ctx.out.p);
    r->code = -1;
    const size_t lds = fixed + (task == 2 ? storeCap * 16 : 0) + 64;
    if (nwKernel && ctx.nwKernelOk) {
        const int nw = (m + 31) / 32;
        int shiftS = 2; while ((1 << shiftS) < 4 * nw) ++shiftS;
        const size_t ldsNw = 2 * ((size_t)256 << shiftS) + ((m + 15) & ~15) + ((T + 15) & ~15) + 2 * (size_t)(T - T / 2 + 2) * 8 + 2 * 3 * 32 * 4 + 64;
        hipLaunchKernelGGL(one_pair_nw_kernel, dim3(1), dim3(128), ldsNw, ctx.stream, ctx.in.p, ctx.out.p);
    } else if (nwKernel && T + nb > 400) return 2;
    else
    hipLaunchKernelGGL(one_pair_kernel, dim3(1), dim3(64), lds, ctx.stream, ctx.in.p, ctx.out.p);
    EDLIB_AMD_HIP(hipGetLastError());
    // alphabetLength (edlib.cpp:162, transformSequences :1417-1462) while the kernel runs: distinct bytes of query and target
    int alpha = 0;
    {
        bool seen[256] = {false};
        for (int i = 0; i < m; ++i) if (!seen[(uint8_t)q[i]]) { seen[(uint8_t)q[i]] = true; ++alpha; }
        for (int i = 0; i < T; ++i) if (!seen[(uint8_t)t[i]]) { seen[(uint8_t)t[i]] = true; ++alpha; }
    }
    EDLIB_AMD_HIP(hipStreamSynchronize(ctx.stream));
    if (r->code == 2) return 2;
    if (r->code != 0) { set_error("the single-pair kernel did not report (code %d)", r->code); return 1; }
    const int* ends = reinterpret_cast<const int*>(ctx.out.p + sizeof(OneResult));
    const int* starts = ends + kOneMaxLoc + 1;
    const uint8_t* ops = reinterpret_cast<const uint8_t*>(starts + kOneMaxLoc + 1);
    out->status = EDLIB_STATUS_OK; out->editDistance = r->editDistance; out->alphabetLength = alpha;
    out->endLocations = nullptr; out->startLocations = nullptr; out->numLocations = 0; out->alignment = nullptr; out->alignmentLength = 0;
    const size_t n = (size_t)r->numLocations;
    if (r->hasEnds) {
        out->endLocations = static_cast<int*>(malloc(sizeof(int) * std::max<size_t>(n, 1)));
        if (out->endLocations) memcpy(out->endLocations, ends, n * sizeof(int));
        out->numLocations = (int)n;
    }
    if (r->hasStarts) {
        out->startLocations = static_cast<int*>(malloc(sizeof(int) * std::max<size_t>(n, 1)));
        if (out->startLocations) memcpy(out->startLocations, starts, n * sizeof(int));
    }
    if (r->hasAlignment) {
        const size_t len = (size_t)r->alignmentLength;
        out->alignment = static_cast<unsigned char*>(malloc(std::max<size_t>(len, 1)));
        if (out->alignment) memcpy(out->alignment, ops, len);
        out->alignmentLength = (int)len;
    }
    if ((r->hasEnds && !out->endLocations) || (r->hasStarts && !out->startLocations) || (r->hasAlignment && !out->alignment)) {
        free(out->endLocations); free(out->startLocations); free(out->alignment);
        set_error("out of host memory");
        return 1;
    }
    return 0;
}

}  // namespace edlib_amd
