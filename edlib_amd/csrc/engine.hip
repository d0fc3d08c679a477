// engine.hip -- batch orchestration (host) for the MI355X edit-distance engine.
//
// The reference does, per call (edlib.cpp:146-301): transform -> Peq -> k-doubling
// scan -> (start locations) -> (path).  Here the same phases run once per BATCH:
//   phase 1  distance + end locations   reads-per-lane kernel (shared target, <= 4
//                                       target symbols, query <= 256) or
//                                       block-per-lane kernel (everything else)
//   phase 2  HW start locations         reverse SHW units on the block-per-lane kernel
//                                       (edlib.cpp:230-266)
//   phase 3  alignment path             NW units with column store + traceback kernel
//                                       (edlib.cpp:276-289, 1161-1213)
// Work is narrowed like in the reference (Ukkonen band, thresholds that grow until they hold the
// distance), but per batch and with thresholds chosen for the hardware: reads run a banded first pass
// at a small k and only the leftovers a full pass (runReads), NW pairs climb lane-ring sizes
// (solveGlobalDistances).  Thresholds only steer work: every result is a function of the full DP
// matrix, so the user's k merely filters it (SURVEY.md §7 "results are band-independent").
#include "engine.hpp"
#include "flat_results.hpp"
#include <sched.h>

#if defined(__x86_64__)
#include <immintrin.h>                 // build_tables: 16 target bytes per step through the alphabet scan (host)
#endif
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <system_error>
#include <thread>

namespace edlib_amd {

// ------------------------------------------------------------------- tables

// byte x byte equality matrix of EqualityDefinition (edlib.cpp:63-94); only built when there are additional
// equalities (the kernels take "no matrix" as the identity)
static void build_eq8(std::vector<uint8_t>& eq8, const EdlibEqualityPair* eqs, int neq) {
    eq8.assign(256 * 256, 0);
    for (int a = 0; a < 256; ++a) eq8[a * 256 + a] = 1;
    for (int i = 0; i < neq; ++i) {
        const int a = (uint8_t)eqs[i].first, b = (uint8_t)eqs[i].second;
        eq8[a * 256 + b] = eq8[b * 256 + a] = 1;
    }
}

// Position of the first byte at or after `from` that is not in the set described by the two nibble tables (or `n`):
// byte b is in the set iff lo[b & 15] & hi[b >> 4] != 0 (every distinct high nibble of the set owns one bit: exact for
// sets with at most 8 distinct high nibbles -- DNA, protein, text).  16 bytes per step with pshufb where the CPU has it:
// the alphabet scan of a 5 Mb target was 1-2 ms of every single edlibAlign() call against it.
#if defined(__x86_64__)
__attribute__((target("ssse3")))
static long long first_outside_ssse3(const uint8_t* p, long long from, long long n, const uint8_t* lo, const uint8_t* hi)
{
    const __m128i L = _mm_loadu_si128(reinterpret_cast<const __m128i*>(lo)), H = _mm_loadu_si128(reinterpret_cast<const __m128i*>(hi));
    const __m128i nib = _mm_set1_epi8(0x0f), zero = _mm_setzero_si128();
    long long i = from;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i));
        const __m128i a = _mm_shuffle_epi8(L, _mm_and_si128(v, nib));
        const __m128i b = _mm_shuffle_epi8(H, _mm_and_si128(_mm_srli_epi16(v, 4), nib));
        const int miss = _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(a, b), zero));
        if (miss) return i + __builtin_ctz((unsigned)miss);
    }
    return i;                                                   // the tail (< 16 bytes) is the caller's
}
#endif

static void build_tables(Tables& tab, const uint8_t* targets, long long totalTargetBytes,
                         const EdlibEqualityPair* eqs, int neq) {
    memset(tab.presence, 0, sizeof tab.presence);
    memset(tab.tlut, 0, sizeof tab.tlut);
    memset(tab.idToByte, 0, sizeof tab.idToByte);
    bool seen[256] = {false};
    tab.sigmaT = 0;
    uint8_t lo[16] = {0}, hi[16] = {0}; int hiBit[16]; int hiBits = 0; bool nibbleOk = true;
    for (int h = 0; h < 16; ++h) hiBit[h] = -1;
#if defined(__x86_64__)
    static const bool haveSsse3 = __builtin_cpu_supports("ssse3");
#else
    static const bool haveSsse3 = false;
#endif
    long long i = 0;
    while (i < totalTargetBytes) {
#if defined(__x86_64__)
        if (haveSsse3 && nibbleOk && tab.sigmaT > 0) {          // skip what is already known, 16 bytes at a time
            i = first_outside_ssse3(targets, i, totalTargetBytes, lo, hi);
            if (i >= totalTargetBytes) break;
        }
#endif
        const uint8_t b = targets[i++];
        if (!seen[b]) {                                          // first appearance order (transformSequences, edlib.cpp:1417-1462)
            seen[b] = true;
            tab.tlut[b] = (uint8_t)tab.sigmaT;
            tab.idToByte[tab.sigmaT] = b;
            tab.presence[b >> 5] |= 1u << (b & 31);
            ++tab.sigmaT;
            if (hiBit[b >> 4] < 0) { if (hiBits < 8) hiBit[b >> 4] = hiBits++; else nibbleOk = false; }
            if (nibbleOk) { hi[b >> 4] = (uint8_t)(1u << hiBit[b >> 4]); lo[b & 15] |= (uint8_t)(1u << hiBit[b >> 4]); }
        }
    }
    // EqualityDefinition (edlib.cpp:63-94) on raw bytes.  The reference keeps a pair only
    // when both characters occur in that call's alphabet; a pair whose characters do not
    // occur can never be consulted, so the byte-level relation gives identical DP matrices.
    tab.eq8.clear();
    if (neq > 0) build_eq8(tab.eq8, eqs, neq);
    for (int q = 0; q < 256; ++q) {
        uint16_t mask = 0;
        for (int s = 0; s < tab.sigmaT && s < 16; ++s)
            if (tab.eq8.empty() ? (q == tab.idToByte[s]) : (tab.eq8[q * 256 + tab.idToByte[s]] != 0)) mask |= (uint16_t)(1u << s);
        tab.eqtbl[q] = mask;
    }
}

// --------------------------------------------------------- alphabetLength

// Number of distinct byte values in query (and, for non-shared batches, target):
// the alphabetLength field (edlib.cpp:162, transformSequences :1417-1462).
// One workgroup per unit: aligned 16-byte loads (the pools are padded by 16 bytes and 16-byte aligned), every byte
// marks its entry of a 256-entry table in LDS (lanes that write the same entry write the same value), and the count
// of marked entries is the answer.  (Round 1 / 2 kept a 256-bit set per thread in registers: ~25 VALU ops per byte
// behind dword loads, 2.0 ms for 100,000 pairs of 10 kb -- 1 TB/s; this form is bound by the loads.)
__global__ void __launch_bounds__(256)
alphabet_count_kernel(const uint8_t* __restrict__ qpool, const long long* __restrict__ qoff,
                      const uint8_t* __restrict__ tpool, const long long* __restrict__ toff,
                      int shared, const uint32_t* __restrict__ basePresence,
                      const int* __restrict__ unitIdx, int* __restrict__ out)
{
    __shared__ uint32_t s_seen[256];
    s_seen[threadIdx.x] = 0u;
    __syncthreads();
    const int u = unitIdx[blockIdx.x];
    auto scan = [&](const uint8_t* pool, long long lo, long long hi) {
        for (long long c = (lo >> 4) + threadIdx.x; (c << 4) < hi; c += 256) {
            const uint4 v = *reinterpret_cast<const uint4*>(pool + (c << 4));
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            const long long at = c << 4;
            if (at >= lo && at + 16 <= hi) {
#pragma unroll
                for (int k = 0; k < 16; ++k) s_seen[(d[k >> 2] >> (8 * (k & 3))) & 0xffu] = 1u;
            } else {                                                     // first / last chunk of the sequence
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (at + k >= lo && at + k < hi) s_seen[(d[k >> 2] >> (8 * (k & 3))) & 0xffu] = 1u;
            }
        }
    };
    scan(qpool, qoff[u], qoff[u + 1]);
    if (!shared) scan(tpool, toff[u], toff[u + 1]);
    __syncthreads();
    bool present = s_seen[threadIdx.x] != 0u;
    if (shared) present = present || ((basePresence[threadIdx.x >> 5] >> (threadIdx.x & 31)) & 1u);
    const int n = __syncthreads_count(present ? 1 : 0);
    if (threadIdx.x == 0) out[blockIdx.x] = n;
}

// The same count for batches of SHORT sequences: a wave per unit, four units per workgroup (a workgroup per unit has
// three idle waves and a launch of 262,144 workgroups for as many 150-base pairs: ~1 ms on the side stream, which the
// collection of a flat batch then waited for).  Each wave owns 256 bytes of the table.
__global__ void __launch_bounds__(256)
alphabet_count_short_kernel(const uint8_t* __restrict__ qpool, const long long* __restrict__ qoff,
                            const uint8_t* __restrict__ tpool, const long long* __restrict__ toff,
                            int shared, const uint32_t* __restrict__ basePresence,
                            const int* __restrict__ unitIdx, int n, int* __restrict__ out)
{
    __shared__ uint32_t s_seen[4][64];                               // per wave: 256 one-byte marks
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wv;
    if (slot >= n) return;
    uint8_t* const seen = reinterpret_cast<uint8_t*>(s_seen[wv]);
    s_seen[wv][lane] = 0u;
    const int u = unitIdx[slot];
    auto scan = [&](const uint8_t* pool, long long lo, long long hi) {
        for (long long c = (lo >> 4) + lane; (c << 4) < hi; c += 64) {
            const uint4 v = *reinterpret_cast<const uint4*>(pool + (c << 4));
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            const long long at = c << 4;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (at + k >= lo && at + k < hi) seen[(d[k >> 2] >> (8 * (k & 3))) & 0xffu] = 1;
        }
    };
    scan(qpool, qoff[u], qoff[u + 1]);
    if (!shared) scan(tpool, toff[u], toff[u + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t w = s_seen[wv][lane];                                   // four marks
    if (shared) {
        const uint32_t bits = (basePresence[lane >> 3] >> (4 * (lane & 7))) & 0xfu;
        w |= (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
    }
    int cnt = ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) out[slot] = cnt;
}

// --------------------------------------------------------------- Batch: init

Batch::~Batch() {
    DeviceGuard guard(device_);
    if (side_) { (void)hipStreamSynchronize(side_); pool_stream_release(side_); }
    if (aux_) { (void)hipStreamSynchronize(aux_); (void)hipStreamDestroy(aux_); }
    if (stream_) { (void)hipStreamSynchronize(stream_); pool_stream_release(stream_); }
    wideGateRelease();           // (behind the synchronisation: a failed run may still have had a wide launch in flight)
    for (auto& p : scanEvents_) { pool_event_release(p.first, device_); pool_event_release(p.second, device_); }
}

int roundup(int x, int q) { return (x + q - 1) / q * q; }

int Batch::init(const char* queries, const long long* qoff, int n, const char* targets,
                const long long* toff, int numTargets, EdlibAlignConfig cfg, int device)
{
    if (n < 0 || (numTargets != 1 && numTargets != n)) { set_error("bad batch shape"); return 1; }
    const int ndev = device_count();
    if (ndev == 0) { set_error("no usable HIP device (this library has no CPU fallback)"); return 1; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return 1; }
    cfg_ = cfg;
    if (cfg.additionalEqualities && cfg.additionalEqualitiesLength > 0)
        eqs_.assign(cfg.additionalEqualities, cfg.additionalEqualities + cfg.additionalEqualitiesLength);
    cfg_.additionalEqualities = eqs_.empty() ? nullptr : eqs_.data();
    cfg_.additionalEqualitiesLength = (int)eqs_.size();
    device_ = device;
    n_ = n;
    shared_ = (numTargets == 1);
    qoff_.assign(qoff, qoff + n + 1);
    toff_.assign(toff, toff + numTargets + 1);
    for (int u = 0; u < n; ++u) {
        if (qoff_[u + 1] < qoff_[u] || qoff_[u + 1] - qoff_[u] > 0x7fffffffLL) { set_error("bad query offsets"); return 1; }
    }
    for (int u = 0; u < numTargets; ++u) {
        if (toff_[u + 1] < toff_[u] || toff_[u + 1] - toff_[u] > 0x7fffffffLL) { set_error("bad target offsets"); return 1; }
    }
    const long long qbytes = qoff_[n] - qoff_[0], tbytes = toff_[numTargets] - toff_[0];
    build_tables(tab_, reinterpret_cast<const uint8_t*>(targets) + toff_[0], tbytes,
                 eqs_.data(), (int)eqs_.size());

    pool_quarantine(false);
    DeviceGuard guard(device_);
    EDLIB_AMD_HIP(guard.status);
    EDLIB_AMD_HIP(pool_stream(&stream_));
    EDLIB_AMD_HIP(evRun0_.create()); EDLIB_AMD_HIP(evRun1_.create());

    // resident inputs (pools are rebased to offset 0).  Offsets and the small tables -- and, for small batches,
    // the sequences themselves -- go up as ONE block through pinned staging with one asynchronous copy: a call
    // of edlibAlign() is a batch of one, and nine blocking hipMemcpy calls from pageable memory were most of
    // what it cost.  Large pools are copied straight from the caller's memory.
    const long long qb = qoff_[0], tb = toff_[0];
    for (auto& v : qoff_) v -= qb;
    for (auto& v : toff_) v -= tb;
    {
        const bool inlinePools = qbytes + tbytes <= (1 << 20);
        size_t at = 0;
        auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 15) & ~(size_t)15; return o; };
        const size_t oQoff = take(qoff_.size() * sizeof(long long)), oToff = take(toff_.size() * sizeof(long long));
        const size_t oTlut = take(256), oId = take(256), oEq4 = take(512), oPres = take(32);
        const size_t oQ = inlinePools ? take((size_t)qbytes + 16) : 0, oT = inlinePools ? take((size_t)tbytes + 16) : 0;
        EDLIB_AMD_HIP(d_in_.alloc(at));
        EDLIB_AMD_HIP(h_in_.alloc(at));
        uint8_t* h = h_in_.p;
        memcpy(h + oQoff, qoff_.data(), qoff_.size() * sizeof(long long));
        memcpy(h + oToff, toff_.data(), toff_.size() * sizeof(long long));
        memcpy(h + oTlut, tab_.tlut, 256); memcpy(h + oId, tab_.idToByte, 256);
        memcpy(h + oEq4, tab_.eqtbl, 512); memcpy(h + oPres, tab_.presence, 32);
        d_qoff_.alias(d_in_.p + oQoff, qoff_.size()); d_toff_.alias(d_in_.p + oToff, toff_.size());
        d_tlut_.alias(d_in_.p + oTlut, 256); d_idToByte_.alias(d_in_.p + oId, 256);
        d_eqtbl_.alias(d_in_.p + oEq4, 256); d_presence_.alias(d_in_.p + oPres, 8);
        if (inlinePools) {
            if (qbytes) memcpy(h + oQ, queries + qb, (size_t)qbytes);
            if (tbytes) memcpy(h + oT, targets + tb, (size_t)tbytes);
            memset(h + oQ + qbytes, 0, 16); memset(h + oT + tbytes, 0, 16);
            d_qpool_.alias(d_in_.p + oQ, (size_t)qbytes + 16); d_tpool_.alias(d_in_.p + oT, (size_t)tbytes + 16);
        } else {
            EDLIB_AMD_HIP(d_qpool_.alloc((size_t)qbytes + 16));
            EDLIB_AMD_HIP(d_tpool_.alloc((size_t)tbytes + 16));
            if (qbytes) EDLIB_AMD_HIP(hipMemcpy(d_qpool_.p, queries + qb, (size_t)qbytes, hipMemcpyHostToDevice));
            if (tbytes) EDLIB_AMD_HIP(hipMemcpy(d_tpool_.p, targets + tb, (size_t)tbytes, hipMemcpyHostToDevice));
        }
        EDLIB_AMD_HIP(hipMemcpyAsync(d_in_.p, h, at, hipMemcpyHostToDevice, stream_));
    }

    // classification of the units for phase 1
    const int mode = (int)cfg_.mode;
    // reads-per-lane kernels: a shared target with at most 4 distinct bytes (every mode), or up to 16 in HW mode
    // (the banded kernel keeps 8 or 16 Peq rows per word in LDS: genomes with N, soft-masked lower case, IUPAC codes)
    const int modeIn = (int)cfg.mode;
    const bool readsOk = shared_ && tlen(0) > 0 && (tab_.sigmaT <= 4 || (tab_.sigmaT <= 16 && modeIn == EDLIB_MODE_HW));
    syms_ = tab_.sigmaT <= 4 ? 4 : (tab_.sigmaT <= 8 ? 8 : 16);
    banded_ = mode == EDLIB_MODE_HW;
    // Long HW queries against the shared target: piece filter + window verification (long_reads.hip) from kFilterFromWords
    // words on; below that the banded groups of kernel A.  257..320 / 321..384 bases on the 10- / 12-word groups: 38 / 41 / 50 /
    // 55 ms per 16,384 reads of 257 / 300 / 321 / 384 bases against 55 / 73 / 73 / 73 through the filter; from 385 on (the 14-
    // and 16-word groups: 14 / 16 KB of LDS rows per wave) the two are level on four symbols and the filter is ahead on five
    // to eight, and from 513 it wins 109 to 189.  EDLIB_AMD_FILTER=<words> moves the switch (9 = rounds 3-5: everything above
    // 256 bases), EDLIB_AMD_FILTER=0 restores round 2's routing (banded groups up to 1024 bases, kernel W above).
    static const int filterFrom = [] {
        const char* e = getenv("EDLIB_AMD_FILTER");
        if (!e || !e[0]) return kFilterFromWords;
        const int v = atoi(e);
        return v <= 0 ? 0 : std::max(kMaxReadWords + 1, std::min(v, kMaxLongReadWords4 + 1));
    }();
    const bool filter = filterFrom > 0 && readsOk && banded_ && modeIn == EDLIB_MODE_HW;
    int groupWords = kMaxReadWords;              // the tallest group of kernel A that takes reads of this batch
    if (banded_ && modeIn == EDLIB_MODE_HW && syms_ <= 8)
        for (int w : {10, 12, 14, 16, 24, 32})
            if (w <= (syms_ == 4 ? kMaxLongReadWords4 : kMaxLongReadWords) && (!filter || w < filterFrom)) groupWords = w;
    const int maxReadLen = 32 * groupWords;
    std::vector<std::vector<int>> byWords(kMaxLongReadWords4 + 1);
    for (int u = 0; u < n; ++u) {
        const int m = qlen(u), T = tlen(u);
        if (m == 0 || T == 0) emptyUnits_.push_back(u);
        else if (readsOk && m <= maxReadLen) { readUnits_.push_back(u); byWords[read_group_words(m)].push_back(u); }
        else if (filter) longUnits_.push_back(u);
        else pairUnits_.push_back(u);
    }
    stats.cells = 0;
    for (int u = 0; u < n; ++u) stats.cells += (long long)qlen(u) * tlen(u);
    {   // units whose alphabetLength the reads path does not produce
        alphaUnits_ = emptyUnits_;
        alphaUnits_.insert(alphaUnits_.end(), pairUnits_.begin(), pairUnits_.end());
        alphaUnits_.insert(alphaUnits_.end(), longUnits_.begin(), longUnits_.end());
        long long total = 0;
        for (int u : alphaUnits_) total += qlen(u) + (shared_ ? 0 : tlen(u));
        alphaOnHost_ = h_in_.p && d_qpool_.p && !d_qpool_.owned && total <= 65536;
        alphaBytes_ = total;
    }

    // reads-per-lane groups: one per query word count, slots padded to whole waves
    for (int w = 1; w <= kMaxLongReadWords4; ++w) {
        if (byWords[w].empty()) continue;
        std::unique_ptr<ReadGroup> g;
        if (makeGroup(byWords[w], w, g)) return 1;
        groups_.push_back(std::move(g));
    }
    if (initFlatPairs()) return 1;
    const int T = shared_ ? tlen(0) : 0;
    if (!groups_.empty() || !longUnits_.empty()) {
        EDLIB_AMD_HIP(d_tpk_.alloc((size_t)(T + 15) / 16 + 4));
        // the banded kernel reads whole dwords; on OUR stream: the null stream does not order with it
        EDLIB_AMD_HIP(hipMemsetAsync(d_tpk_.p, 0, d_tpk_.bytes(), stream_));
        EDLIB_AMD_HIP(d_wordSteps_.alloc(1));
        EDLIB_AMD_HIP(d_trows_.alloc(((size_t)(T + 15) / 16 + 2) * 8));
    }
    return 0;
}

// One reads-per-lane group: the units of one word count, padded to whole waves, with its resident buffers.
// oneRoundWaves > 0: the group runs on a kernel of which the chip holds that many waves without two sharing a SIMD (the
// full-height kernels of 24 / 32 words: 24 / 32 KB of LDS rows per wave) -- one launch of at most that many waves, segments as
// long as that allows (long_reads.hip, solveTallFull: the 1025th wave costs a third of the rate, every warm-up is work)
int Batch::makeGroup(const std::vector<int>& units, int w, std::unique_ptr<ReadGroup>& g, long long oneRoundWaves)
{
    const int mode = (int)cfg_.mode;
    const int T = shared_ ? tlen(0) : 0;
    g.reset(new ReadGroup);
    g->nwords = w;
    g->nslots = roundup((int)units.size(), 64);
    g->perm.assign(g->nslots, -1);
    std::copy(units.begin(), units.end(), g->perm.begin());
    const int nrblk = g->nslots / 64;
    if (mode == EDLIB_MODE_HW) {
        // enough waves to fill 256 CUs x 4 SIMDs x 8 slots many times over, segments >= 4096 columns
        // ~16 waves per resident slot: the launch ends on a thin tail (65,536 -> 131,072 waves: +1 % at 1M reads)
        const long long wantWaves = 131072;
        g->warm = 2 * 32 * w - 1;                        // 2m-1 columns
        long long S = (wantWaves + nrblk - 1) / nrblk;
        // gridDim.y limit; segments of at least 4096 columns and four warm-ups (the groups of 24 / 32 words warm up
        // over 1535 / 2047 columns: 4096-column segments were half warm-up)
        const long long maxS = std::max<long long>(1, std::min<long long>(std::min(65535, std::max(1, T / 4096)), T / (4LL * g->warm)));
        S = std::max(1LL, std::min(S, maxS));
        if (oneRoundWaves > 0 && oneRoundWaves / nrblk >= 1) S = std::min(S, oneRoundWaves / nrblk);
        g->segLen = roundup((int)((T + S - 1) / S), 16);
        g->numSegments = (T + g->segLen - 1) / g->segLen;
    } else {
        g->numSegments = 1; g->segLen = roundup(T, 16); g->warm = 0;
    }
    const size_t ns = (size_t)g->nslots, S = (size_t)g->numSegments;
    EDLIB_AMD_HIP(g->d_perm.alloc(ns));
    // (on the batch's own stream: nothing of this library runs on the null stream; perm lives as long as the group)
    EDLIB_AMD_HIP(hipMemcpyAsync(g->d_perm.p, g->perm.data(), ns * sizeof(int), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(g->d_qlen.alloc(ns)); EDLIB_AMD_HIP(g->d_kinit.alloc(ns));
    // Small groups (a call of edlibAlign() is one slot block): the merged per-slot results live in device-visible
    // pinned host memory -- the merge / Peq / census kernels write them there, nothing is downloaded, and the
    // host reads them after the one stream synchronisation a run needs anyway.
    g->zeroCopy = ns <= 1024 && pool_enabled();
    if (g->zeroCopy) {
        EDLIB_AMD_HIP(g->hostOut.alloc((ns * 20 + ns + 1) * sizeof(int)));
        int* h = reinterpret_cast<int*>(g->hostOut.p);
        memset(h, 0, (ns * 20 + ns + 1) * sizeof(int));
        g->d_best.alias(h, ns); g->d_total.alias(h + ns, ns); g->d_alphaExtra.alias(h + 2 * ns, ns);
        g->d_flags.alias(h + 3 * ns, ns + 1); g->d_pos.alias(h + 4 * ns + 1, ns * 16);
    } else
    EDLIB_AMD_HIP(g->d_alphaExtra.alloc(ns));
    EDLIB_AMD_HIP(g->d_peq.alloc(ns * (size_t)syms_ * w));
    EDLIB_AMD_HIP(g->d_segBest.alloc(ns * S)); EDLIB_AMD_HIP(g->d_segCnt.alloc(ns * S));
    EDLIB_AMD_HIP(g->d_segPos.alloc(ns * S * 8));
    if (!g->zeroCopy) {
        EDLIB_AMD_HIP(g->d_best.alloc(ns)); EDLIB_AMD_HIP(g->d_total.alloc(ns));
        EDLIB_AMD_HIP(g->d_pos.alloc(ns * 16)); EDLIB_AMD_HIP(g->d_flags.alloc(ns + 1));
    }
    return 0;
}

// the 64 KB equality matrix goes up only when a pair kernel needs it and there are additional equalities
hipError_t Batch::uploadEq8() {
    if (tab_.eq8.empty() || d_eq8_.p) return hipSuccess;
    hipError_t e = d_eq8_.alloc(65536);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(d_eq8_.p, tab_.eq8.data(), 65536, hipMemcpyHostToDevice, stream_);
}

// ------------------------------------------------------------- scan timing

void Batch::scanTimerStart() {
    if (scanEventsUsed_ == scanEvents_.size()) {
        hipEvent_t a = nullptr, b = nullptr; int dev = 0;       // run() holds the DeviceGuard: dev == device_
        (void)pool_event(&a, &dev); (void)pool_event(&b, &dev);
        scanEvents_.push_back({a, b});
    }
    (void)hipEventRecord(scanEvents_[scanEventsUsed_].first, stream_);
}
void Batch::scanTimerStop() {
    (void)hipEventRecord(scanEvents_[scanEventsUsed_].second, stream_);
    ++scanEventsUsed_;
    ++stats.scan_launches;
}

// ------------------------------------------------- result semantics (host)

// End locations of HW / SHW from the exact best bottom-row score over the target
// columns and the complete ascending list of columns attaining it:
//  * candidates are columns scoring <= min(k, m) (HW clamps k to m, edlib.cpp:566-568;
//    SHW's best never exceeds m either);
//  * the empty target prefix (position -1, score m) takes part exactly when the
//    reference's padded last block would see it, i.e. when W = 64*ceil(m/64)-m > 0
//    (edlib.cpp:661,670,681-693; oracle-verified: m=64 all-mismatch has no -1).
void finalize_semiglobal(UnitResult& r, int kcfg, int m, int best, const int* pos, long long npos) {
    const int W = ((m + 63) / 64) * 64 - m;
    const bool kAllowsM = (kcfg < 0 || kcfg >= m);
    r.ends.clear();
    if (best < 0) {
        if (W > 0 && kAllowsM) { r.editDistance = m; r.ends.push_back(-1); r.hasEnds = true; }
        else { r.editDistance = -1; r.hasEnds = false; }
        return;
    }
    r.editDistance = best;
    r.hasEnds = true;
    if (W > 0 && best == m) r.ends.push_back(-1);
    r.ends.append(pos, (size_t)npos);
}

// a recycled record of the run before last, as a fresh one
void blank_record(UnitResult& r) {
    r.status = EDLIB_STATUS_OK; r.editDistance = -1; r.alphabetLength = 0;
    r.hasEnds = r.hasStarts = r.hasAlignment = false;
    r.ends.clear(); r.starts.clear(); r.opsView = nullptr; r.opsViewLen = 0;
}

void finalize_global(UnitResult& r, int kcfg, int mode, int T, int score) {
    if (kcfg >= 0 && score > kcfg) { r.editDistance = -1; r.hasEnds = false; return; }   // edlib.cpp:744-747, 917
    r.editDistance = score;
    if (mode == EDLIB_MODE_NW) { r.hasEnds = true; r.ends.assign(1, T - 1); }              // edlib.cpp:221-225
    else r.hasEnds = false;    // unknown mode: distance as NW, no end location
}

// alphabetLength of the units the reads path does not cover (reference transformSequences, edlib.cpp:1417-1462:
// the number of distinct bytes of query and target).  It depends on the sequences only, not on any scan, so it runs
// on a side stream next to phase 1 and is collected after it.
// the side stream and the buffers of the count (once per batch)
int Batch::alphabetBuffers()
{
    const size_t n = alphaUnits_.size();
    if (!side_) EDLIB_AMD_HIP(pool_stream(&side_));
    EDLIB_AMD_HIP(evA_.create());
    if (!d_alphaIdx_.p) {
        EDLIB_AMD_HIP(d_alphaIdx_.alloc(n)); EDLIB_AMD_HIP(d_alphaOut_.alloc(n)); EDLIB_AMD_HIP(alphaPin_.alloc(n * sizeof(int)));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_alphaIdx_.p, alphaUnits_.data(), n * sizeof(int), hipMemcpyHostToDevice, side_));
    }
    return 0;
}

int Batch::alphabetLengthsBegin(hipEvent_t after)
{
    alphaPending_ = false;
    if (alphaUnits_.empty() || alphaOnHost_) return 0;
    const size_t n = alphaUnits_.size();
    if (alphabetBuffers()) return 1;
    // the inputs went up on stream_ (init): the side stream starts behind whatever stream_ holds now
    EDLIB_AMD_HIP(hipEventRecord(evA_.e, stream_));
    EDLIB_AMD_HIP(hipStreamWaitEvent(side_, evA_.e, 0));
    if (after) EDLIB_AMD_HIP(hipStreamWaitEvent(side_, after, 0));
    if (alphaBytes_ <= 4096LL * (long long)n)                        // short sequences: a wave per unit
        hipLaunchKernelGGL(alphabet_count_short_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, side_,
                           d_qpool_.p, d_qoff_.p, d_tpool_.p, d_toff_.p, shared_ ? 1 : 0, d_presence_.p,
                           d_alphaIdx_.p, (int)n, d_alphaOut_.p);
    else
    hipLaunchKernelGGL(alphabet_count_kernel, dim3((unsigned)n), dim3(256), 0, side_,
                       d_qpool_.p, d_qoff_.p, d_tpool_.p, d_toff_.p, shared_ ? 1 : 0, d_presence_.p,
                       d_alphaIdx_.p, d_alphaOut_.p);
    EDLIB_AMD_HIP(hipGetLastError());
    EDLIB_AMD_HIP(evB_.create());
    EDLIB_AMD_HIP(hipEventRecord(evB_.e, side_));                   // (a flat batch's collection reads d_alphaOut_ on stream_ behind this)
    EDLIB_AMD_HIP(hipMemcpyAsync(alphaPin_.p, d_alphaOut_.p, n * sizeof(int), hipMemcpyDeviceToHost, side_));
    alphaPending_ = true;
    return 0;
}

int Batch::alphabetLengthsEnd(std::vector<UnitResult>& res)
{
    if (alphaUnits_.empty()) return 0;
    if (alphaOnHost_) {
        // a handful of short sequences (edlibAlign() on a pair): counting distinct bytes on the host costs less than
        // a launch and a round trip; the sequences are still in the staging block of init()
        const uint8_t* hq = h_in_.p + (d_qpool_.p - d_in_.p);
        const uint8_t* ht = h_in_.p + (d_tpool_.p - d_in_.p);
        for (int u : alphaUnits_) {
            bool seen[256] = {false};
            int cnt = 0;
            auto add = [&](const uint8_t* p, long long len) { for (long long i = 0; i < len; ++i) if (!seen[p[i]]) { seen[p[i]] = true; ++cnt; } };
            add(hq + qoff_[u], qlen(u));
            if (!shared_) add(ht + toff_[u], tlen(u));
            else for (int b = 0; b < 256; ++b) if ((tab_.presence[b >> 5] >> (b & 31)) & 1u) { if (!seen[b]) { seen[b] = true; ++cnt; } }
            res[u].alphabetLength = cnt;
        }
        return 0;
    }
    if (!alphaPending_) return 0;
    EDLIB_AMD_HIP(hipStreamSynchronize(side_));
    alphaPending_ = false;
    const int* out = reinterpret_cast<const int*>(alphaPin_.p);
    for (size_t i = 0; i < alphaUnits_.size(); ++i) res[alphaUnits_[i]].alphabetLength = out[i];
    return 0;
}


// --------------------------------------------------------------------- run

// Every failure path of a run ends here: the device's wide gate goes back (another batch on the device -- a resident Python
// batch kept after an error, a second host thread -- would otherwise wait for it for ever), behind a synchronisation so
// that no spinning launch of this batch is still on the device when the next one takes the gate.
int Batch::run()
{
    const int rc = runImpl();
    if (rc && wideGateHeld_) {
        if (stream_) { DeviceGuard guard(device_); (void)hipStreamSynchronize(stream_); (void)hipGetLastError(); }
        wideGateRelease();
    }
    return rc;
}

int Batch::runImpl()
{
    pool_quarantine(false);
    Lap lap;
    DeviceGuard guard(device_);
    EDLIB_AMD_HIP(guard.status);
    const long long cells = stats.cells;
    stats = EdlibAmdBatchStats{};
    stats.cells = cells;
    scanEventsUsed_ = 0;
    haveResults_ = false;
    opsKeep_.clear();            // (the previous run's views die with the reset of their records below)
    knownSplits_.clear();
    wideGateRelease();           // (a run that failed between a wide launch and its check)
    wideSerial_ = false;         // every run tries the pipelined strips first
    viewReady_ = false; cigar_[0].ready = cigar_[1].ready = false; lastRunFlat_ = false;      // (views of the previous run end here)
    opsOwned_.clear();
    // TASK_DISTANCE over reads-path units only: nothing is assembled on the host until results() asks for it, so
    // the per-unit records (160 bytes each) are not even allocated in the timed run
    const bool lazy = (cfg_.task == EDLIB_TASK_DISTANCE && pairUnits_.empty() && longUnits_.empty() && emptyUnits_.empty() && !groups_.empty()) || flatPairs_;
    pairsCollected_ = true;
    // the records of the run before last are recycled (no 160-byte-per-unit allocation + page faults per run)
    std::vector<UnitResult>& res = work_;
    bool deferReset = false;
    deferReadsReset_ = false;
    if (lazy) res.clear();
    else {
        const size_t keep = std::min(res.size(), (size_t)n_);
        res.resize((size_t)n_);
        // a batch of pair units only rewrites every record in its finalize loop: the recycled records are blanked there, in
        // the same pass over the 16 MB of 100,000 records, instead of in a walk of their own (0.4 ms)
        deferReset = emptyUnits_.empty() && groups_.empty() && longUnits_.empty() && !flatPairs_ && pairUnits_.size() == (size_t)n_;
        // the same for a batch of reads-path units only whose records are assembled in this run (LOC / PATH): collectGroup
        // visits every one of them (15 ms of a 1M-read run were this walk over 160 MB)
        deferReadsReset_ = !deferReset && cfg_.task != EDLIB_TASK_DISTANCE && emptyUnits_.empty() && pairUnits_.empty() &&
                           longUnits_.empty() && !flatPairs_ && readUnits_.size() == (size_t)n_;
        if (!deferReset && !deferReadsReset_) for (size_t u = 0; u < keep; ++u) blank_record(res[u]);
    }
    // (results_ keeps the previous run's records until the swap at the end: they are the NEXT run's recycled `work_`, blanked
    // before they are filled; destroying and re-creating 100,000 of them was 0.4 ms of every config-4 step.  Nothing reads
    // them meanwhile: haveResults_ is false until this run has succeeded.)
    const int mode = (int)cfg_.mode;
    const int scanMode = (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) ? mode : EDLIB_MODE_NW;
    EDLIB_AMD_HIP(hipEventRecord(evRun0_.e, stream_));
    if (d_ringSteps_.p) EDLIB_AMD_HIP(hipMemsetAsync(d_ringSteps_.p, 0, sizeof(unsigned long long), stream_));
    ringStepsUsed_ = false;
    lap("run: reset records");

    // ---- empty sequences: answered without any DP (edlib.cpp:166-184)
    for (int u : emptyUnits_) {
        UnitResult& r = res[u];
        const int m = qlen(u), T = tlen(u);
        if (mode == EDLIB_MODE_NW) { r.editDistance = std::max(m, T); r.ends.assign(1, T - 1); r.hasEnds = true; }
        else if (mode == EDLIB_MODE_SHW || mode == EDLIB_MODE_HW) { r.editDistance = m; r.ends.assign(1, -1); r.hasEnds = true; }
        else r.status = EDLIB_STATUS_ERROR;
    }
    // (a big NW distance batch of pairs builds the Peq of all its units beside the divergence probe, and its main scan waits
    // for that: the count -- 2 GB of reads for config 4 -- is queued BEHIND that build instead of next to it
    // (solveGlobalDistances).  Not behind the main scan: that holds every wave slot of the chip, and the count took 17 ms.)
    alphaDeferred_ = !flatPairs_ && cfg_.mode == EDLIB_MODE_NW && cfg_.task == EDLIB_TASK_DISTANCE && pairUnits_.size() >= 8192 &&
                     groups_.empty() && longUnits_.empty();
    if (!alphaDeferred_ && alphabetLengthsBegin()) return 1;
    bool flatDone = false;
    if (flatPairs_) {                                   // ---- a flat pair batch: everything stays on the device
        bool over = false, fell = false;
        if (runPairsFlat(over, fell)) return 1;
        if (fell) {
            // something the flat layouts do not hold (more than 16 end locations with starts / paths asked for, a band
            // level that failed): this run takes the general path from the start
            res.resize((size_t)n_);
            for (size_t u = 0; u < res.size(); ++u) blank_record(res[u]);
        } else { flatDone = true; pairsCollected_ = false; lastRunFlat_ = true; }
        lap("run: flat pairs");
    }
    // ---- phase 1: distance + end locations
    if (packTarget()) return 1;
    if (runReads()) return 1;
    readsCollected_ = groups_.empty();
    // TASK_DISTANCE leaves the reads-path results in HBM until results(); LOC/PATH need them now
    lap("run: reads scans");
    if (!readsCollected_ && cfg_.task != EDLIB_TASK_DISTANCE && collectReads(res)) return 1;
    lap("run: collect reads");
    // long HW queries against the shared target: piece filter + window verification; what it hands back (low
    // complexity, thresholds beyond a quarter of the piece) joins the pair units below
    pairNow_ = pairUnits_;
    if (!longUnits_.empty()) {
        std::vector<int> fb;
        if (solveLongReads(res, fb)) return 1;
        lap("run: long reads");
        // What the filter hands back is mostly unrelated sequence: every row of every column is needed.  Up to 1024 rows
        // (512 above four target symbols) the lane-per-read full-height kernel does that at ~20 VALU ops per 64 rows and
        // column with every lane busy; longer queries take kernel W's strips.
        const int fullMax = syms_ == 4 ? 32 * kMaxLongReadWords4 : (syms_ == 8 ? 32 * kMaxLongReadWords : 0);
        std::vector<std::vector<int>> byWords(kMaxLongReadWords4 + 1);
        std::vector<int> tall;
        for (int u : fb) {
            if (qlen(u) <= fullMax) byWords[read_group_words(qlen(u))].push_back(u);
            else if (fullMax > 0) tall.push_back(u);
            else pairNow_.push_back(u);
        }
        for (int w = 1; w <= kMaxLongReadWords4; ++w) {
            if (byWords[w].empty()) continue;
            std::unique_ptr<ReadGroup> g;
            // (many short segments here: a single strip warms up over at most 2047 columns, and 7,930 waves balance themselves
            // over the SIMDs where one round of 1,014 does not -- 513 / 768 / 1024-base reads: 110 / 125 / 166 ms against
            // 129 / 144 / 172 with makeGroup's oneRoundWaves; the chained strips of solveTallFull warm up over 2m - 1 columns
            // of the WHOLE query, which is what makes one round the better plan there)
            if (makeGroup(byWords[w], w, g)) return 1;
            stats.path |= 1;
            if (runGroupScans(*g, true) || runGroupExact(*g) || collectGroup(*g, res)) return 1;
        }
        if (!tall.empty()) {                    // taller: strips of fullMax rows on the same kernel, chained through HBM
            std::vector<int> back;
            if (solveTallFull(tall, res, back)) return 1;
            pairNow_.insert(pairNow_.end(), back.begin(), back.end());
        }
        lap("run: handed back (full height)");
    }
    if (banded_ && (!groups_.empty() || !longUnits_.empty())) {   // read back with the run's final synchronisation
        EDLIB_AMD_HIP(h_wordSteps_.alloc(sizeof(unsigned long long)));
        EDLIB_AMD_HIP(hipMemcpyAsync(h_wordSteps_.p, d_wordSteps_.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
        wordStepsPending_ = true;
    }
    if (flatDone) pairNow_.clear();
    const std::vector<int>& pairUnits_ = pairNow_;          // (shadows the member: the units of THIS run's pair phase)
    if (!pairUnits_.empty()) {
        // scratch that a run needs per unit lives in the batch: a fresh 10 MB std::vector is an mmap, its page faults
        // and a munmap (45 MB of them were 5 of the 9 ms a run over 262,144 short pairs took)
        std::vector<UnitSpec>& units = pairSpecs_;
        // (the specs depend on the batch only: a run over the same pair units as the last one keeps them)
        if (pairSpecsFor_ != pairUnits_) {
            units.resize(pairUnits_.size());
            for (size_t i = 0; i < units.size(); ++i) {
                const int u = pairUnits_[i], m = qlen(u);
                // (SHW: D[m][j] >= j - m > m >= best beyond column 2m: the rest of a long target cannot matter)
                const int T = scanMode == EDLIB_MODE_SHW ? (int)std::min<long long>(tlen(u), 2LL * m + 1) : tlen(u);
                units[i] = UnitSpec{qoff_[u], m, 1, tbase(u), T, 1,
                                    (cfg_.k < 0 || cfg_.k > m) ? m : cfg_.k};
            }
            pairSpecsFor_ = pairUnits_;
            ++pairSpecsVersion_;
        }
        SolveOut& so = soMain_;
        if (scanMode == EDLIB_MODE_NW) {
            std::vector<int>& score = scoreMain_;
            // TASK_PATH over pairs that all stay below the 1 MiB rule (edlib.cpp:1188-1190): the reference scans twice
            // (distance, then the storing scan with k = distance, :1196-1199); here a unit's first successful level
            // stores its columns and is traced back right away -- any threshold >= the distance gives the same walk
            // (every neighbour that could be "one less than here" is <= the distance, hence exact inside the band)
            bool fuse = cfg_.task == EDLIB_TASK_PATH && mode == EDLIB_MODE_NW;
            for (size_t i = 0; fuse && i < units.size(); ++i) fuse = !needs_hirschberg(units[i].qlen, units[i].tlen);
            fusedOps_.clear();
            lap("run: pair specs");
            // What the records of NW units hold besides the distance does not depend on the scan (one end location, T - 1:
            // edlib.cpp:221-225; alphabetLength from the side stream): a level that takes every unit (runLevelAll) runs this
            // pass over the 16 MB of 100,000 records while the chip scans, and only the distances are filled in behind it.
            bool prefilled = false;
            if (!fuse && mode == EDLIB_MODE_NW) {
                whileScanning_ = [&]() {
                    if (alphaIsPairsVersion_ != pairSpecsVersion_) { alphaIsPairs_ = alphaUnits_ == pairUnits_; alphaIsPairsVersion_ = pairSpecsVersion_; }
                    const int* alphaOut = nullptr;
                    if (alphaPending_ && alphaIsPairs_ && !alphaOnHost_ && hipStreamSynchronize(side_) == hipSuccess) {
                        alphaPending_ = false;
                        alphaOut = reinterpret_cast<const int*>(alphaPin_.p);
                    }
                    for (size_t i = 0; i < units.size(); ++i) {
                        UnitResult& r = res[pairUnits_[i]];
                        if (deferReset) blank_record(r);
                        r.hasEnds = true; r.ends.assign(1, units[i].tlen - 1);
                        if (alphaOut) r.alphabetLength = alphaOut[i];
                    }
                    prefilled = true;
                };
            }
            const int solved = solveGlobalDistances(units, score, fuse ? &fusedOps_ : nullptr);
            whileScanning_ = nullptr;
            if (solved) return 1;
            lap("run: global distances");
            // (alphabetLength of the same units, counted on the side stream meanwhile: set in this pass over the records
            // instead of in one of its own -- alphabetLengthsEnd() then finds nothing pending)
            if (alphaIsPairsVersion_ != pairSpecsVersion_) { alphaIsPairs_ = alphaUnits_ == pairUnits_; alphaIsPairsVersion_ = pairSpecsVersion_; }
            const int* alphaOut = nullptr;
            if (alphaPending_ && alphaIsPairs_ && !alphaOnHost_) {
                EDLIB_AMD_HIP(hipStreamSynchronize(side_));
                alphaPending_ = false;
                alphaOut = reinterpret_cast<const int*>(alphaPin_.p);
            }
            if (prefilled) {
                for (size_t i = 0; i < units.size(); ++i) {
                    UnitResult& r = res[pairUnits_[i]];
                    if (cfg_.k >= 0 && score[i] > cfg_.k) { r.editDistance = -1; r.hasEnds = false; r.ends.clear(); }      // as finalize_global
                    else r.editDistance = score[i];
                    if (alphaOut) r.alphabetLength = alphaOut[i];
                }
            } else
            for (size_t i = 0; i < units.size(); ++i) {
                UnitResult& r = res[pairUnits_[i]];
                if (deferReset) blank_record(r);
                finalize_global(r, cfg_.k, mode, units[i].tlen, score[i]);
                if (alphaOut) r.alphabetLength = alphaOut[i];
            }
            if (fuse)
                for (size_t i = 0; i < units.size(); ++i) {
                    UnitResult& r = res[pairUnits_[i]];
                    if (r.editDistance >= 0 && fusedOps_[i].p) { r.opsView = fusedOps_[i].p; r.opsViewLen = fusedOps_[i].len; r.hasAlignment = true; }
                }
        } else {
            if (solveSemiGlobal(scanMode, true, units, so)) return 1;
            for (size_t i = 0; i < units.size(); ++i) {
                UnitResult& r = res[pairUnits_[i]];
                if (deferReset) blank_record(r);
                finalize_semiglobal(r, cfg_.k, units[i].qlen, so.score[i],
                                    so.posFlat.data() + so.posStart[i], so.posStart[i + 1] - so.posStart[i]);
            }
        }
    }
    lap("run: finalize pairs");
    if (!flatDone && alphabetLengthsEnd(res)) return 1;      // alphabetLength for everything the reads path did not cover (flat pairs: at collection)
    lap("run: phase 1 (distance)");
    std::vector<int>& live = live_;            // non-empty units with a solution (only the later phases want them)
    live.clear();
    // (a flat batch has done its phases 2 and 3 on the device: its records do not exist yet)
    const bool laterPhases = !flatDone && (cfg_.task == EDLIB_TASK_LOC || cfg_.task == EDLIB_TASK_PATH);
    if (laterPhases)
        for (int u = 0; u < n_; ++u)
            if (qlen(u) > 0 && tlen(u) > 0 && res[u].editDistance >= 0) live.push_back(u);

    // ---- phase 2: start locations (edlib.cpp:228-272)
    if (laterPhases) {
        std::vector<UnitSpec>& units = startUnits_; std::vector<std::pair<int, int>>& where = startWhere_;   // capacity kept across runs
        units.clear(); where.clear();
        units.reserve(live.size() + live.size() / 8); where.reserve(live.size() + live.size() / 8);
        for (int u : live) {
            UnitResult& r = res[u];
            r.hasStarts = true;
            r.starts.assign(r.ends.size(), 0);
            if (mode != EDLIB_MODE_HW) continue;
            const int m = qlen(u);
            for (size_t j = 0; j < r.ends.size(); ++j) {
                const int e = r.ends[j];
                if (e == -1) continue;                                   // :237-249
                // reverse query against the reversed prefix target[0..e], prefix mode, k = distance
                // (:253-257); columns past m+distance cannot score <= distance, so the window stops there
                const long long win = std::min<long long>((long long)e + 1, (long long)m + r.editDistance);
                units.push_back(UnitSpec{qoff_[u] + m - 1, m, -1, tbase(u) + e, (int)win, -1, r.editDistance});
                where.push_back({u, (int)j});
            }
        }
        lap("starts: units");
        if (!units.empty()) {
            SolveOut so;
            if (solveSemiGlobal(EDLIB_MODE_SHW, false, units, so)) return 1;
            lap("starts: solve");
            for (size_t i = 0; i < units.size(); ++i) {
                UnitResult& r = res[where[i].first];
                // last reported position of the reverse scan (:260); -1 when only the empty prefix qualifies
                r.starts[where[i].second] = r.ends[where[i].second] - so.last[i];
            }
        }
    }
    lap("run: phase 2 (starts)");
    // ---- phase 3: alignment path of the first location (edlib.cpp:276-289, 1161-1213)
    if (cfg_.task == EDLIB_TASK_PATH) {
        std::vector<PathPiece> jobs; std::vector<int> where;
        jobs.reserve(live.size()); where.reserve(live.size());
        for (int u : live) {
            UnitResult& r = res[u];
            if (r.ends.empty() || r.hasAlignment) continue;         // (hasAlignment: traced back in phase 1)
            const int m = qlen(u);
            const int s = r.starts[0], e = r.ends[0];
            const int len = e - s + 1;
            if (len <= 0) {                                                                         // :1168-1175
                opsOwned_.emplace_back((size_t)m, (uint8_t)EDLIB_EDOP_INSERT);
                r.opsView = opsOwned_.back().data(); r.opsViewLen = m; r.hasAlignment = true; continue;
            }
            jobs.push_back(PathPiece{qoff_[u], m, tbase(u) + s, len, r.editDistance});
            where.push_back(u);
        }
        lap("paths: jobs");
        if (!jobs.empty()) {
            std::vector<OpsOut> ops; std::vector<int> st;
            if (solvePaths(jobs, ops, st)) return 1;
            for (size_t i = 0; i < jobs.size(); ++i) {
                UnitResult& r = res[where[i]];
                if (st[i] != EDLIB_STATUS_OK) { r.status = EDLIB_STATUS_ERROR; continue; }
                if (!ops[i].own.empty()) {
                    opsOwned_.emplace_back(std::move(ops[i].own));
                    r.opsView = opsOwned_.back().data(); r.opsViewLen = (int)opsOwned_.back().size();
                } else { r.opsView = ops[i].p; r.opsViewLen = ops[i].len; }
                r.hasAlignment = true;
            }
        }
    }
    lap("run: phase 3 (paths)");
    if (ringStepsUsed_) EDLIB_AMD_HIP(hipMemcpyAsync(h_ringSteps_.p, d_ringSteps_.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipEventRecord(evRun1_.e, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    if (ringStepsUsed_) stats.word_steps += (long long)*reinterpret_cast<unsigned long long*>(h_ringSteps_.p);
    float ms = 0;
    EDLIB_AMD_HIP(hipEventElapsedTime(&ms, evRun0_.e, evRun1_.e));
    stats.run_ms = ms;
    if (wordStepsPending_) { stats.word_steps += (long long)*reinterpret_cast<unsigned long long*>(h_wordSteps_.p); wordStepsPending_ = false; }
    for (size_t i = 0; i < scanEventsUsed_; ++i) {
        float t = 0;
        EDLIB_AMD_HIP(hipEventElapsedTime(&t, scanEvents_[i].first, scanEvents_[i].second));
        stats.scan_ms += t;
    }
    // algorithmic bytes: target + query + Peq + result header + end locations
    if ((!readsCollected_ && pairUnits_.empty() && longUnits_.empty() && emptyUnits_.empty()) || flatDone) {
        // everything is still resident on the device (reads path or flat pairs, TASK_DISTANCE): every unit is priced with
        // sigma = |target alphabet| and one end location -- a constant of the batch, summed once
        if (algoBase_ < 0) {
            algoBase_ = 0;
            for (int u = 0; u < n_; ++u) {
                const long long m = qlen(u);
                algoBase_ += tlen(u) + m + 8LL * (tab_.sigmaT + 1) * ((m + 63) / 64) + 16 + 4;
            }
        }
        stats.algo_bytes = algoBase_;
        algoDirty_ = false;
    } else algoDirty_ = true;                   // a walk over every record: done when somebody asks (finishStats)
    results_.swap(res);
    haveResults_ = true;
    lap("run: stats");
    return 0;
}

// algorithmic bytes of the last run: target + query + Peq + result header + end locations.  A walk
// over every record: done when the statistics are asked for, not in every run.
void Batch::finishStats()
{
    if (!algoDirty_) return;
    if (haveResults_ && ensureCollected()) return;   // (a DISTANCE batch that mixes reads-path and pair units: the read units' records are still on the device)
    algoDirty_ = false;
    const std::vector<UnitResult>& res = results_;
    stats.algo_bytes = 0;
    for (int u = 0; u < n_ && (size_t)u < res.size(); ++u) {
        const long long m = qlen(u);
        const long long sigma = res[u].alphabetLength ? res[u].alphabetLength : tab_.sigmaT;
        stats.algo_bytes += tlen(u) + m + 8LL * (sigma + 1) * ((m + 63) / 64) + 16
                            + 4LL * std::max<long long>(1, (long long)res[u].ends.size());
    }
}

// ------------------------------------------------------------ marshalling

static int* malloc_ints(const LocList& v) {
    int* p = static_cast<int*>(malloc(sizeof(int) * std::max<size_t>(v.size(), 1)));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(int));
    return p;
}

int Batch::results(EdlibAlignResult* out)
{
    // the failure contract of edlib_amd.h: on ANY failure every entry is blank with status ERROR (nothing to free)
    for (int u = 0; u < n_; ++u) {
        EdlibAlignResult& o = out[u];
        o.status = EDLIB_STATUS_ERROR; o.editDistance = -1; o.endLocations = nullptr; o.startLocations = nullptr;
        o.numLocations = 0; o.alignment = nullptr; o.alignmentLength = 0; o.alphabetLength = 0;
    }
    if (!haveResults_) { set_error("results() before a successful run()"); return 1; }
    // a DISTANCE run over read groups: the arrays of the device-made view (engine_flat.hip: buildReadsView) are what the
    // results are copied from -- no per-read record in between (an end-location array per result is the reference's contract)
    const bool fromView = readsViewOnDevice();
    EdlibAmdResultsView v{};
    if (fromView ? resultsView(&v) : ensureCollected()) return 1;
    std::atomic<int> oom(0);                      // a failed malloc: the unit reports EDLIB_STATUS_ERROR, the call fails
    auto marshal = [&](int lo, int hi) {
        for (int u = lo; fromView && u < hi; ++u) {
            EdlibAlignResult& o = out[u];
            o.status = v.status[u]; o.editDistance = v.editDistance[u]; o.alphabetLength = v.alphabetLength[u];
            o.endLocations = nullptr; o.startLocations = nullptr; o.numLocations = 0; o.alignment = nullptr; o.alignmentLength = 0;
            if (o.editDistance < 0) continue;
            const int nl = v.numLocations[u];
            o.endLocations = static_cast<int*>(malloc(sizeof(int) * (size_t)std::max(nl, 1)));
            if (!o.endLocations) { o.status = EDLIB_STATUS_ERROR; o.editDistance = -1; oom.store(1); continue; }
            if (nl) memcpy(o.endLocations, v.endLocations + v.locOffsets[u], (size_t)nl * sizeof(int));
            o.numLocations = nl;
        }
        for (int u = lo; !fromView && u < hi; ++u) {
            const UnitResult& r = results_[u];
            EdlibAlignResult& o = out[u];
            o.status = r.status;
            o.editDistance = r.editDistance;
            o.endLocations = nullptr; o.startLocations = nullptr; o.numLocations = 0;
            o.alignment = nullptr; o.alignmentLength = 0;
            o.alphabetLength = r.alphabetLength;
            if (r.hasEnds) { o.endLocations = malloc_ints(r.ends); o.numLocations = (int)r.ends.size(); }
            if (r.hasStarts) o.startLocations = malloc_ints(r.starts);
            if (r.hasAlignment) {
                const uint8_t* src = r.opsView;
                const size_t len = (size_t)r.opsViewLen;
                o.alignment = static_cast<unsigned char*>(malloc(std::max<size_t>(len, 1)));
                if (len && o.alignment) memcpy(o.alignment, src, len);
                o.alignmentLength = (int)len;
            }
            if ((r.hasEnds && !o.endLocations) || (r.hasStarts && !o.startLocations) || (r.hasAlignment && !o.alignment)) {
                free(o.endLocations); free(o.startLocations); free(o.alignment);
                o.endLocations = nullptr; o.startLocations = nullptr; o.alignment = nullptr;
                o.numLocations = 0; o.alignmentLength = 0; o.status = EDLIB_STATUS_ERROR; o.editDistance = -1;
                oom.store(1);
            }
        }
    };
    // one malloc per array is the reference's ownership contract (edlib.h:177-205); a million of them are worth a
    // few threads (glibc arenas are per thread; free() of a block from any thread is fine)
    if (n_ >= 65536) {
        const int nthreads = host_threads(6);
        std::vector<std::thread> th;
        th.reserve(nthreads);
        int done = 0;                                 // units [0, done) are covered by a started thread
        {
            ThreadJoiner join(th);
            try {
                for (int t = 0; t < nthreads; ++t) {
                    const int hi = (int)((long long)n_ * (t + 1) / nthreads);
                    th.emplace_back(marshal, done, hi);
                    done = hi;
                }
            } catch (const std::system_error&) {}     // thread limit: this thread does the rest
        }
        if (done < n_) marshal(done, n_);
    } else marshal(0, n_);
    if (oom.load()) {                             // all or nothing: the caller gets no half-filled array to clean up
        for (int u = 0; u < n_; ++u) {
            EdlibAlignResult& o = out[u];
            free(o.endLocations); free(o.startLocations); free(o.alignment);
            o.endLocations = nullptr; o.startLocations = nullptr; o.alignment = nullptr;
            o.numLocations = 0; o.alignmentLength = 0; o.status = EDLIB_STATUS_ERROR; o.editDistance = -1;
        }
        set_error("out of host memory while marshalling results");
        return 1;
    }
    return 0;
}

// The caller-facing arrays of a general batch, from its records (a flat batch makes them on the device: engine_flat.hip).
int Batch::buildHostView()
{
    if (viewReady_) return 0;
    if (ensureCollected()) return 1;
    const size_t n = (size_t)n_;
    long long nloc = 0, naln = 0;
    bool anyStarts = false;
    for (size_t u = 0; u < n; ++u) {
        const UnitResult& r = results_[u];
        nloc += r.hasEnds ? (long long)r.ends.size() : 0;
        naln += r.hasAlignment ? (long long)r.opsViewLen : 0;
        anyStarts = anyStarts || r.hasStarts;
    }
    viewInts_.assign(4 * n + 2 * (size_t)nloc + 2, 0); viewOffs_.assign(2 * (n + 1), 0); viewOps_.assign((size_t)naln + 1, 0);
    int* st = viewInts_.data(); int* ed = st + n; int* nl = ed + n; int* al = nl + n; int* ends = al + n; int* starts = ends + nloc;
    long long* lo = viewOffs_.data(); long long* ao = lo + n + 1;
    long long li = 0, ai = 0;
    for (size_t u = 0; u < n; ++u) {
        const UnitResult& r = results_[u];
        st[u] = r.status; ed[u] = r.editDistance; al[u] = r.alphabetLength;
        const size_t c = r.hasEnds ? r.ends.size() : 0;
        nl[u] = (int)c; lo[u] = li; ao[u] = ai;
        if (c) memcpy(ends + li, r.ends.data(), c * sizeof(int));
        for (size_t i = 0; i < c; ++i) starts[li + (long long)i] = r.hasStarts ? r.starts[i] : -1;
        li += (long long)c;
        if (r.hasAlignment && r.opsViewLen > 0) { memcpy(viewOps_.data() + ai, r.opsView, (size_t)r.opsViewLen); ai += r.opsViewLen; }
    }
    lo[n] = li; ao[n] = ai;
    view_ = EdlibAmdResultsView{};
    view_.numUnits = n_; view_.status = st; view_.editDistance = ed; view_.numLocations = nl; view_.alphabetLength = al;
    view_.locOffsets = lo; view_.endLocations = ends; view_.startLocations = anyStarts ? starts : nullptr;
    view_.alnOffsets = ao; view_.alignment = cfg_.task == EDLIB_TASK_PATH ? viewOps_.data() : nullptr;
    viewAlnDev_ = nullptr; viewAlnOffDev_ = nullptr;
    viewReady_ = true;
    return 0;
}

int Batch::resultsView(EdlibAmdResultsView* out)
{
    if (!haveResults_) { set_error("results before a successful run()"); return 1; }
    DeviceGuard guard(device_);
    EDLIB_AMD_HIP(guard.status);
    if (!viewReady_ && (lastRunFlat_ ? buildFlatView() : (readsViewOnDevice() ? buildReadsView() : buildHostView()))) return 1;
    if (out) *out = view_;
    return 0;
}

// Flat form of results() as malloc'd copies of the view (edlibAmdBatchResultsFlat).
int Batch::resultsFlat(int* status, int* editDistance, int* numLocations, int* alphabetLength,
                       long long* locOffsets, int** endLocations, int** startLocations,
                       long long* alnOffsets, unsigned char** alignment)
{
    if (endLocations) *endLocations = nullptr;
    if (startLocations) *startLocations = nullptr;
    if (alignment) *alignment = nullptr;
    EdlibAmdResultsView v;
    if (resultsView(&v)) return 1;
    const size_t n = (size_t)n_;
    if (status) memcpy(status, v.status, n * sizeof(int));
    if (editDistance) memcpy(editDistance, v.editDistance, n * sizeof(int));
    if (numLocations) memcpy(numLocations, v.numLocations, n * sizeof(int));
    if (alphabetLength) memcpy(alphabetLength, v.alphabetLength, n * sizeof(int));
    if (locOffsets) memcpy(locOffsets, v.locOffsets, (n + 1) * sizeof(long long));
    if (alnOffsets) memcpy(alnOffsets, v.alnOffsets, (n + 1) * sizeof(long long));
    const long long nloc = v.locOffsets[n], naln = v.alnOffsets[n];
    int* ends = endLocations ? static_cast<int*>(malloc(sizeof(int) * (size_t)std::max<long long>(nloc, 1))) : nullptr;
    int* starts = (startLocations && v.startLocations) ? static_cast<int*>(malloc(sizeof(int) * (size_t)std::max<long long>(nloc, 1))) : nullptr;
    unsigned char* aln = alignment ? static_cast<unsigned char*>(malloc((size_t)std::max<long long>(naln, 1))) : nullptr;
    if ((endLocations && !ends) || (startLocations && v.startLocations && !starts) || (alignment && !aln)) {
        free(ends); free(starts); free(aln); set_error("out of memory"); return 1;
    }
    if (ends && nloc) memcpy(ends, v.endLocations, (size_t)nloc * sizeof(int));
    if (starts && nloc) memcpy(starts, v.startLocations, (size_t)nloc * sizeof(int));
    if (aln && naln && v.alignment) memcpy(aln, v.alignment, (size_t)naln);
    if (endLocations) *endLocations = ends;
    if (startLocations) *startLocations = starts;
    if (alignment) *alignment = aln;
    return 0;
}

// edlibAlignmentToCigar (edlib.cpp:303-350) over every op string of the last run.  A flat batch's op bytes are dense on the
// device: lengths, a prefix sum and the strings are three launches there (flat_results.hip) and one block comes back;
// other batches run-length encode their view on the host.
int Batch::cigarView(int format, const char** chars, const long long** offsets)
{
    if (format != EDLIB_CIGAR_STANDARD && format != EDLIB_CIGAR_EXTENDED) { set_error("unknown CIGAR format"); return 1; }
    EdlibAmdResultsView v;
    if (resultsView(&v)) return 1;
    CigarOut& c = cigar_[format == EDLIB_CIGAR_STANDARD ? 1 : 0];
    const size_t n = (size_t)n_;
    cigarSticky_ = true;
    if (!c.ready) {
        DeviceGuard guard(device_);
        EDLIB_AMD_HIP(guard.status);
        if (viewAlnDev_ && v.alignment) {
            const int f = format == EDLIB_CIGAR_STANDARD ? 1 : 0;
            const size_t cap = 2 * (size_t)v.alnOffsets[n] + n + 64;
            if (enqueueCigars(f, viewAlnDev_, viewAlnOffDev_, cap, stream_)) return 1;
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            if (fetchCigars(f, stream_)) return 1;
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            c.p = reinterpret_cast<const char*>(c.chars.p); c.off = reinterpret_cast<const long long*>(c.offs.p);
        } else {
            static const char ext[4] = {'=', 'I', 'D', 'X'}, stdc[4] = {'M', 'I', 'D', 'M'};
            const char* letters = format == EDLIB_CIGAR_STANDARD ? stdc : ext;
            c.hostChars.clear(); c.hostOffs.assign(n + 1, 0);
            for (size_t u = 0; u < n; ++u) {
                c.hostOffs[u] = (long long)c.hostChars.size();
                const long long a0 = v.alnOffsets[u], a1 = v.alnOffsets[u + 1];
                long long i = a0;
                while (v.alignment && i < a1) {
                    const unsigned char op = v.alignment[i];
                    if (op > 3) { set_error("CIGAR: invalid op code"); return 1; }
                    const char ch = letters[op];
                    long long run = 0;
                    while (i < a1 && v.alignment[i] <= 3 && letters[v.alignment[i]] == ch) { ++run; ++i; }
                    char buf[24];
                    const int w = snprintf(buf, sizeof buf, "%lld", run);
                    c.hostChars.insert(c.hostChars.end(), buf, buf + w);
                    c.hostChars.push_back(ch);
                }
                c.hostChars.push_back('\0');
            }
            c.hostOffs[n] = (long long)c.hostChars.size();
            c.p = c.hostChars.data(); c.off = c.hostOffs.data();
        }
        c.ready = true;
    }
    if (chars) *chars = c.p;
    if (offsets) *offsets = c.off;
    return 0;
}

// ----------------------------------------------------------------- one pair

int align_one(const char* q, int qn, const char* t, int tn, EdlibAlignConfig cfg, EdlibAlignResult* out)
{
    const long long qoff[2] = {0, qn}, toff[2] = {0, tn};
    Lap lap;
    int rc;
    {
        Batch b;
        if (b.init(q, qoff, 1, t, toff, 1, cfg, default_device())) return 1;
        lap("one: init");
        if (b.run()) return 1;
        lap("one: run");
        rc = b.results(out);
        lap("one: results");
    }
    lap("one: destroy");
    return rc;
}

}  // namespace edlib_amd
