// engine_reads.hip -- host side of the reads-per-lane path: many short queries against one shared target.
// The reference's k-doubling loop around myersCalcEditDistanceSemiGlobal (edlib.cpp:197-217, 550-704), once per BATCH:
// Peq rows, a probe that picks the first threshold and the levels in between, the banded first pass, the full-height pass
// over what it left open, the exact second pass for end-location lists that did not fit, and the collection.
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace edlib_amd {

// overflow census of the reads path: how many slots need the exact second pass
__global__ void __launch_bounds__(256)
count_flags_kernel(const int* __restrict__ flags, int n, int* __restrict__ counter)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) atomicAdd(counter, 1);
}

// full-height pass of the reads path: a lane's threshold drops to what a scan of the target's first columns found
__global__ void __launch_bounds__(256)
seed_thresholds_kernel(int* __restrict__ kinit, const int* __restrict__ best, const int* __restrict__ cnt, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cnt[i] > 0 && best[i] < kinit[i]) kinit[i] = best[i];
}

// ------------------------------------------------------- reads-per-lane path

// One scan launch over a group's slots (or a subset through d_slotmap), banded or not.
int Batch::scanGroup(ReadGroup& g, int mode, const int* d_slotmap, int nlanes, int kcap, const int* d_kinit,
                     int numSegments, int segLen, int warm, int* segBest, int* segCnt, int* segPos, int cap,
                     const long long* posOff, const int* posCap, bool unbanded, unsigned long long* wordSteps,
                     const uint32_t* peqDense, const int* qlenDense)
{
    ReadScanArgs a{};
    // peqDense / qlenDense (+ d_kinit): rows rebuilt for exactly the lanes of this launch, in lane order (pass 2)
    a.peq = peqDense ? peqDense : g.d_peq.p; a.tpk = d_tpk_.p; a.trows = d_trows_.p; a.targetLength = tlen(0);
    // SHW (prefix mode: row -1 is 0, 1, 2, ...): D[m][j] >= j - m, and the best score never exceeds m (the empty prefix), so
    // no column beyond 2m can tie it -- the scan stops there instead of walking the whole shared target
    if (mode == EDLIB_MODE_SHW) a.targetLength = (int)std::min<long long>(a.targetLength, 64LL * g.nwords + 1);
    a.qlen = qlenDense ? qlenDense : g.d_qlen.p; a.kinit = d_kinit; a.slotmap = d_slotmap; a.nlanes = nlanes;
    a.numSegments = numSegments; a.segLen = segLen; a.warm = warm;
    a.segBest = segBest; a.segCnt = segCnt; a.segPos = segPos; a.cap = cap;
    a.posOff = posOff; a.posCap = posCap;
    a.kcap = kcap; a.wordSteps = wordSteps ? wordSteps : d_wordSteps_.p;
    a.filter = filterScan_ ? 1 : 0;
    a.chainIn = chain_.in; a.chainOut = chain_.out; a.chainSrc = chain_.src; a.chainInLanes = chain_.inLanes;
    a.chainBlocks = chain_.blocks; a.rowBase = chain_.rowBase;
    const bool chained = chain_.in != nullptr || chain_.out != nullptr;
    static const bool dbg = getenv("EDLIB_AMD_DEBUG") != nullptr;
    if (dbg) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        fprintf(stderr, "[edlib_amd] scanGroup nwords=%d mode=%d nlanes=%d S=%d segLen=%d warm=%d cap=%d kcap=%d slotmap=%p posOff=%p\n",
                g.nwords, mode, nlanes, numSegments, segLen, warm, cap, kcap, (const void*)d_slotmap, (const void*)posOff);
    }
    scanTimerStart();
    // full-height HW scans (pass 2 over unrelated reads): scan_reads_kernel for four symbols (register-resident rows: 288 ms
    // per 1M-read step; the full-height kernel with LDS rows picked by M0 took 314 ms there, the banded kernel at full
    // height 325 ms: measured in round 2, the variants are gone), scan_reads_full_kernel above four symbols and for the
    // long word groups
    const bool longGroup = g.nwords > kMaxReadWords;                  // no plain kernel for 12 / 16 words
    // columns a lane walks: the segments' own columns (a launch may cover a prefix of the target only) and their warm-ups
    const long long colsScanned = std::min<long long>(a.targetLength, (long long)numSegments * segLen) + (long long)(numSegments - 1) * warm;
    const bool fullHeight = banded_ && mode == EDLIB_MODE_HW && unbanded && (chained || syms_ > 4 || longGroup);
    if (fullHeight) {
        EDLIB_AMD_HIP(launch_scan_reads_full(g.nwords, syms_, a, stream_));
        stats.word_steps += (long long)((nlanes + 63) / 64 * 64) * g.nwords * colsScanned;
    } else if (banded_ && mode == EDLIB_MODE_HW && (!unbanded || syms_ > 4 || longGroup)) EDLIB_AMD_HIP(launch_scan_reads_banded(g.nwords, syms_, a, stream_));
    else {
        EDLIB_AMD_HIP(launch_scan_reads(g.nwords, mode, a, stream_));
        stats.word_steps += (long long)((nlanes + 63) / 64 * 64) * g.nwords * colsScanned;
    }
    scanTimerStop();
    if (dbg) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        fprintf(stderr, "[edlib_amd] scanGroup done\n");
    }
    return 0;
}

// segmentation of a launch over `nlanes` lanes: enough waves to fill the chip, segments >= 4096 columns
void plan_segments(int nlanes, int T, int mode, int warmFull, long long wantWaves,
                          int& S, int& segLen, int& warm, int minSegCols)
{
    S = 1; segLen = roundup(T, 16); warm = 0;
    if (mode != EDLIB_MODE_HW) return;
    const long long nrblk = ((long long)nlanes + 63) / 64;
    long long want = (wantWaves + nrblk - 1) / nrblk;
    want = std::max(1LL, std::min<long long>(want, std::min(65535, std::max(1, T / minSegCols))));   // gridDim.y limit
    if (warmFull > 0) want = std::max(1LL, std::min<long long>(want, std::max<long long>(1, T / (4LL * warmFull))));   // >= four warm-ups per segment
    segLen = roundup((int)((T + want - 1) / want), 16);
    S = (T + segLen - 1) / segLen;
    warm = warmFull;
}

int Batch::runReads()
{
    if (groups_.empty()) return 0;
    stats.path |= 1;
    for (auto& gp : groups_) if (runGroupScans(*gp, false)) return 1;
    for (auto& gp : groups_) if (runGroupExact(*gp)) return 1;
    return 0;
}

// The scans of one group: Peq rows, then the k-doubling levels (fullOnly: one pass at the full threshold on the
// full-height kernel -- units the piece filter handed back, whose band is the whole query).
int Batch::runGroupScans(ReadGroup& g, bool fullOnly)
{
    const int T = tlen(0);
    // unknown mode values are computed as NW (edlib.cpp:205-215)
    const int mode = (cfg_.mode == EDLIB_MODE_HW || cfg_.mode == EDLIB_MODE_SHW) ? (int)cfg_.mode : (int)EDLIB_MODE_NW;
    const bool banded = banded_ && mode == EDLIB_MODE_HW;
    const int kNoCap = 0x3fffffff;
    static const bool dbgLadder = getenv("EDLIB_AMD_DEBUG") != nullptr;
    EDLIB_AMD_HIP(launch_build_peq_reads(g.nwords, syms_, d_qpool_.p, d_qoff_.p, g.d_perm.p, g.nslots,
                                         d_eqtbl_.p, d_presence_.p, cfg_.k, g.d_peq.p, g.d_qlen.p,
                                         g.d_kinit.p, g.d_alphaExtra.p, stream_));
    // ---- pass 1: all slots; banded: threshold min(k, kFirst)
    // first threshold of the k-doubling (edlib.cpp:197-217 starts at 64): 8 up to 512 bases; the groups of 24 / 32
    // words take 12 / 16 -- at 1 % error a 1024-base read has distance ~10, and a read that fails the first level
    // pays the full 32-word height over the whole target
    const int kFirstMax = std::max(8, g.nwords / 2);
    int kFirst = kFirstMax;
    bool twoPass = !fullOnly && banded && (cfg_.k < 0 || cfg_.k > kFirst) && 32 * g.nwords > kFirst;
    std::vector<int> ladder;                    // thresholds of the banded passes between the first and the full one
    if (twoPass && g.nslots >= 16384) {
        // k-doubling only pays when most units resolve at the small threshold (pass 1 costs ~2/NWD of a
        // full scan, unresolved units then pay the full scan on top).  Probe 2048 evenly strided slots
        // first (0.2 % of the work at 1M reads) and fall back to one full-threshold pass if fewer than
        // 30 % of them resolve (e.g. noisy long-read chemistry, unrelated sequences).
        // (2048 slots are an eighth of a 16,384-read batch: 2.5 of its 28.6 ms, and 8 MB of per-segment results to walk;
        // smaller batches probe a 32nd of their slots, at least 512)
        const int np = std::max(512, std::min(2048, g.nslots / 32));
        std::vector<int> probe(np);
        for (int i = 0; i < np; ++i) probe[i] = (int)((long long)i * g.nslots / np);
        // best score of the probe slots in `map` with thresholds capped at kc (-1: nothing <= kc)
        auto probe_scan = [&](const std::vector<int>& map, int kc, std::vector<int>& bestOut) -> int {
            const int nm = (int)map.size();
            int S2, segLen2, warm2;
            plan_segments(nm, T, mode, g.warm, 16384, S2, segLen2, warm2);
            const size_t items = (size_t)nm * S2;
            DevBuf<int> d_map, d_sb, d_sc;
            EDLIB_AMD_HIP(d_map.alloc(nm)); EDLIB_AMD_HIP(d_sb.alloc(items)); EDLIB_AMD_HIP(d_sc.alloc(items));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_map.p, map.data(), nm * sizeof(int), hipMemcpyHostToDevice, stream_));
            if (scanGroup(g, mode, d_map.p, nm, kc, g.d_kinit.p, S2, segLen2, warm2,
                          d_sb.p, d_sc.p, d_sb.p /*unused*/, 0, nullptr, nullptr)) return 1;
            std::vector<int> cnts(items), bests(items);
            EDLIB_AMD_HIP(hipMemcpyAsync(cnts.data(), d_sc.p, items * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipMemcpyAsync(bests.data(), d_sb.p, items * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            bestOut.assign(nm, -1);
            for (int i = 0; i < nm; ++i) {
                int b = 0x7fffffff;
                for (int sg = 0; sg < S2; ++sg)
                    if (cnts[(size_t)i * S2 + sg] > 0) b = std::min(b, bests[(size_t)i * S2 + sg]);
                if (b <= kc) bestOut[i] = b;
            }
            return 0;
        };
        std::vector<int> pbest;
        if (probe_scan(probe, kFirst, pbest)) return 1;
        int resolved = 0, real = 0;
        std::vector<int> hist(kFirstMax + 1, 0);                  // distances of the resolved probe reads
        std::vector<int> open;                                    // probe slots with nothing <= kFirstMax
        for (int i = 0; i < np; ++i) {
            if (g.perm[probe[i]] < 0) continue;
            ++real;
            if (pbest[i] >= 0) { ++resolved; ++hist[pbest[i]]; }
            else if (i == 0 || probe[i] != probe[i - 1]) open.push_back(probe[i]);
        }
        // The band of pass 1 is one 32-row word while the score 32 rows down stays above k + 4; against
        // unrelated sequence that score hovers around 13, so every unit of k below 8 keeps the second
        // word out more often.  Take the smallest threshold (>= 4) that still resolves 99.5 % of what 8
        // resolves: the few reads above it just join pass 2.
        if (resolved > 0) {
            int acc = 0, kq = kFirstMax;
            for (int d = 0; d <= kFirstMax; ++d) { acc += hist[d]; if (acc * 1000LL >= resolved * 995LL) { kq = d; break; } }
            kFirst = std::max(4, std::min(kFirstMax, kq));
        }
        if (real > 0 && resolved * 10 < real * 3) twoPass = false;
        // ---- the levels between the first and the full threshold (the reference doubles k: edlib.cpp:197-217).  When
        // more than a tenth of the probe is still open, the open probe reads are scanned once more with thresholds
        // capped at 64: their distances say which intermediate thresholds pay.  A level at threshold t costs every
        // read that reaches it a band of about 1 + (t - 6) / 8 words per column; it pays when what it resolves
        // would otherwise meet a taller band.  All subsets of {12, 16, 24, 32, 48, 64} are priced; reads at
        // Illumina-like error rates (leftovers = unrelated sequence) keep the two levels they always had.
        if (twoPass && (int)open.size() * 10 > real && open.size() >= 32) {
            const int kTop = std::min(64, 32 * g.nwords - 1);
            std::vector<int> obest;
            if (probe_scan(open, kTop, obest)) return 1;
            static const int cand[6] = {12, 16, 24, 32, 48, 64};
            auto words = [&](int t) { return std::min<double>(g.nwords, 1.0 + std::max(0, t - 6) / 8.0); };
            auto frac_le = [&](int t) {                           // share of the open reads with distance <= t
                size_t c = 0;
                for (int b : obest) if (b >= 0 && b <= t) ++c;
                return (double)c / (double)obest.size();
            };
            double bestCost = 1e30; int bestMask = 0;
            for (int mask = 0; mask < 64; ++mask) {
                double cost = 0.0, reach = 1.0; bool ok = true;
                for (int q = 0; q < 6; ++q) {
                    if (!((mask >> q) & 1)) continue;
                    if (cand[q] <= kFirst || cand[q] > kTop) { ok = false; break; }
                    cost += reach * words(cand[q]);
                    reach = 1.0 - frac_le(cand[q]);
                }
                if (!ok) continue;
                cost += reach * g.nwords;                         // what is left takes the full threshold
                if (cost < bestCost - 1e-9) { bestCost = cost; bestMask = mask; }
            }
            for (int q = 0; q < 6; ++q) if ((bestMask >> q) & 1) ladder.push_back(cand[q]);
            if (dbgLadder) {
                fprintf(stderr, "[edlib_amd] ladder nwords=%d kFirst=%d open=%zu/%d levels:", g.nwords, kFirst, open.size(), real);
                for (int t : ladder) fprintf(stderr, " %d(%.2f)", t, frac_le(t));
                fprintf(stderr, " full\n");
            }
        }
    }
    if (scanGroup(g, mode, nullptr, g.nslots, twoPass ? kFirst : kNoCap, g.d_kinit.p, g.numSegments, g.segLen,
                  g.warm, g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, 8, nullptr, nullptr, /*unbanded=*/fullOnly)) return 1;
    EDLIB_AMD_HIP(launch_merge_segments(g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, g.numSegments, 8,
                                        g.nslots, nullptr, 16, g.d_best.p, g.d_total.p, g.d_pos.p,
                                        g.d_flags.p, stream_));
    // ---- the next levels (k-doubling): slots with nothing <= the last threshold are rescanned with the next one,
    // the last time with their full threshold
    ladder.push_back(kNoCap);
    int kDone = kFirst;
    for (size_t lv = 0; twoPass && lv < ladder.size(); ++lv) {
        const int kcapL = ladder[lv];
        const bool last = kcapL == kNoCap;
        PinBuf totalPin;                               // pinned: the copy runs at link rate
        const int* total = g.d_total.p;
        if (!g.zeroCopy) {
            EDLIB_AMD_HIP(totalPin.alloc((size_t)g.nslots * sizeof(int)));
            EDLIB_AMD_HIP(hipMemcpyAsync(totalPin.p, g.d_total.p, (size_t)g.nslots * sizeof(int), hipMemcpyDeviceToHost, stream_));
            total = reinterpret_cast<const int*>(totalPin.p);
        }
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        std::vector<int> todo;
        for (int s = 0; s < g.nslots; ++s) {
            const int u = g.perm[s];
            if (u < 0 || total[s] > 0) continue;
            if (std::min(qlen(u), cfg_.k < 0 ? 0x3fffffff : cfg_.k) > kDone) todo.push_back(s);   // its threshold min(k, m) is above what was tried
        }
        if (todo.empty()) break;
        {
            const size_t no = todo.size();
            int S2, segLen2, warm2;
            plan_segments((int)no, T, mode, g.warm, 65536, S2, segLen2, warm2);
            const size_t items = no * (size_t)S2;
            DevBuf<int> d_map, d_sb, d_sc, d_sp;
            EDLIB_AMD_HIP(d_map.alloc(no)); EDLIB_AMD_HIP(d_sb.alloc(items)); EDLIB_AMD_HIP(d_sc.alloc(items));
            EDLIB_AMD_HIP(d_sp.alloc(items * 8));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_map.p, todo.data(), no * sizeof(int), hipMemcpyHostToDevice, stream_));
            // What the last level leaves over is usually unrelated sequence whose band is the whole query; there the
            // plain full-height kernel (register-resident Peq rows, no band bookkeeping) is ~12 % faster per
            // column than the banded one.  256 strided leftovers tell: the banded kernel reports its band
            // height (word-steps), and a band above 85 % of the words sends the pass to the plain kernel.
            bool plain = false;
            if (last && no >= 4096) {
                const int np2 = 256;
                std::vector<int> sub(np2);
                for (int i = 0; i < np2; ++i) sub[i] = todo[(size_t)((long long)i * no / np2)];
                int S3, segLen3, warm3;
                plan_segments(np2, T, mode, g.warm, 16384, S3, segLen3, warm3);
                const size_t it3 = (size_t)np2 * S3;
                DevBuf<int> d_m3, d_b3, d_c3, d_p3; DevBuf<unsigned long long> d_ws;
                EDLIB_AMD_HIP(d_m3.alloc(np2)); EDLIB_AMD_HIP(d_b3.alloc(it3)); EDLIB_AMD_HIP(d_c3.alloc(it3));
                EDLIB_AMD_HIP(d_p3.alloc(it3 * 8)); EDLIB_AMD_HIP(d_ws.alloc(1));
                EDLIB_AMD_HIP(hipMemcpyAsync(d_m3.p, sub.data(), np2 * sizeof(int), hipMemcpyHostToDevice, stream_));
                EDLIB_AMD_HIP(hipMemsetAsync(d_ws.p, 0, sizeof(unsigned long long), stream_));
                if (scanGroup(g, mode, d_m3.p, np2, kNoCap, g.d_kinit.p, S3, segLen3, warm3,
                              d_b3.p, d_c3.p, d_p3.p, 8, nullptr, nullptr, false, d_ws.p)) return 1;
                unsigned long long ws = 0;
                EDLIB_AMD_HIP(hipMemcpyAsync(&ws, d_ws.p, sizeof ws, hipMemcpyDeviceToHost, stream_));
                EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
                const double cols = (double)np2 * ((double)T + (double)(S3 - 1) * warm3);
                plain = (double)ws >= 0.85 * g.nwords * cols;
                stats.word_steps += (long long)ws;
            }
            // The leftovers are scattered over the batch: through the slot map every lane of a wave would pull its
            // rows from a different 256-byte line (16x the bytes, once per segment: 3 GB of fetch per 1M-read step
            // in round 1).  Their rows are rebuilt in lane order instead -- the builder reads each query once.
            const size_t no64 = (no + 63) / 64 * 64;
            std::vector<int> perm2(no64, -1);
            for (size_t i = 0; i < no; ++i) perm2[i] = g.perm[todo[i]];
            DevBuf<int> d_perm2, d_qlen2, d_kinit2, d_extra2; DevBuf<uint32_t> d_peq2;
            EDLIB_AMD_HIP(d_perm2.alloc(no64)); EDLIB_AMD_HIP(d_qlen2.alloc(no64)); EDLIB_AMD_HIP(d_kinit2.alloc(no64));
            EDLIB_AMD_HIP(d_extra2.alloc(no64)); EDLIB_AMD_HIP(d_peq2.alloc(no64 * (size_t)syms_ * g.nwords));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_perm2.p, perm2.data(), no64 * sizeof(int), hipMemcpyHostToDevice, stream_));
            EDLIB_AMD_HIP(launch_build_peq_reads(g.nwords, syms_, d_qpool_.p, d_qoff_.p, d_perm2.p, (int)no64,
                                                 d_eqtbl_.p, d_presence_.p, cfg_.k, d_peq2.p, d_qlen2.p,
                                                 d_kinit2.p, d_extra2.p, stream_));
            // Every segment starts from its lane's threshold, and a lane records a position whenever its best improves: from
            // min(k, m) an unrelated read walks down ~100 improvements per segment, each a scattered 4-byte store (1.2 GB of
            // write traffic per 1M-read step in round 2).  The first columns of the target give every lane a score that
            // some column does reach; all segments start from that one (results do not depend on it: the best over
            // the whole target is at most that score, and equal scores are still recorded).
            const int seedCols = 4096;                               // (a lone wave per SIMD: 0.27 ms)
            DevBuf<int> d_b0, d_c0, d_p0;                            // (live until the synchronisation below)
            if (mode == EDLIB_MODE_HW && last && no >= 4096 && S2 > 1 && T >= 16 * seedCols) {
                EDLIB_AMD_HIP(d_b0.alloc(no)); EDLIB_AMD_HIP(d_c0.alloc(no)); EDLIB_AMD_HIP(d_p0.alloc(no * 8));
                if (scanGroup(g, mode, nullptr, (int)no, kcapL, d_kinit2.p, 1, seedCols, 0,
                              d_b0.p, d_c0.p, d_p0.p, 8, nullptr, nullptr, plain, nullptr, d_peq2.p, d_qlen2.p)) return 1;
                hipLaunchKernelGGL(seed_thresholds_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, stream_,
                                   d_kinit2.p, d_b0.p, d_c0.p, (int)no);
                EDLIB_AMD_HIP(hipGetLastError());
            }
            if (scanGroup(g, mode, nullptr, (int)no, kcapL, d_kinit2.p, S2, segLen2, warm2,
                          d_sb.p, d_sc.p, d_sp.p, 8, nullptr, nullptr, plain, nullptr, d_peq2.p, d_qlen2.p)) return 1;
            EDLIB_AMD_HIP(launch_merge_segments(d_sb.p, d_sc.p, d_sp.p, S2, 8, (int)no, d_map.p, 16,
                                                g.d_best.p, g.d_total.p, g.d_pos.p, g.d_flags.p, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));            // temporaries die here
            if (dbgLadder) fprintf(stderr, "[edlib_amd] level kcap=%d: %zu slots rescanned (plain=%d)\n", kcapL, no, (int)plain);
        }
        kDone = kcapL;
    }
    // census of slots whose end-location list did not fit (small groups: counted on the host from the pinned flags)
    if (!g.zeroCopy) {
        int* counter = g.d_flags.p + g.nslots;
        EDLIB_AMD_HIP(hipMemsetAsync(counter, 0, sizeof(int), stream_));
        hipLaunchKernelGGL(count_flags_kernel, dim3((g.nslots + 255) / 256), dim3(256), 0, stream_,
                           g.d_flags.p, g.nslots, counter);
    }
    return 0;
}

// exact second pass for the (rare) slots with more end locations than the first pass keeps.
// Their best score b is already exact, so "score <= b" selects exactly the end locations:
// (a) a counting scan over fine segments gives the number of hits of every (slot, segment),
// (b) after a prefix sum the same scan writes them to their final place.  Fine segments keep
// the pass parallel (a handful of slots still fills the chip).
int Batch::runGroupExact(ReadGroup& g)
{
    const int T = tlen(0);
    const int mode = (cfg_.mode == EDLIB_MODE_HW || cfg_.mode == EDLIB_MODE_SHW) ? (int)cfg_.mode : (int)EDLIB_MODE_NW;
    const int kNoCap = 0x3fffffff;
    const size_t ns = (size_t)g.nslots;
    g.ovfSlots.clear(); g.ovfOff.assign(1, 0);
    int novf = 0;
    if (g.zeroCopy) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        for (size_t s = 0; s < ns; ++s) novf += g.d_flags.p[s] != 0;
    }
    else {
        EDLIB_AMD_HIP(hipMemcpyAsync(&novf, g.d_flags.p + g.nslots, sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    }
    if (novf <= 0 || mode == EDLIB_MODE_NW) return 0;
    std::vector<int> flags(ns), total(ns);
    if (g.zeroCopy) memcpy(flags.data(), g.d_flags.p, ns * sizeof(int));
    else {
        EDLIB_AMD_HIP(hipMemcpyAsync(flags.data(), g.d_flags.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    }
    for (size_t s = 0; s < ns; ++s)
        if (flags[s] && g.perm[s] >= 0) g.ovfSlots.push_back((int)s);
    const size_t no = g.ovfSlots.size();
    if (!no) return 0;
    int S2, segLen2, warm2;
    // (a handful of slots is a handful of waves per segment: the pass takes what ONE wave takes for its segment, so the
    // segments go down to four warm-ups -- 18 slots of a 16,384-read batch: 2 x 0.83 ms at 4,112 columns)
    plan_segments((int)no, T, mode, g.warm, 16384, S2, segLen2, warm2, no <= 256 ? 1024 : 4096);
    const size_t items = no * (size_t)S2;
    DevBuf<int> d_map, d_caps, d_sb, d_sc; DevBuf<long long> d_off;
    EDLIB_AMD_HIP(d_map.alloc(no)); EDLIB_AMD_HIP(d_sb.alloc(items)); EDLIB_AMD_HIP(d_sc.alloc(items));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_map.p, g.ovfSlots.data(), no * sizeof(int), hipMemcpyHostToDevice, stream_));
    // (a) count; threshold = the exact best (d_best), so the band is as narrow as it gets
    if (scanGroup(g, mode, d_map.p, (int)no, kNoCap, g.d_best.p, S2, segLen2, warm2,
                  d_sb.p, d_sc.p, d_sb.p /*unused*/, 0, nullptr, nullptr)) return 1;
    std::vector<int> cnts(items);
    EDLIB_AMD_HIP(hipMemcpyAsync(cnts.data(), d_sc.p, items * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    std::vector<long long> offs(items);
    long long acc = 0;
    g.ovfOff.assign(no + 1, 0);
    for (size_t i = 0; i < no; ++i) {
        for (int sg = 0; sg < S2; ++sg) { offs[i * S2 + sg] = acc; acc += cnts[i * S2 + sg]; }
        g.ovfOff[i + 1] = acc;
    }
    EDLIB_AMD_HIP(d_caps.alloc(items)); EDLIB_AMD_HIP(d_off.alloc(items)); EDLIB_AMD_HIP(g.d_ovfPool.ensure((size_t)acc));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_caps.p, cnts.data(), items * sizeof(int), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_off.p, offs.data(), items * sizeof(long long), hipMemcpyHostToDevice, stream_));
    // (b) write
    if (scanGroup(g, mode, d_map.p, (int)no, kNoCap, g.d_best.p, S2, segLen2, warm2,
                  d_sb.p, d_sc.p, g.d_ovfPool.p, 0, d_off.p, d_caps.p)) return 1;
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));                    // temporaries die here
    stats.overflow_units += (int)no;
    return 0;
}

// the shared target in the forms the reads-per-lane kernels read (once per run; the work counter of the banded kernel)
int Batch::packTarget()
{
    if (groups_.empty() && longUnits_.empty()) return 0;
    const int T = tlen(0);
    EDLIB_AMD_HIP(hipMemsetAsync(d_wordSteps_.p, 0, sizeof(unsigned long long), stream_));
    if (syms_ == 4) EDLIB_AMD_HIP(launch_pack_target_2bit(d_tpool_.p, d_tlut_.p, T, d_tpk_.p, stream_));
    if (banded_) EDLIB_AMD_HIP(launch_pack_target_rows(d_tpool_.p, d_tlut_.p, T, d_trows_.p, (int)d_trows_.n, stream_));
    return 0;
}

int Batch::collectReads(std::vector<UnitResult>& res)
{
    if (groups_.empty()) return 0;
    for (auto& gp : groups_) if (collectGroup(*gp, res)) return 1;
    readsCollected_ = true;
    return 0;
}

// D2H of one group's merged per-slot results + the result semantics of its units
int Batch::collectGroup(ReadGroup& g, std::vector<UnitResult>& res)
{
    const int T = tlen(0);
    const int mode = (cfg_.mode == EDLIB_MODE_HW || cfg_.mode == EDLIB_MODE_SHW) ? (int)cfg_.mode : (int)EDLIB_MODE_NW;
    const size_t ns = (size_t)g.nslots;
    // merged per-slot results: read in place when they already live in pinned host memory (small groups), else
    // downloaded into pinned staging (a copy into pageable memory runs at a fraction of the link rate: 64 bytes
    // per read were 20 ms per 1M reads)
    std::vector<int> ovfPos((size_t)g.ovfOff.back());
    PinBuf stage;
    const int *best, *total, *extra, *pos;
    if (g.zeroCopy) {                          // run() synchronised the stream
        best = g.d_best.p; total = g.d_total.p; extra = g.d_alphaExtra.p; pos = g.d_pos.p;
    } else {
        EDLIB_AMD_HIP(stage.alloc(ns * 19 * sizeof(int)));
        int* h = reinterpret_cast<int*>(stage.p);
        EDLIB_AMD_HIP(hipMemcpyAsync(h, g.d_best.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(h + ns, g.d_total.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(h + 2 * ns, g.d_alphaExtra.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(h + 3 * ns, g.d_pos.p, ns * 16 * sizeof(int), hipMemcpyDeviceToHost, stream_));
        best = h; total = h + ns; extra = h + 2 * ns; pos = h + 3 * ns;
    }
    if (!ovfPos.empty())
        EDLIB_AMD_HIP(hipMemcpyAsync(ovfPos.data(), g.d_ovfPool.p, ovfPos.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    size_t oi = 0;
    for (size_t s = 0; s < ns; ++s) {
        const int u = g.perm[s];
        if (u < 0) continue;
        UnitResult& r = res[u];
        if (deferReadsReset_) blank_record(r);
        r.alphabetLength = tab_.sigmaT + extra[s];
        const int m = qlen(u);
        if (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) {
            if (oi < g.ovfSlots.size() && g.ovfSlots[oi] == (int)s) {
                finalize_semiglobal(r, cfg_.k, m, best[s], ovfPos.data() + g.ovfOff[oi], g.ovfOff[oi + 1] - g.ovfOff[oi]);
                ++oi;
            } else {
                finalize_semiglobal(r, cfg_.k, m, best[s], pos + s * 16, best[s] < 0 ? 0 : total[s]);
            }
        } else {
            finalize_global(r, cfg_.k, (int)cfg_.mode, T, best[s]);
        }
    }
    return 0;
}


}  // namespace edlib_amd
