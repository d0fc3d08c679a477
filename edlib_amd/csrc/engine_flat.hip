// engine_flat.hip -- flat pair batches: short independent pairs whose descriptors, scans, start locations, paths AND
// caller-facing result arrays all stay on the device (host side of the engine; see engine.hip for the general path).
//
// The reference's per-call phases (edlib.cpp:146-301: distance and end locations, :228-272 start locations, :276-289 the
// path of the first location) for a batch whose units are pairs of at most 16 blocks: nothing about such a batch needs a
// per-unit decision on the host, so a run is a fixed sequence of launches over resident descriptors, and collection is
// three small kernels + one block copied to pinned host memory (flat_results.hip).
#include "engine.hpp"
#include "flat_results.hpp"

#include <algorithm>
#include <cstring>

namespace edlib_amd {

// flat pair path: how many units have more end locations than their list keeps (they need the exact second pass)
__global__ void __launch_bounds__(256)
count_over_kernel(const int* __restrict__ count, int n, int cap, int* __restrict__ counter)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && count[i] > cap) atomicAdd(counter, 1);
}

// flat pair path, NW with the column store: units whose band level failed (score above the level's threshold while a
// larger one was still allowed)
__global__ void __launch_bounds__(256)
count_failed_levels_kernel(const PairDesc* __restrict__ descs, const int* __restrict__ score, int n, int kcap, int* __restrict__ counter)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PairDesc d = descs[i];
    const int whole = d.qlen > d.tlen ? d.qlen : d.tlen;
    if (score[i] > d.kinit && d.kinit < whole && d.kinit < kcap) atomicAdd(counter, 1);
}

// flat pair path, HW start locations (reference edlib.cpp:228-266): every end location e of every unit gets a reverse
// prefix scan -- reversed query against the reversed prefix target[0..e], threshold = the distance, at most m + distance
// columns (:253-257).  One thread per unit writes the descriptors of its (at most posCap) scans into slots it takes from
// a counter; slotOf[u * posCap + j] remembers which scan answers location j.
__global__ void __launch_bounds__(256)
flat_start_descs_kernel(const PairDesc* __restrict__ descs, const int* __restrict__ score, const int* __restrict__ count,
                        const int* __restrict__ pos, int n, int posCap, long long revPeqBase, int ring,
                        PairDesc* __restrict__ out, int* __restrict__ slotOf, int* __restrict__ counter, int cap)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const PairDesc d = descs[u];
    const int ed = score[u];
    int c = ed < 0 ? 0 : count[u];
    c = c > posCap ? posCap : c;
    for (int j = 0; j < posCap; ++j) {
        int slot = -1;
        if (j < c) {
            const int e = pos[(long long)u * posCap + j];
            slot = atomicAdd(counter, 1);
            if (slot < cap) {
                PairDesc x{};
                x.qoff = d.qoff + d.qlen - 1; x.qstep = -1; x.qlen = d.qlen;
                x.toff = d.toff + e; x.tstep = -1;
                const long long win = (long long)e + 1 < (long long)d.qlen + ed ? (long long)e + 1 : (long long)d.qlen + ed;
                x.tlen = (int)win; x.kinit = ed;
                x.peqOff = revPeqBase + d.peqOff;
                x.posCap = 0; x.posOff = 0; x.storeOff = 0; x.auxOff = 0; x.colOff = -1; x.bandT = 0; x.skip = 0; x.ring = ring;
                out[slot] = x;
            } else slot = -1;
        }
        slotOf[(long long)u * posCap + j] = slot;
    }
}
// start = e - (last position of the reverse scan)   (edlib.cpp:260)
__global__ void __launch_bounds__(256)
flat_starts_kernel(const int* __restrict__ slotOf, const int* __restrict__ pos, const int* __restrict__ lastOfScan, long long total,
                   int* __restrict__ starts)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int s = slotOf[i];
    starts[i] = s < 0 ? 0 : pos[i] - lastOfScan[s];
}
// flat pair path, TASK_PATH of SHW / HW units (edlib.cpp:276-289): NW of the query against target[start0 .. end0] of the
// FIRST location, with the column store.  A unit without a solution, or whose first location is the empty prefix (-1:
// the host writes its m inserts), gets an inactive descriptor (threshold below |T - m|).
__global__ void __launch_bounds__(256)
flat_path_descs_kernel(const PairDesc* __restrict__ descs, const int* __restrict__ score, const int* __restrict__ count,
                       const int* __restrict__ pos, const int* __restrict__ starts, int n, int posCap,
                       const long long* __restrict__ storeBase, int ring, PairDesc* __restrict__ out)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    PairDesc x = descs[u];
    const int m = x.qlen, ed = score[u];
    const int nb = (m + 63) >> 6, W = 64 * nb - m;
    const bool lead = W > 0 && ed == m;                              // position -1 comes first
    x.posCap = 0; x.posOff = 0; x.ring = ring; x.storeOff = storeBase[u]; x.colOff = -1; x.bandT = 0; x.skip = 0; x.auxOff = 0;
    if (ed < 0 || count[u] <= 0 || lead) { x.tlen = 1; x.kinit = -1; }
    else {
        const int e0 = pos[(long long)u * posCap];
        const int s0 = starts ? starts[(long long)u * posCap] : 0;
        x.toff += s0; x.tlen = e0 - s0 + 1;
        x.kinit = m > x.tlen ? m : x.tlen;                           // every block sits on the ring: the whole matrix
    }
    out[u] = x;
}

// ------------------------------------------------------------ flat pair path

static const int kFlatPosCap = 16;

// Batches of short independent pairs (the verification step of a seed-and-extend mapper: 262,144 x 150 bp in 400 bp
// windows) were host-bound: every run rebuilt 88-byte descriptors for every unit, uploaded them, downloaded 16 end
// positions per unit and walked 160-byte records five times (22..28 ns per pair with the kernels a fifth of it).  When
// every unit is a pair of at most 16 blocks and only distances are asked for, nothing about the descriptors depends on a
// run: they are built once, here, and stay resident; a run is Peq build + one ring scan (whole matrix on a 4- or 16-lane
// ring: exact for any distance, no levels) + a census of overflowing end-location lists, and the results stay in HBM until
// results() asks for them -- the lazy form the reads path has had since round 1.
PairDesc Batch::flatDesc(int u) const
{
    const int mode = (int)cfg_.mode;
    const int scanMode = (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) ? mode : EDLIB_MODE_NW;
    const int m = qlen(u);
    const int T = scanMode == EDLIB_MODE_SHW ? (int)std::min<long long>(tlen(u), 2LL * m + 1) : tlen(u);   // (SHW: nothing beyond column 2m can tie the best)
    PairDesc x{};
    x.qoff = qoff_[u]; x.toff = tbase(u); x.qlen = m; x.tlen = T; x.qstep = 1; x.tstep = 1;
    // NW: the ring holds every block of the unit, so the band is the whole matrix (threshold max(m, T)); SHW / HW:
    // columns scoring <= min(k, m) are end-location candidates
    x.kinit = scanMode == EDLIB_MODE_NW ? std::max(m, T) : ((cfg_.k < 0 || cfg_.k > m) ? m : cfg_.k);
    // NW with the column store (flatNwStore_): the first level of solveGlobalDistances -- the ring's band limit for a unit of
    // more blocks than the ring has lanes, capped by the caller's k; a unit that fails it sends the run to the general path
    if (flatNwStore_) {
        const int kcap = cfg_.k >= 0 ? cfg_.k : 0x3fffffff;
        // (words of 32 rows on 8-lane rings, ring32_kernels.hip: a query of up to 8 words sits whole on its ring; above, the
        // first band level K = 128 -- well inside what the 4-lane rings of 64-row blocks hold, and inside ring32_max_k(8) = 192)
        if (flatRing32_) x.kinit = std::min(kcap, (m + 31) / 32 <= flatG32_ ? std::max(m, T) : 128);
        else x.kinit = std::min(kcap, (m + 63) / 64 <= flatRing_ ? std::max(m, T) : ring_max_k(flatRing_));
    }
    x.peqOff = flatPeqOff_[u];
    x.storeOff = 0; x.auxOff = 0; x.posCap = scanMode == EDLIB_MODE_NW ? 0 : kFlatPosCap; x.posOff = (long long)u * kFlatPosCap;
    x.colOff = -1; x.bandT = 0; x.skip = 0; x.ring = flatRing_;
    return x;
}

int Batch::initFlatPairs()
{
    flatPairs_ = false;
    flatStarts_ = flatPaths_ = flatNwStore_ = flatRing32_ = false;
    flatOvfUnit_.clear(); flatOvfOff_.assign(1, 0);
    if (!emptyUnits_.empty() || !groups_.empty() || !longUnits_.empty()) return 0;
    if ((int)pairUnits_.size() != n_ || n_ < 1024) return 0;       // (a handful of units: the zero-copy path of solveChunk)
    const int mode = (int)cfg_.mode;
    if (mode != EDLIB_MODE_NW && mode != EDLIB_MODE_SHW && mode != EDLIB_MODE_HW) return 0;
    const int scanMode = (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) ? mode : EDLIB_MODE_NW;
    int maxBlocks = 0, maxT = 0;
    for (int u = 0; u < n_; ++u) { maxBlocks = std::max(maxBlocks, (qlen(u) + 63) / 64); maxT = std::max(maxT, tlen(u)); }
    // (long targets: the general path cuts HW targets into segments when the batch alone does not fill the chip)
    if (maxBlocks > 16 || maxT > 65536) return 0;
    flatMaxBlocks_ = maxBlocks;
    flatRing_ = maxBlocks <= 4 ? 4 : (maxBlocks <= 8 ? 8 : 16);      // (the smallest ring that holds every query whole)
    // window of a unit's alignment: the whole target (NW), at most 2 m + 1 columns (SHW: the scan stops there; HW: m + distance)
    auto window = [&](int u) { return scanMode == EDLIB_MODE_NW ? tlen(u) : (int)std::min<long long>(tlen(u), 2LL * qlen(u) + 1); };
    if (cfg_.task == EDLIB_TASK_PATH) {
        // paths stay flat when every unit's store fits a 4-lane ring: SHW / HW queries of at most 4 blocks (whole matrix of
        // the window), NW pairs of up to 16 blocks inside the first band level; never in the Hirschberg regime (:1188-1190)
        if (scanMode != EDLIB_MODE_NW && maxBlocks > 4) return 0;
        for (int u = 0; u < n_; ++u) if (needs_hirschberg(qlen(u), window(u))) return 0;
        // (the resident column store and op slots are upper bounds per unit: a batch whose bounds add up to more than a
        // slice of the HBM keeps the general path, which sizes them per chunk)
        long long storeBytes = 0, opBytes = 0, store32 = 0;
        int maxWords = 0;
        // (8-lane rings of words.  16-lane rings -- one DPP row rotation per carried word instead of two moves and a select,
        // 34 instructions per step against 38, twice the waves -- were measured at config 5: 0.195 against 0.180 ms of scan,
        // A/B on one box; the scan is not bound by its instruction count alone)
        for (int u = 0; u < n_; ++u) {
            storeBytes += 16LL * ring_store_entries(4, qlen(u), window(u));
            store32 += 8LL * ring32_store_entries(flatG32_, qlen(u), window(u));
            opBytes += qlen(u) + window(u) + 8;
            maxWords = std::max(maxWords, (qlen(u) + 31) / 32);
        }
        if (storeBytes > (32LL << 30) || opBytes > (8LL << 30)) return 0;
        flatPaths_ = true;
        flatNwStore_ = scanMode == EDLIB_MODE_NW;
        flatRing_ = 4;
        // Storing scans and walks on rings of 32-row words (ring32_kernels.hip) whenever the batch allows it: a store the
        // kernel's 32-bit offsets reach, a Peq table per unit that fits a wave's LDS.  SHW / HW paths: queries of at most
        // 4 blocks = 8 words, whole on an 8-lane ring.
        flatMaxWords_ = maxWords;
        // NW (the phase-1 scan IS the storing scan): a batch whose store exceeds what 32-bit offsets reach runs in chunks of
        // units, each scanned and walked before the next reuses the store (160,000 x 1 kb: 6.4 GB of store as two chunks)
        flatRing32_ = (flatNwStore_ || store32 < 0xF0000000LL) && ring32_lds_bytes(flatG32_, tab_.sigmaT, maxWords) <= 48 * 1024;
    }
    flatStarts_ = cfg_.task != EDLIB_TASK_DISTANCE && mode == EDLIB_MODE_HW;
    PinBuf pin;
    EDLIB_AMD_HIP(pin.alloc((size_t)n_ * sizeof(PairDesc)));
    PairDesc* d = reinterpret_cast<PairDesc*>(pin.p);
    long long peqWords = 0;
    flatPeqOff_.resize((size_t)n_);
    for (int u = 0; u < n_; ++u) {
        const long long nb = (qlen(u) + 63) / 64;
        flatPeqOff_[u] = peqWords; peqWords += nb * tab_.sigmaT;
        d[u] = flatDesc(u);
    }
    EDLIB_AMD_HIP(d_flatDescs_.alloc((size_t)n_));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_flatDescs_.p, d, (size_t)n_ * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(d_peq64_.ensure((size_t)peqWords));
    EDLIB_AMD_HIP(d_flatOut3_.alloc(3 * (size_t)n_));
    EDLIB_AMD_HIP(d_flatPos_.alloc(scanMode == EDLIB_MODE_NW ? 1 : (size_t)n_ * kFlatPosCap));
    EDLIB_AMD_HIP(d_flatCensus_.alloc(2));
    EDLIB_AMD_HIP(h_flatCensus_.alloc(2 * sizeof(int)));
    if (flatStarts_) {
        // reversed-query Peq rows (same layout as the forward ones, behind them) and their builder's descriptors
        flatRevPeqBase_ = peqWords;
        EDLIB_AMD_HIP(d_peq64_.ensure((size_t)(2 * peqWords)));
        PinBuf rp;
        EDLIB_AMD_HIP(rp.alloc((size_t)n_ * sizeof(PairDesc)));
        PairDesc* r = reinterpret_cast<PairDesc*>(rp.p);
        for (int u = 0; u < n_; ++u) { r[u] = d[u]; r[u].qoff = d[u].qoff + d[u].qlen - 1; r[u].qstep = -1; r[u].peqOff = flatRevPeqBase_ + d[u].peqOff; }
        EDLIB_AMD_HIP(d_flatRevDescs_.alloc((size_t)n_));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_flatRevDescs_.p, r, (size_t)n_ * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        flatStartCap_ = 2 * (size_t)n_ + 1024;
        EDLIB_AMD_HIP(d_flatStartDescs_.alloc(flatStartCap_));
        EDLIB_AMD_HIP(d_flatStartOut3_.alloc(3 * flatStartCap_));
        EDLIB_AMD_HIP(d_flatSlotOf_.alloc((size_t)n_ * kFlatPosCap));
        EDLIB_AMD_HIP(d_flatStartsOut_.alloc((size_t)n_ * kFlatPosCap));
    }
    if (flatPaths_) {
        // op slots (m + window + 8 bytes per unit, filled from the back) and store ranges (a 4-lane ring over the window):
        // upper bounds that depend on the batch only, laid out once
        flatOpsOffHost_.assign((size_t)n_ + 1, 0);
        std::vector<long long> storeBase((size_t)n_);
        long long entries = 0, maxEntries = 0;
        flatChunkStart_.assign(1, 0);
        const long long chunkCap = 0xE0000000LL / 8;                      // 8-byte entries a ring32 launch can address
        for (int u = 0; u < n_; ++u) {
            flatOpsOffHost_[u + 1] = flatOpsOffHost_[u] + qlen(u) + window(u) + 8;
            const long long e = flatRing32_ ? ring32_store_entries(flatG32_, qlen(u), window(u)) : ring_store_entries(flatRing_, qlen(u), window(u));
            if (flatRing32_ && flatNwStore_ && entries + e > chunkCap && entries > 0) { flatChunkStart_.push_back(u); maxEntries = std::max(maxEntries, entries); entries = 0; }
            storeBase[u] = entries;
            entries += e;
        }
        flatChunkStart_.push_back(n_);
        entries = std::max(maxEntries, entries);
        flatOpsTotal_ = flatOpsOffHost_[n_];
        EDLIB_AMD_HIP(d_flatOpsOff_.alloc((size_t)n_ + 1)); EDLIB_AMD_HIP(d_flatStoreBase_.alloc((size_t)n_));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_flatOpsOff_.p, flatOpsOffHost_.data(), ((size_t)n_ + 1) * sizeof(long long), hipMemcpyHostToDevice, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_flatStoreBase_.p, storeBase.data(), (size_t)n_ * sizeof(long long), hipMemcpyHostToDevice, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        EDLIB_AMD_HIP(d_flatOps_.alloc((size_t)flatOpsTotal_)); EDLIB_AMD_HIP(d_flatOpsLen_.alloc((size_t)n_));
        EDLIB_AMD_HIP(d_store_.ensure(flatRing32_ ? (size_t)(entries + 1) / 2 : (size_t)entries));      // (ring32: 8-byte entries)
        if (flatRing32_) EDLIB_AMD_HIP(d_tsym_.alloc(d_tpool_.n));
        if (flatNwStore_) {
            // the phase-1 descriptors ARE the storing scans: give them their store ranges
            for (int u = 0; u < n_; ++u) d[u].storeOff = storeBase[u];
            EDLIB_AMD_HIP(hipMemcpyAsync(d_flatDescs_.p, d, (size_t)n_ * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        } else {
            EDLIB_AMD_HIP(d_flatPathDescs_.alloc((size_t)n_)); EDLIB_AMD_HIP(d_flatPathOut3_.alloc(3 * (size_t)n_));
        }
    }
    // the word-steps of a run over the resident descriptors never change: counted here, once
    {
        unsigned long long* ctr = ringStepsCounter();
        if (!ctr) return 1;
        EDLIB_AMD_HIP(hipMemsetAsync(ctr, 0, sizeof(unsigned long long), stream_));
        EDLIB_AMD_HIP(launch_count_ring_steps(d_flatDescs_.p, n_, scanMode, 1, ctr, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(h_ringSteps_.p, ctr, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    }
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));                // `pin` dies here
    flatWordSteps_ = (long long)*reinterpret_cast<unsigned long long*>(h_ringSteps_.p);
    if (flatRing32_ && flatNwStore_) flatWordSteps_ = ring32_word_steps(flatG32_, d, n_);      // (phase 1 runs on the rings of 32-row words)
    ringStepsUsed_ = false;
    flatPairs_ = true;
    return 0;
}

int Batch::runPairsFlat(bool& overflowed, bool& fellBack)
{
    fellBack = false;
    const int mode = (int)cfg_.mode;
    const int scanMode = (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) ? mode : EDLIB_MODE_NW;
    stats.path |= 2;
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_flatDescs_.p, n_, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT, d_peq64_.p, stream_));
    PairScanArgs a{};
    a.descs = d_flatDescs_.p; a.numUnits = n_; a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = nullptr;
    a.peqRowStride = peq_row_stride(std::max(flatRing_, flatMaxBlocks_));
    a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
    a.store = flatNwStore_ ? d_store_.p : nullptr;
    a.outScore = d_flatOut3_.p; a.outCount = d_flatOut3_.p + n_; a.outLast = d_flatOut3_.p + 2 * (size_t)n_; a.posPool = d_flatPos_.p;
    a.wordSteps = nullptr;
    stats.word_steps += flatWordSteps_;
    const bool ring32 = flatRing32_ && flatNwStore_;
    if (flatRing32_) {                                   // the targets as symbol ids, once per run (the refills of the ring32 scans read them)
        EDLIB_AMD_HIP(launch_target_symbols(d_tpool_.p, d_tlut_.p, (long long)d_tpool_.n, d_tsym_.p, stream_));
        a.tsym = d_tsym_.p;
    }
    const bool chunked = ring32 && flatChunkStart_.size() > 2;
    if (!chunked) {
        scanTimerStart();
        if (ring32) EDLIB_AMD_HIP(launch_scan_pairs_ring32(flatG32_, true, a, flatMaxWords_, stream_));
        else EDLIB_AMD_HIP(launch_scan_pairs_ring(flatRing_, scanMode, flatNwStore_, a, stream_));
        scanTimerStop();
    }
    overflowed = false;
    if (flatNwStore_) {
        // NW paths: the distance scan was the storing scan (one band level); walk it, and see whether every unit got its answer
        TracebackArgs tb{};
        tb.descs = d_flatDescs_.p; tb.numUnits = n_; tb.score = d_flatOut3_.p; tb.store = d_store_.p;
        tb.ops = d_flatOps_.p; tb.opsOff = d_flatOpsOff_.p; tb.opsLen = d_flatOpsLen_.p;
        if (chunked) {
            // a store beyond the 32-bit offsets of the ring32 kernels: chunk by chunk, each scanned and walked before the next
            // reuses the store (store offsets are relative to the chunk: initFlatPairs)
            for (size_t c = 0; c + 1 < flatChunkStart_.size(); ++c) {
                const int u0 = flatChunkStart_[c], nu = flatChunkStart_[c + 1] - u0;
                PairScanArgs ac = a;
                ac.descs = a.descs + u0; ac.numUnits = nu;
                ac.outScore = a.outScore + u0; ac.outCount = a.outCount + u0; ac.outLast = a.outLast + u0;
                scanTimerStart();
                EDLIB_AMD_HIP(launch_scan_pairs_ring32(flatG32_, true, ac, flatMaxWords_, stream_));
                scanTimerStop();
                TracebackArgs tc = tb;
                tc.descs = tb.descs + u0; tc.numUnits = nu; tc.score = tb.score + u0; tc.opsOff = tb.opsOff + u0; tc.opsLen = tb.opsLen + u0;
                EDLIB_AMD_HIP(launch_traceback32(tc, flatG32_, stream_));
            }
        } else
        if (ring32) EDLIB_AMD_HIP(launch_traceback32(tb, flatG32_, stream_));
        else EDLIB_AMD_HIP(launch_traceback(tb, stream_));
        EDLIB_AMD_HIP(hipMemsetAsync(d_flatCensus_.p, 0, 2 * sizeof(int), stream_));
        hipLaunchKernelGGL(count_failed_levels_kernel, dim3((n_ + 255) / 256), dim3(256), 0, stream_, d_flatDescs_.p, d_flatOut3_.p, n_,
                           cfg_.k >= 0 ? cfg_.k : 0x3fffffff, d_flatCensus_.p);
        EDLIB_AMD_HIP(hipMemcpyAsync(h_flatCensus_.p, d_flatCensus_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        if (*reinterpret_cast<const int*>(h_flatCensus_.p) > 0) { fellBack = true; return 0; }     // some unit needs the next level: the general path has them
        return 0;
    }
    if (scanMode != EDLIB_MODE_NW) {
        EDLIB_AMD_HIP(hipMemsetAsync(d_flatCensus_.p, 0, sizeof(int), stream_));
        hipLaunchKernelGGL(count_over_kernel, dim3((n_ + 255) / 256), dim3(256), 0, stream_, d_flatOut3_.p + n_, n_, kFlatPosCap, d_flatCensus_.p);
        EDLIB_AMD_HIP(hipMemcpyAsync(h_flatCensus_.p, d_flatCensus_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        const int novf = *reinterpret_cast<const int*>(h_flatCensus_.p);
        flatOvfUnit_.clear(); flatOvfOff_.assign(1, 0); flatOvfPos_.clear();
        if (novf > 0) {
            // exact second pass for the (rare) units with more end locations than a list keeps: their best score is already
            // exact, so a scan with threshold = best and a list of the right size finds every location (strip kernel)
            const size_t n = (size_t)n_;
            PinBuf sc; EDLIB_AMD_HIP(sc.alloc(2 * n * sizeof(int)));
            EDLIB_AMD_HIP(hipMemcpyAsync(sc.p, d_flatOut3_.p, 2 * n * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            const int* score = reinterpret_cast<const int*>(sc.p); const int* count = score + n;
            std::vector<PairDesc> d2;
            for (int u = 0; u < n_; ++u)
                if (count[u] > kFlatPosCap) {
                    PairDesc x = flatDesc(u);
                    x.kinit = score[u]; x.posCap = count[u]; x.posOff = flatOvfOff_.back(); x.ring = 0;
                    d2.push_back(x); flatOvfUnit_.push_back(u); flatOvfOff_.push_back(flatOvfOff_.back() + count[u]);
                }
            DevBuf<PairDesc> dd; DevBuf<int> s2;
            DevBuf<int>& pool2 = d_flatOvfPool_;                     // (the lists stay on the device: the collection reads them there)
            EDLIB_AMD_HIP(dd.alloc(d2.size())); EDLIB_AMD_HIP(pool2.ensure((size_t)flatOvfOff_.back())); EDLIB_AMD_HIP(s2.alloc(3 * d2.size()));
            EDLIB_AMD_HIP(hipMemcpyAsync(dd.p, d2.data(), d2.size() * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
            PairScanArgs a2 = a;
            a2.descs = dd.p; a2.numUnits = (int)d2.size(); a2.posPool = pool2.p;
            a2.outScore = s2.p; a2.outCount = s2.p + d2.size(); a2.outLast = s2.p + 2 * d2.size();
            scanTimerStart();
            EDLIB_AMD_HIP(launch_scan_pairs(scanMode, false, a2, stream_));
            scanTimerStop();
            // which list belongs to which unit, for the device-side collection
            {
                std::vector<int> at((size_t)n_, -1);
                for (size_t j = 0; j < flatOvfUnit_.size(); ++j) at[(size_t)flatOvfUnit_[j]] = (int)j;
                EDLIB_AMD_HIP(d_flatOvfAt_.ensure((size_t)n_)); EDLIB_AMD_HIP(d_flatOvfOff_.ensure(flatOvfOff_.size()));
                EDLIB_AMD_HIP(hipMemcpyAsync(d_flatOvfAt_.p, at.data(), (size_t)n_ * sizeof(int), hipMemcpyHostToDevice, stream_));
                EDLIB_AMD_HIP(hipMemcpyAsync(d_flatOvfOff_.p, flatOvfOff_.data(), flatOvfOff_.size() * sizeof(long long), hipMemcpyHostToDevice, stream_));
            }
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            stats.overflow_units += (int)d2.size();
            for (const PairDesc& x : d2) stats.word_steps += 2LL * ((x.qlen + 63) / 64) * x.tlen;
        }
    }
    if (flatStarts_ || flatPaths_) return runFlatStartsAndPaths(fellBack);
    return 0;
}

// Phases 2 and 3 of a flat SHW / HW batch (reference edlib.cpp:228-289), everything on the device: descriptors of the
// reverse prefix scans written by a kernel from the phase-1 results (HW), one ring scan over them, the starts; then one
// storing NW scan per unit over its first location's window + the traceback into the resident op slots.
int Batch::runFlatStartsAndPaths(bool& fellBack)
{
    const int mode = (int)cfg_.mode;
    const int* score = d_flatOut3_.p; const int* count = d_flatOut3_.p + n_;
    PairScanArgs a{};
    a.qpool = d_qpool_.p; a.tpool = d_tpool_.p; a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = nullptr;
    a.peqRowStride = peq_row_stride(std::max(flatRing_, flatMaxBlocks_));
    a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
    a.posPool = d_flatPos_.p; a.wordSteps = nullptr;
    if (flatStarts_) {
        EDLIB_AMD_HIP(hipMemsetAsync(d_flatCensus_.p, 0, 2 * sizeof(int), stream_));
        hipLaunchKernelGGL(flat_start_descs_kernel, dim3((n_ + 255) / 256), dim3(256), 0, stream_, d_flatDescs_.p, score, count, d_flatPos_.p,
                           n_, kFlatPosCap, flatRevPeqBase_, flatRing_, d_flatStartDescs_.p, d_flatSlotOf_.p, d_flatCensus_.p, (int)flatStartCap_);
        EDLIB_AMD_HIP(hipMemcpyAsync(h_flatCensus_.p, d_flatCensus_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
        // (the reversed queries' Peq rows do not depend on the count: built while it travels)
        EDLIB_AMD_HIP(launch_build_peq_pairs(d_flatRevDescs_.p, n_, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT, d_peq64_.p, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        const int nscan = *reinterpret_cast<const int*>(h_flatCensus_.p);
        if ((size_t)nscan > flatStartCap_) { fellBack = true; return 0; }
        if (nscan > 0) {
            PairScanArgs b = a;
            b.descs = d_flatStartDescs_.p; b.numUnits = nscan; b.store = nullptr;
            b.outScore = d_flatStartOut3_.p; b.outCount = d_flatStartOut3_.p + flatStartCap_; b.outLast = d_flatStartOut3_.p + 2 * flatStartCap_;
            scanTimerStart();
            EDLIB_AMD_HIP(launch_scan_pairs_ring(flatRing_, EDLIB_MODE_SHW, false, b, stream_));
            scanTimerStop();
        }
        const long long total = (long long)n_ * kFlatPosCap;
        hipLaunchKernelGGL(flat_starts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream_, d_flatSlotOf_.p, d_flatPos_.p,
                           d_flatStartOut3_.p + 2 * flatStartCap_, total, d_flatStartsOut_.p);
        EDLIB_AMD_HIP(hipGetLastError());
    }
    if (flatPaths_ && !flatNwStore_) {
        hipLaunchKernelGGL(flat_path_descs_kernel, dim3((n_ + 255) / 256), dim3(256), 0, stream_, d_flatDescs_.p, score, count, d_flatPos_.p,
                           mode == EDLIB_MODE_HW ? d_flatStartsOut_.p : nullptr, n_, kFlatPosCap, d_flatStoreBase_.p, flatRing_, d_flatPathDescs_.p);
        PairScanArgs b = a;
        b.descs = d_flatPathDescs_.p; b.numUnits = n_; b.store = d_store_.p;
        b.outScore = d_flatPathOut3_.p; b.outCount = d_flatPathOut3_.p + n_; b.outLast = d_flatPathOut3_.p + 2 * (size_t)n_;
        b.tsym = flatRing32_ ? d_tsym_.p : nullptr;
        scanTimerStart();
        if (flatRing32_) EDLIB_AMD_HIP(launch_scan_pairs_ring32(flatG32_, true, b, flatMaxWords_, stream_));
        else EDLIB_AMD_HIP(launch_scan_pairs_ring(flatRing_, EDLIB_MODE_NW, true, b, stream_));
        scanTimerStop();
        TracebackArgs tb{};
        tb.descs = d_flatPathDescs_.p; tb.numUnits = n_; tb.score = d_flatPathOut3_.p; tb.store = d_store_.p;
        tb.ops = d_flatOps_.p; tb.opsOff = d_flatOpsOff_.p; tb.opsLen = d_flatOpsLen_.p;
        if (flatRing32_) EDLIB_AMD_HIP(launch_traceback32(tb, flatG32_, stream_));
        else EDLIB_AMD_HIP(launch_traceback(tb, stream_));
    }
    return 0;
}

// ---------------------------------------------------------------- collection

// CIGAR strings of format f (0 extended, 1 standard) from dense op bytes on the device: lengths, their prefix sum, the
// strings into a buffer of `cap` characters (an upper bound: a run of one op is two characters), and the offsets on their
// way to pinned host memory -- everything enqueued on `st`, nothing waited for.
int Batch::enqueueCigars(int f, const uint8_t* aln, const long long* alnOff, size_t cap, hipStream_t st)
{
    const size_t n = (size_t)n_, nblocks = (n + 255) / 256;
    const size_t words = 3 * n + nblocks + 4;
    EDLIB_AMD_HIP(d_cigWork_.ensure(2 * words));
    long long* cigLen = d_cigWork_.p + (size_t)f * words; long long* cigRel = cigLen + n; long long* blockTot = cigRel + n;
    long long* totals = blockTot + nblocks; long long* cigOff = totals + 2;
    DevBuf<char>& chars = f ? d_cigChars2_ : d_cigChars_;
    EDLIB_AMD_HIP(chars.ensure(cap));
    CigarOut& c = cigar_[f];
    if (c.offs.n < (n + 1) * sizeof(long long)) EDLIB_AMD_HIP(c.offs.alloc((n + 1) * sizeof(long long)));
    EDLIB_AMD_HIP(launch_cigars(aln, alnOff, n_, f, cigLen, cigRel, blockTot, totals, nullptr, cigOff, 0, st));
    EDLIB_AMD_HIP(launch_cigars(aln, alnOff, n_, f, cigLen, cigRel, blockTot, totals, chars.p, cigOff, 1, st));
    EDLIB_AMD_HIP(hipMemcpyAsync(c.offs.p, cigOff, (n + 1) * sizeof(long long), hipMemcpyDeviceToHost, st));
    return 0;
}
// ... and, once the offsets are on the host, exactly as many characters as they say
int Batch::fetchCigars(int f, hipStream_t st)
{
    CigarOut& c = cigar_[f];
    const long long total = reinterpret_cast<const long long*>(c.offs.p)[n_];
    DevBuf<char>& chars = f ? d_cigChars2_ : d_cigChars_;
    if (total < (long long)n_ || (size_t)total > chars.n) { set_error("CIGAR: bad total"); return 1; }
    if (c.chars.n < (size_t)total) EDLIB_AMD_HIP(c.chars.alloc((size_t)total + (size_t)total / 8));
    EDLIB_AMD_HIP(hipMemcpyAsync(c.chars.p, chars.p, (size_t)total, hipMemcpyDeviceToHost, st));
    return 0;
}

// The caller-facing arrays of the last flat run, made on the device (flat_results.hip) and brought over as ONE block of
// pinned host memory: what edlibAmdBatchResultsView() hands out, what the per-unit records and the malloc'd arrays of the
// older entry points are copied from.  Valid until the next run().
int Batch::buildFlatView()
{
    if (viewReady_) return 0;
    const int mode = (int)cfg_.mode;
    const int scanMode = (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) ? mode : EDLIB_MODE_NW;
    const size_t n = (size_t)n_;
    const bool wantStarts = cfg_.task != EDLIB_TASK_DISTANCE, wantPath = flatPaths_;
    const size_t nblocks = (n + 255) / 256;
    const long long capLoc = (scanMode == EDLIB_MODE_NW ? (long long)n : (long long)n * (kFlatPosCap + 1)) + (flatOvfOff_.empty() ? 0 : flatOvfOff_.back());
    const long long capAln = wantPath ? flatOpsTotal_ : 0;
    // ---- layout of the block (the same offsets on the device and in pinned host memory)
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 63) & ~(size_t)63; return o; };
    // (what the host wants first -- the D2H of this head is most of what a DISTANCE batch brings over -- then the device's
    // scratch; status is all zeros for a flat batch and is filled in on the host)
    const size_t oTotals = take(16);
    const size_t oEd = take(n * 4), oNloc = take(n * 4), oAlpha = take(n * 4);
    const size_t oLocOff = take((n + 1) * 8), oAlnOff = take((n + 1) * 8);
    const size_t headBytes = at;
    const size_t oStatus = take(n * 4), oAlnLen = take(n * 4), oBlockLoc = take(nblocks * 8), oBlockAln = take(nblocks * 8);
    const size_t hostHead = headBytes + ((n * 4 + 63) & ~(size_t)63);     // host: the head + the status array
    // (device: room for every location / op byte the batch could have; host: the fixed part now, the rest once the totals are known)
    const size_t oEnds = take((size_t)capLoc * 4), oStarts = wantStarts ? take((size_t)capLoc * 4) : 0, oAln = take((size_t)capAln + 16);
    EDLIB_AMD_HIP(d_view_.ensure(at));
    if (h_view_.n < hostHead) EDLIB_AMD_HIP(h_view_.alloc(hostHead));
    uint8_t* const dv = d_view_.p; uint8_t* const hv = h_view_.p;
    FlatResultArgs a{};
    a.descs = d_flatDescs_.p; a.n = n_; a.mode = scanMode; a.k = cfg_.k; a.wantPath = wantPath ? 1 : 0; a.posCap = kFlatPosCap;
    a.score = d_flatOut3_.p; a.count = d_flatOut3_.p + n; a.pos = d_flatPos_.p;
    a.devStarts = flatStarts_ ? d_flatStartsOut_.p : nullptr;
    if (!flatOvfUnit_.empty()) { a.ovfAt = d_flatOvfAt_.p; a.ovfOff = d_flatOvfOff_.p; a.ovfPos = d_flatOvfPool_.p; }
    // alphabetLength comes from the side stream's kernel (or, for a handful of short sequences, from the host below)
    const bool alphaOnDevice = alphaPending_ && !alphaOnHost_ && alphaUnits_.size() == n;
    if (alphaOnDevice) { EDLIB_AMD_HIP(hipStreamWaitEvent(stream_, evB_.e, 0)); a.alphabet = d_alphaOut_.p; }
    if (wantPath) { a.opsLen = d_flatOpsLen_.p; a.opsOff = d_flatOpsOff_.p; a.ops = d_flatOps_.p; }
    a.status = reinterpret_cast<int*>(dv + oStatus); a.editDistance = reinterpret_cast<int*>(dv + oEd);
    a.numLocations = reinterpret_cast<int*>(dv + oNloc); a.alphabetLength = reinterpret_cast<int*>(dv + oAlpha);
    a.alnLen = reinterpret_cast<int*>(dv + oAlnLen);
    a.locOff = reinterpret_cast<long long*>(dv + oLocOff); a.alnOff = reinterpret_cast<long long*>(dv + oAlnOff);
    a.blockLoc = reinterpret_cast<long long*>(dv + oBlockLoc); a.blockAln = reinterpret_cast<long long*>(dv + oBlockAln);
    a.ends = reinterpret_cast<int*>(dv + oEnds); a.starts = wantStarts ? reinterpret_cast<int*>(dv + oStarts) : nullptr;
    a.aln = dv + oAln;
    EDLIB_AMD_HIP(launch_flat_results(a, reinterpret_cast<long long*>(dv + oTotals), stream_));
    // A PATH batch whose caller asked for CIGARs after an earlier run gets them made NOW, on the side stream, while the op
    // bytes travel: the run-length encoding of both formats (0.16 ms of kernels for config 5) overlaps the 10 MB copy
    // instead of following it, and shares its synchronisations.  (The first request of a session is served on demand.)
    bool prefetchCigars = cigarSticky_ && wantPath && capAln > 0;
    if (prefetchCigars) {
        if (!side_) EDLIB_AMD_HIP(pool_stream(&side_));
        EDLIB_AMD_HIP(evView_.create());
        EDLIB_AMD_HIP(hipEventRecord(evView_.e, stream_));
        EDLIB_AMD_HIP(hipStreamWaitEvent(side_, evView_.e, 0));
        // Best effort: the prefetch sizes its character buffers for the worst case (two characters per op slot, both formats)
        // before the totals are known.  When that does not fit the device the view is still good -- the strings are then made
        // on demand (cigarView), sized by what there is.
        for (int f = 0; f < 2 && prefetchCigars; ++f)
            if (enqueueCigars(f, dv + oAln, reinterpret_cast<const long long*>(dv + oAlnOff), (size_t)(2 * capAln) + n + 64, side_)) {
                prefetchCigars = false;
                (void)hipGetLastError();
                (void)hipStreamSynchronize(side_);               // (whatever of it was queued is over before its buffers are asked for again)
                d_cigChars_.release(); d_cigChars2_.release();
            }
    }
    // ---- the fixed part (per-unit fields, offsets, totals), then exactly as many locations / op bytes as there are
    EDLIB_AMD_HIP(hipMemcpyAsync(hv, dv, headBytes, hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    const long long* totals = reinterpret_cast<const long long*>(hv + oTotals);
    const long long nloc = totals[0], naln = totals[1];
    if (nloc < 0 || nloc > capLoc || naln < 0 || naln > capAln) { set_error("flat results: totals out of range"); return 1; }
    // the variable part in a pinned block of its own, sized by what there is (a block for everything a batch COULD have --
    // 146 MB of op slots for 262,144 x 150 bp HW paths that come to 39 MB -- cost its pinning on the first collection)
    const size_t vEnds = 0, vStarts = ((size_t)nloc * 4 + 63) & ~(size_t)63, vAln = vStarts + (wantStarts ? vStarts : 0);
    const size_t varBytes = vAln + (size_t)naln + 64;
    if (h_viewVar_.n < varBytes) EDLIB_AMD_HIP(h_viewVar_.alloc(varBytes + varBytes / 8));
    uint8_t* const hvar = h_viewVar_.p;
    if (nloc) EDLIB_AMD_HIP(hipMemcpyAsync(hvar + vEnds, dv + oEnds, (size_t)nloc * 4, hipMemcpyDeviceToHost, stream_));
    if (nloc && wantStarts) EDLIB_AMD_HIP(hipMemcpyAsync(hvar + vStarts, dv + oStarts, (size_t)nloc * 4, hipMemcpyDeviceToHost, stream_));
    if (naln) EDLIB_AMD_HIP(hipMemcpyAsync(hvar + vAln, dv + oAln, (size_t)naln, hipMemcpyDeviceToHost, stream_));
    if (prefetchCigars) {                              // the strings: as many characters as the offsets say (the side stream has them)
        EDLIB_AMD_HIP(hipStreamSynchronize(side_));
        for (int f = 0; f < 2; ++f) if (fetchCigars(f, side_)) return 1;
        EDLIB_AMD_HIP(hipStreamSynchronize(side_));
        for (int f = 0; f < 2; ++f) { cigar_[f].p = reinterpret_cast<const char*>(cigar_[f].chars.p); cigar_[f].off = reinterpret_cast<const long long*>(cigar_[f].offs.p); cigar_[f].ready = true; }
    }
    if (alphaPending_ && !alphaOnDevice) { EDLIB_AMD_HIP(hipStreamSynchronize(side_)); }
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    alphaPending_ = false;
    view_ = EdlibAmdResultsView{};
    view_.numUnits = n_;
    memset(hv + headBytes, 0, n * 4);
    view_.status = reinterpret_cast<const int*>(hv + headBytes); view_.editDistance = reinterpret_cast<const int*>(hv + oEd);
    view_.numLocations = reinterpret_cast<const int*>(hv + oNloc); view_.alphabetLength = reinterpret_cast<const int*>(hv + oAlpha);
    view_.locOffsets = reinterpret_cast<const long long*>(hv + oLocOff); view_.alnOffsets = reinterpret_cast<const long long*>(hv + oAlnOff);
    view_.endLocations = reinterpret_cast<const int*>(hvar + vEnds);
    view_.startLocations = wantStarts ? reinterpret_cast<const int*>(hvar + vStarts) : nullptr;
    view_.alignment = wantPath ? hvar + vAln : nullptr;
    viewAlnDev_ = dv + oAln; viewAlnOffDev_ = reinterpret_cast<const long long*>(dv + oAlnOff);
    // ---- what the device does not know
    int* alpha = reinterpret_cast<int*>(hv + oAlpha);
    if (!alphaOnDevice) {
        if (alphaOnHost_) {
            // a handful of short sequences: counted on the host from the staging block of init() (alphabetLengthsEnd's rule)
            std::vector<UnitResult> tmp(n);
            if (alphabetLengthsEnd(tmp)) return 1;
            for (size_t u = 0; u < n; ++u) alpha[u] = tmp[u].alphabetLength;
        } else if (alphaPin_.p && alphaUnits_.size() == n) {
            memcpy(alpha, alphaPin_.p, n * sizeof(int));
        }
    }
    // start locations beyond the 16 a unit's flat list keeps (a unit of the exact second pass, HW): their reverse scans run now
    if (flatStarts_ && !flatOvfUnit_.empty()) {
        int* starts = reinterpret_cast<int*>(hvar + vStarts);
        std::vector<UnitSpec> late; std::vector<long long> where;
        for (int u : flatOvfUnit_) {
            const int ed = view_.editDistance[u], m = qlen(u);
            if (ed < 0) continue;
            const long long lo = view_.locOffsets[u]; const int cnt = view_.numLocations[u];
            const int lead = (cnt > 0 && view_.endLocations[lo] == -1) ? 1 : 0;
            for (int j = lead + kFlatPosCap; j < cnt; ++j) {
                const int e = view_.endLocations[lo + j];
                const long long win = std::min<long long>((long long)e + 1, (long long)m + ed);
                late.push_back(UnitSpec{qoff_[u] + m - 1, m, -1, tbase(u) + e, (int)win, -1, ed});
                where.push_back(lo + j);
            }
        }
        if (!late.empty()) {
            SolveOut so;
            if (solveSemiGlobal(EDLIB_MODE_SHW, false, late, so)) return 1;
            for (size_t i = 0; i < late.size(); ++i) starts[where[i]] = view_.endLocations[where[i]] - so.last[i];      // (edlib.cpp:260)
        }
    }
    viewReady_ = true;
    return 0;
}

// the per-unit records of the older entry points, from the view
int Batch::collectPairsFlat(std::vector<UnitResult>& res)
{
    if (buildFlatView()) return 1;
    const EdlibAmdResultsView& v = view_;
    const bool wantStarts = cfg_.task != EDLIB_TASK_DISTANCE;
    for (size_t u = 0; u < (size_t)n_; ++u) {
        UnitResult& r = res[u];
        r.status = v.status[u]; r.editDistance = v.editDistance[u]; r.alphabetLength = v.alphabetLength[u];
        r.hasStarts = r.hasAlignment = false;
        const int cnt = v.numLocations[u];
        const long long lo = v.locOffsets[u];
        r.hasEnds = cnt > 0;
        r.ends.clear(); r.starts.clear();
        if (cnt > 0) r.ends.append(v.endLocations + lo, (size_t)cnt);
        if (!wantStarts || r.editDistance < 0 || !r.hasEnds) continue;
        r.hasStarts = true;
        r.starts.append(v.startLocations + lo, (size_t)cnt);
        if (cfg_.task != EDLIB_TASK_PATH || !v.alignment) continue;
        r.hasAlignment = true;                                       // (:276-289: the path of the first location)
        r.opsView = v.alignment + v.alnOffsets[u]; r.opsViewLen = (int)(v.alnOffsets[u + 1] - v.alnOffsets[u]);
    }
    pairsCollected_ = true;
    return 0;
}

// ------------------------------------------------------ the view of a DISTANCE batch of reads

// out[slots[i]] = i: the overflow index of the slots the exact second pass served (flat_results.hip: ovfAt)
__global__ void __launch_bounds__(256)
scatter_index_kernel(const int* __restrict__ slots, int count, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[slots[i]] = i;
}

// slot -> unit order: what flat_results.hip reads per unit, gathered from one group's per-slot arrays
__global__ void __launch_bounds__(256)
gather_group_kernel(const int* __restrict__ perm, int nslots, const int* __restrict__ best, const int* __restrict__ total,
                    const int* __restrict__ qlen, const int* __restrict__ extra, const int* __restrict__ pos, int posCap,
                    int* __restrict__ uScore, int* __restrict__ uCount, int* __restrict__ uQlen, int* __restrict__ uAlpha,
                    int* __restrict__ uPos)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int u = perm[s];
    if (u < 0) return;
    uScore[u] = best[s]; uCount[u] = total[s]; uQlen[u] = qlen[s]; uAlpha[u] = extra[s];
    // (a small group's arrays are views into pinned host memory at an odd offset: no vector loads)
    for (int i = 0; i < posCap; ++i) uPos[(size_t)u * posCap + i] = pos[(size_t)s * posCap + i];
}

// A TASK_DISTANCE run over reads-path units only leaves its results in HBM (per slot of each word-count group: best score,
// number of end locations, the first 16 of them, the lists of the exact second pass).  That is what flat_results.hip lays out
// for a flat pair batch: the caller-facing arrays are made on the device and come over as one block, no per-read record is
// built (1M reads: 63 ms of records and copies before, 15 now).  One group of all units (reads of one word count: the
// north-star shape) has slot == unit and is read in place; several groups are first gathered into unit order.  results() --
// per-unit malloc'd arrays -- still builds its records.
bool Batch::readsViewOnDevice() const
{
    if (readsCollected_ || flatPairs_ || groups_.empty() || cfg_.task != EDLIB_TASK_DISTANCE) return false;
    if (!pairUnits_.empty() || !longUnits_.empty() || !emptyUnits_.empty() || readUnits_.size() != (size_t)n_) return false;
    const int mode = (int)cfg_.mode;
    if (mode != EDLIB_MODE_NW && mode != EDLIB_MODE_SHW && mode != EDLIB_MODE_HW) return false;
    if (groups_.size() == 1 && groups_[0]->zeroCopy) return false;          // (a handful of reads: their results are on the host already)
    return true;
}

int Batch::buildReadsView()
{
    if (viewReady_) return 0;
    const int mode = (int)cfg_.mode;
    const size_t n = (size_t)n_;
    const size_t nblocks = (n + 255) / 256;
    const bool inPlace = groups_.size() == 1;          // slot == unit (makeGroup lists a group's units in unit order)
    size_t novf = 0; long long ovfTotal = 0;
    for (auto& gp : groups_) { novf += gp->ovfSlots.size(); ovfTotal += gp->ovfOff.empty() ? 0 : gp->ovfOff.back(); }
    const int posCap = mode == EDLIB_MODE_NW ? 0 : kFlatPosCap;
    const long long capLoc = (mode == EDLIB_MODE_NW ? (long long)n : (long long)n * (kFlatPosCap + 1)) + ovfTotal;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 63) & ~(size_t)63; return o; };
    const size_t oTotals = take(16);
    const size_t oEd = take(n * 4), oNloc = take(n * 4), oAlpha = take(n * 4);
    const size_t oLocOff = take((n + 1) * 8), oAlnOff = take((n + 1) * 8);
    const size_t headBytes = at;
    const size_t oStatus = take(n * 4), oAlnLen = take(n * 4), oBlockLoc = take(nblocks * 8), oBlockAln = take(nblocks * 8);
    const size_t oOvfAt = take(novf ? n * 4 : 4), oOvfOff = take((novf + 1) * 8), oOvfUnits = take((novf + 1) * 4);
    const size_t oOvfPool = take(inPlace ? 4 : (size_t)ovfTotal * 4 + 4);
    const size_t oUScore = take(inPlace ? 4 : n * 4), oUCount = take(inPlace ? 4 : n * 4), oUQlen = take(inPlace ? 4 : n * 4), oUAlpha = take(inPlace ? 4 : n * 4);
    const size_t oUPos = take(inPlace ? 4 : n * (size_t)posCap * 4 + 4);
    const size_t hostHead = headBytes + ((n * 4 + 63) & ~(size_t)63);
    const size_t oEnds = take((size_t)capLoc * 4), oAln = take(64);
    EDLIB_AMD_HIP(d_view_.ensure(at));
    if (h_view_.n < hostHead) EDLIB_AMD_HIP(h_view_.alloc(hostHead));
    uint8_t* const dv = d_view_.p; uint8_t* const hv = h_view_.p;
    FlatResultArgs a{};
    a.descs = nullptr; a.sharedT = tlen(0); a.alphaBase = tab_.sigmaT;
    a.n = n_; a.mode = mode; a.k = cfg_.k; a.wantPath = 0; a.posCap = kFlatPosCap;
    // ---- the per-unit inputs: in place, or gathered from the groups
    std::vector<int> ovfUnits; std::vector<long long> ovfOffAll;
    if (inPlace) {
        ReadGroup& g = *groups_[0];
        a.qlens = g.d_qlen.p; a.score = g.d_best.p; a.count = g.d_total.p; a.pos = g.d_pos.p; a.alphabet = g.d_alphaExtra.p;
        a.ovfPos = g.d_ovfPool.p;
        ovfUnits = g.ovfSlots; ovfOffAll = g.ovfOff;
    } else {
        int* uScore = reinterpret_cast<int*>(dv + oUScore); int* uCount = reinterpret_cast<int*>(dv + oUCount);
        int* uQlen = reinterpret_cast<int*>(dv + oUQlen); int* uAlpha = reinterpret_cast<int*>(dv + oUAlpha);
        int* uPos = reinterpret_cast<int*>(dv + oUPos); int* pool = reinterpret_cast<int*>(dv + oOvfPool);
        ovfOffAll.push_back(0);
        long long poolAt = 0;
        for (auto& gp : groups_) {
            ReadGroup& g = *gp;
            hipLaunchKernelGGL(gather_group_kernel, dim3((unsigned)((g.nslots + 255) / 256)), dim3(256), 0, stream_,
                               g.d_perm.p, g.nslots, g.d_best.p, g.d_total.p, g.d_qlen.p, g.d_alphaExtra.p, g.d_pos.p, posCap,
                               uScore, uCount, uQlen, uAlpha, uPos);
            EDLIB_AMD_HIP(hipGetLastError());
            for (size_t i = 0; i < g.ovfSlots.size(); ++i) {
                ovfUnits.push_back(g.perm[g.ovfSlots[i]]);
                ovfOffAll.push_back(poolAt + g.ovfOff[i + 1]);
            }
            const long long mine = g.ovfOff.empty() ? 0 : g.ovfOff.back();
            if (mine) EDLIB_AMD_HIP(hipMemcpyAsync(pool + poolAt, g.d_ovfPool.p, (size_t)mine * sizeof(int), hipMemcpyDeviceToDevice, stream_));
            poolAt += mine;
        }
        a.qlens = uQlen; a.score = uScore; a.count = uCount; a.pos = uPos; a.alphabet = uAlpha; a.ovfPos = pool;
    }
    if (novf) {
        // (pageable sources: the lists are a few thousand entries; the stream is synchronised below before they go out of scope)
        EDLIB_AMD_HIP(hipMemsetAsync(dv + oOvfAt, 0xff, n * 4, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(dv + oOvfUnits, ovfUnits.data(), novf * sizeof(int), hipMemcpyHostToDevice, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(dv + oOvfOff, ovfOffAll.data(), (novf + 1) * sizeof(long long), hipMemcpyHostToDevice, stream_));
        hipLaunchKernelGGL(scatter_index_kernel, dim3((unsigned)((novf + 255) / 256)), dim3(256), 0, stream_,
                           reinterpret_cast<const int*>(dv + oOvfUnits), (int)novf, reinterpret_cast<int*>(dv + oOvfAt));
        EDLIB_AMD_HIP(hipGetLastError());
        a.ovfAt = reinterpret_cast<const int*>(dv + oOvfAt); a.ovfOff = reinterpret_cast<const long long*>(dv + oOvfOff);
    } else a.ovfPos = nullptr;
    a.status = reinterpret_cast<int*>(dv + oStatus); a.editDistance = reinterpret_cast<int*>(dv + oEd);
    a.numLocations = reinterpret_cast<int*>(dv + oNloc); a.alphabetLength = reinterpret_cast<int*>(dv + oAlpha);
    a.alnLen = reinterpret_cast<int*>(dv + oAlnLen);
    a.locOff = reinterpret_cast<long long*>(dv + oLocOff); a.alnOff = reinterpret_cast<long long*>(dv + oAlnOff);
    a.blockLoc = reinterpret_cast<long long*>(dv + oBlockLoc); a.blockAln = reinterpret_cast<long long*>(dv + oBlockAln);
    a.ends = reinterpret_cast<int*>(dv + oEnds); a.starts = nullptr; a.aln = dv + oAln;
    EDLIB_AMD_HIP(launch_flat_results(a, reinterpret_cast<long long*>(dv + oTotals), stream_));
    EDLIB_AMD_HIP(hipMemcpyAsync(hv, dv, headBytes, hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    const long long nloc = reinterpret_cast<const long long*>(hv + oTotals)[0];
    if (nloc < 0 || nloc > capLoc) { set_error("reads view: totals out of range"); return 1; }
    const size_t varBytes = (size_t)nloc * 4 + 64;
    if (h_viewVar_.n < varBytes) EDLIB_AMD_HIP(h_viewVar_.alloc(varBytes + varBytes / 8));
    if (nloc) EDLIB_AMD_HIP(hipMemcpyAsync(h_viewVar_.p, dv + oEnds, (size_t)nloc * 4, hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    memset(hv + headBytes, 0, n * 4);                    // status: every unit of a reads group is EDLIB_STATUS_OK
    view_ = EdlibAmdResultsView{};
    view_.numUnits = n_;
    view_.status = reinterpret_cast<const int*>(hv + headBytes); view_.editDistance = reinterpret_cast<const int*>(hv + oEd);
    view_.numLocations = reinterpret_cast<const int*>(hv + oNloc); view_.alphabetLength = reinterpret_cast<const int*>(hv + oAlpha);
    view_.locOffsets = reinterpret_cast<const long long*>(hv + oLocOff); view_.alnOffsets = reinterpret_cast<const long long*>(hv + oAlnOff);
    view_.endLocations = reinterpret_cast<const int*>(h_viewVar_.p);
    view_.startLocations = nullptr; view_.alignment = nullptr;
    viewAlnDev_ = nullptr; viewAlnOffDev_ = nullptr;
    viewReady_ = true;
    if (getenv("EDLIB_AMD_DEBUG")) fprintf(stderr, "[edlib_amd] reads view made on the device: %d units in %zu group(s), %lld locations, %zu lists of the exact pass\n", n_, groups_.size(), nloc, novf);
    return 0;
}

int Batch::ensureCollected()
{
    if (readsCollected_ && pairsCollected_) return 0;
    DeviceGuard guard(device_);
    EDLIB_AMD_HIP(guard.status);
    if (results_.size() != (size_t)n_) results_.assign((size_t)n_, UnitResult{});
    if (!readsCollected_ && collectReads(results_)) return 1;
    if (!pairsCollected_ && collectPairsFlat(results_)) return 1;
    return 0;
}


}  // namespace edlib_amd
