// engine_pairs.hip -- host side of the block-per-lane path: independent (query, target) units of any length,
// alphabet and mode.  solve() / solveChunk() run one launch family over a set of units (strips, lane rings, the wide
// kernel); above them the reference's k-doubling (edlib.cpp:197-217) as threshold levels over the ring sizes for NW
// (solveGlobalDistances), the packing of semi-global units onto rings, SHW inside the band of a threshold, HW targets
// cut into segments.
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace edlib_amd {

// ------------------------------------------------------ block-per-lane path



// Row length of the LDS-resident Peq of the ring kernels: the next power of two up to 32 blocks, a
// multiple of 32 above (bank-conflict-free lookups, scan_pairs_ring_kernel)
int peq_row_stride(long long nb) {
    if (nb > 32) return (int)std::min<long long>((nb + 31) / 32 * 32, 1 << 20);
    int s = 1; while (s < nb) s <<= 1;
    return s;
}

// the counter the ring kernels of this run add their live word-steps to (zeroed by run(), read back at its end)
unsigned long long* Batch::ringStepsCounter()
{
    if (!d_ringSteps_.p) {
        if (d_ringSteps_.alloc(1) != hipSuccess || h_ringSteps_.alloc(sizeof(unsigned long long)) != hipSuccess) return nullptr;
        if (hipMemsetAsync(d_ringSteps_.p, 0, sizeof(unsigned long long), stream_) != hipSuccess) return nullptr;
    }
    ringStepsUsed_ = true;
    return d_ringSteps_.p;
}

// Can the storing scans of units [a, b) run on G-lane rings of 32-row words (ring32_kernels.hip)?  Forward units whose band
// fits the ring (or whose words all sit on it), a chunk that cannot fill the chip (the kernels are built for lone waves), a
// target pool small enough to be mapped to symbol ids per call, a store the kernel's 32-bit offsets reach, a Peq table per
// unit that fits a wave's LDS.
bool Batch::ring32_fits(const std::vector<UnitSpec>& units, size_t a, size_t b, int G) const
{
    if (b - a > 20000 || d_tpool_.n > ((size_t)64 << 20)) return false;
    long long store = 0; int maxWords = 1;
    for (size_t i = a; i < b; ++i) {
        const UnitSpec& u = units[i];
        const int nw = (u.qlen + 31) / 32;
        if (u.qstep != 1 || u.tstep != 1 || !(nw <= G || u.kinit <= ring32_max_k(G))) return false;
        store += 8LL * ring32_store_entries(G, u.qlen, u.tlen);
        maxWords = std::max(maxWords, nw);
    }
    return store < 0xF0000000LL && ring32_lds_bytes(G, tab_.sigmaT, maxWords) <= 48 * 1024;
}

int Batch::solve(int mode, bool wantPositions, bool wantPath, const std::vector<UnitSpec>& units, SolveOut& out,
                 int ring, int ringH)
{
    const size_t n = units.size();
    out.score.assign(n, -1); out.count.assign(n, 0); out.last.assign(n, -1);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    if (n == 0) return 0;
    if (ring == kWide && wantPath) { set_error("the wide kernel keeps no column store"); return 1; }
    stats.path |= 2;
    // chunk so that the Peq pool and (for PATH) the column store stay within a budget
    const long long peqBudget = 4LL << 30, storeBudget = 12LL << 30;
    size_t a = 0;
    while (a < n) {
        long long peqBytes = 0, storeBytes = 0;
        size_t b = a;
        while (b < n) {
            const long long nb = (units[b].qlen + 63) / 64;
            const long long pb = nb * tab_.sigmaT * 8;
            const long long sb = !wantPath ? 0 : (ring == kRing32 ? 8LL * ring32_store_entries(16, units[b].qlen, units[b].tlen)
                                                 : (long long)sizeof(StoreEntry) * (ring > 0 ? ring_store_entries(ring, units[b].qlen, units[b].tlen)
                                                            : pair_store_entries(units[b].qlen, units[b].tlen)));
            if (b > a && (peqBytes + pb > peqBudget || storeBytes + sb > storeBudget)) break;
            peqBytes += pb; storeBytes += sb; ++b;
        }
        if (solveChunk(mode, wantPositions, wantPath, units, a, b, out, ring, ringH)) return 1;
        a = b;
    }
    return 0;
}

int Batch::solveChunk(int mode, bool wantPositions, bool wantPath, const std::vector<UnitSpec>& units,
                      size_t ua, size_t ub, SolveOut& out, int ring, int ringH)
{
    const size_t n = ub - ua;
    Lap lap;
    PinBuf descsPin;                                   // built in pinned staging: the H2D runs at link rate
    EDLIB_AMD_HIP(descsPin.alloc(n * sizeof(PairDesc)));
    PairDesc* descs = reinterpret_cast<PairDesc*>(descsPin.p);
    std::vector<long long> opsOff(wantPath ? n + 1 : 1, 0);           // [n] = total op bytes (0 without PATH)
    long long peqWords = 0, auxInts = 0, storeEntries = 0, nbMax = 0;
    // A storing scan on 4-lane rings over a chunk that cannot fill the chip (edlibAlign() with TASK_PATH on a 1 kb pair is
    // one unit) is bound by the latency of its waves' instruction streams: it takes the rings of 32-row words and their
    // walk instead (ring32_kernels.hip, DESIGN.md 4d: 0.18 against 0.48 us per step, a walk of ~T / 32 trips).  Every unit
    // of a 4-lane launch fits a ring of words: at most 4 blocks = 8 words on 8 lanes, or -- a unit of more words at the 4-lane
    // ring's band limit of 196, above ring32_max_k(8) = 192 -- the band on 16 lanes (ring32_max_k(16) = 448).
    // (ring == kRing32: the caller asks for 16-lane rings of words and has checked ring32_fits())
    int g32 = ring == kRing32 ? 16 : 8;
    bool use32 = wantPath && (ring == 4 || ring == kRing32) && mode == EDLIB_MODE_NW && ring32_fits(units, ua, ub, g32);
    if (!use32 && wantPath && ring == 4 && mode == EDLIB_MODE_NW && ring32_fits(units, ua, ub, 16)) { g32 = 16; use32 = true; }
    int maxWords32 = 1;
    for (size_t i = 0; use32 && i < n; ++i) maxWords32 = std::max(maxWords32, (units[ua + i].qlen + 31) / 32);
    if (ring == kRing32 && !use32) { set_error("ring32 level on units that do not fit it"); return 1; }
    if (ring == kRing32) ring = 16;                    // (descriptor bookkeeping below: any ring size > 0)
    for (size_t i = 0; i < n; ++i) {
        const UnitSpec& s = units[ua + i];
        PairDesc& d = descs[i];
        const long long nb = (s.qlen + 63) / 64;
        nbMax = std::max(nbMax, nb);
        d.qoff = s.qoff; d.toff = s.toff; d.qlen = s.qlen; d.tlen = s.tlen; d.qstep = s.qstep; d.tstep = s.tstep;
        d.kinit = s.kinit; d.skip = s.skip;
        d.peqOff = peqWords; peqWords += nb * tab_.sigmaT;
        d.auxOff = auxInts; if (nb > 64 && !ring) auxInts += s.tlen;
        d.storeOff = storeEntries;
        if (wantPath) storeEntries += use32 ? ring32_store_entries(g32, s.qlen, s.tlen)               // (8-byte entries)
                                           : (ring > 0 ? ring_store_entries(ring, s.qlen, s.tlen) : pair_store_entries(s.qlen, s.tlen));
        d.posCap = wantPositions ? kPosCap : 0;
        d.posOff = (long long)i * kPosCap;
        d.colOff = -1; d.bandT = (s.band && (mode == EDLIB_MODE_SHW || mode == EDLIB_MODE_HW)) ? -1 : 0; d.ring = ring > 0 ? ring : 0;
        // op slot of the unit, filled from the back.  An alignment has (m + T + inserts + deletes) / 2 ops, and a ring scan
        // is only walked when its distance is within kinit: (m + T + kinit) / 2 bounds the length (config 5: 1064 bytes
        // instead of 2000 per pair to bring back over PCIe)
        if (wantPath) {
            const long long full = (long long)s.qlen + s.tlen;
            opsOff[i + 1] = opsOff[i] + (ring > 0 && s.kinit >= 0 && s.kinit < full ? (full + s.kinit) / 2 + 8 : full);
        }
        // executed work: whole matrix, or one 64-block wave per column inside the band
        // executed work: the strips update every block of every column; the rings count the updates inside the band themselves
        if (!ring) stats.word_steps += 2 * nb * (long long)s.tlen;
        else if (ring == kWide) stats.word_steps += wide_word_steps(mode, s.qlen, s.tlen, d.bandT, s.kinit);
    }
    WidePlan wplan;
    if (ring == kWide && planWide(mode, descs, n, wplan)) return 1;
    // A handful of units (edlibAlign() is one): the kernels write scores, positions and op strings straight into
    // device-visible pinned host memory -- no download commands behind the launches, one stream synchronisation.
    // (Descriptors still go up with a copy: the packed rings re-read them, and every read of host memory is a PCIe
    // round trip.)  Larger chunks stage through HBM: a PCIe transaction per store does not scale.
    const long long opsTotal = wantPath ? opsOff[n] : 0;
    const bool zeroCopy = n <= 16 && opsTotal <= (64 << 10) && pool_enabled();
    PinBuf outPin;
    int* hOut3 = nullptr; int* hPos = nullptr; int* hOpsLen = nullptr; long long* hOpsOff = nullptr;
    std::shared_ptr<PinBuf> ops;
    EDLIB_AMD_HIP(d_peq64_.ensure((size_t)peqWords));
    EDLIB_AMD_HIP(d_aux_.ensure((size_t)auxInts));
    if (zeroCopy) {
        const size_t bytes = (3 * n + n * kPosCap + n) * sizeof(int) + (n + 1) * sizeof(long long);
        EDLIB_AMD_HIP(outPin.alloc(bytes));
        hOpsOff = reinterpret_cast<long long*>(outPin.p);
        hOut3 = reinterpret_cast<int*>(hOpsOff + n + 1); hPos = hOut3 + 3 * n; hOpsLen = hPos + n * kPosCap;
        if (wantPath) memcpy(hOpsOff, opsOff.data(), (n + 1) * sizeof(long long));
        else memset(hOpsOff, 0, (n + 1) * sizeof(long long));
        EDLIB_AMD_HIP(d_descs_.ensure(n));
        d_out3_.alias(hOut3, 3 * n); d_posPool_.alias(hPos, n * kPosCap);
        d_opsLen_.alias(hOpsLen, n); d_opsOff_.alias(hOpsOff, n + 1);
        if (wantPath && opsOff[n] > 0) {
            ops = std::make_shared<PinBuf>();
            EDLIB_AMD_HIP(ops->alloc((size_t)opsOff[n]));
            d_ops_.alias(ops->p, (size_t)opsOff[n]);
        }
    } else {
        if (!d_out3_.owned) d_out3_.release();
        if (!d_posPool_.owned) d_posPool_.release();
        if (!d_opsLen_.owned) d_opsLen_.release();
        if (!d_opsOff_.owned) d_opsOff_.release();
        if (!d_ops_.owned) d_ops_.release();
        EDLIB_AMD_HIP(d_descs_.ensure(n));
        // score / count / last of the chunk side by side: one copy brings all three back
        EDLIB_AMD_HIP(d_out3_.ensure(3 * n));
        EDLIB_AMD_HIP(d_posPool_.ensure(n * kPosCap));
    }
    d_outScore_.alias(d_out3_.p, n); d_outCount_.alias(d_out3_.p + n, n); d_outLast_.alias(d_out3_.p + 2 * n, n);
    if (wantPath) {
        EDLIB_AMD_HIP(d_store_.ensure(use32 ? (size_t)(storeEntries + 1) / 2 : (size_t)storeEntries));
        if (!zeroCopy) {
            EDLIB_AMD_HIP(d_ops_.ensure((size_t)opsOff[n])); EDLIB_AMD_HIP(d_opsOff_.ensure(n + 1));
            EDLIB_AMD_HIP(d_opsLen_.ensure(n));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_opsOff_.p, opsOff.data(), (n + 1) * sizeof(long long), hipMemcpyHostToDevice, stream_));
        }
    }
    EDLIB_AMD_HIP(hipMemcpyAsync(d_descs_.p, descs, n * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_descs_.p, (int)n, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT,
                                         d_peq64_.p, stream_));
    lap("chunk: descs+alloc");
    PairScanArgs a{};
    a.descs = d_descs_.p; a.numUnits = (int)n; a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = d_aux_.p;
    a.peqRowStride = peq_row_stride(nbMax);
    a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
    if (getenv("EDLIB_AMD_PEQFULL") && getenv("EDLIB_AMD_PEQFULL")[0] == '0') a.peqFullStride = 0;
    a.store = d_store_.p;
    a.outScore = d_outScore_.p; a.outCount = d_outCount_.p; a.outLast = d_outLast_.p; a.posPool = d_posPool_.p;
    a.colP = nullptr; a.colM = nullptr; a.colS = nullptr;
    a.wordSteps = ring > 0 ? ringStepsCounter() : nullptr;
    if (use32) {                                       // the targets as symbol ids (the refills of the ring32 scan read them)
        EDLIB_AMD_HIP(d_tsym_.ensure(d_tpool_.n));
        EDLIB_AMD_HIP(launch_target_symbols(d_tpool_.p, d_tlut_.p, (long long)d_tpool_.n, d_tsym_.p, stream_));
        a.tsym = d_tsym_.p; a.wordSteps = nullptr;
        stats.word_steps += ring32_word_steps(g32, descs, (int)n);
    }
    scanTimerStart();
    if (ring == kWide) { if (launchWide(mode, a, descs, n, wplan)) return 1; }
    else if (use32) EDLIB_AMD_HIP(launch_scan_pairs_ring32(g32, true, a, maxWords32, stream_));
    else if (ring) EDLIB_AMD_HIP(launch_scan_pairs_ring(ring, mode, wantPath, a, stream_, ringH));
    else EDLIB_AMD_HIP(launch_scan_pairs(mode, wantPath, a, stream_));
    scanTimerStop();
    if (wantPath) {
        TracebackArgs tb{};
        tb.descs = d_descs_.p; tb.numUnits = (int)n; tb.score = d_outScore_.p;
        tb.store = d_store_.p;
        tb.ops = d_ops_.p; tb.opsOff = d_opsOff_.p; tb.opsLen = d_opsLen_.p;
        if (use32) EDLIB_AMD_HIP(launch_traceback32(tb, g32, stream_));
        else EDLIB_AMD_HIP(launch_traceback(tb, stream_));
    }
    if (lap.on) { EDLIB_AMD_HIP(hipStreamSynchronize(stream_)); lap("chunk: kernels"); }
    // downloads land in pinned staging (a pageable std::vector took 5 ms for the 17 MB of end positions of 262,144
    // short HW pairs); a zero-copy chunk is read where the kernels wrote it
    PinBuf stage;
    const int* score = nullptr; const int* pool = nullptr; const int* opsLen = nullptr;
    if (zeroCopy) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        score = hOut3; pool = hPos; opsLen = hOpsLen;
        if (wantPath && ops) out.opsBufs.push_back(ops);
    } else {
        const size_t nPos = wantPositions ? n * kPosCap : 0, nLen = wantPath ? n : 0;
        EDLIB_AMD_HIP(stage.alloc((3 * n + nPos + nLen) * sizeof(int)));
        int* h = reinterpret_cast<int*>(stage.p);
        score = h; pool = h + 3 * n; opsLen = h + 3 * n + nPos;
        EDLIB_AMD_HIP(hipMemcpyAsync(h, d_out3_.p, 3 * n * sizeof(int), hipMemcpyDeviceToHost, stream_));
        if (nPos) EDLIB_AMD_HIP(hipMemcpyAsync(h + 3 * n, d_posPool_.p, nPos * sizeof(int), hipMemcpyDeviceToHost, stream_));
        if (nLen) {
            EDLIB_AMD_HIP(hipMemcpyAsync(h + 3 * n + nPos, d_opsLen_.p, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
            if (opsOff[n] > 0) {
                // the op slots (qlen + tlen bytes per unit, filled from the back) land in pinned staging and
                // are read from there by results(): no intermediate host copies
                ops = std::make_shared<PinBuf>();
                EDLIB_AMD_HIP(ops->alloc((size_t)opsOff[n]));
                EDLIB_AMD_HIP(hipMemcpyAsync(ops->p, d_ops_.p, (size_t)opsOff[n], hipMemcpyDeviceToHost, stream_));
                out.opsBufs.push_back(ops);
            }
        }
    }
    const int* count = score + n; const int* last = count + n;
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    if (ring == kWide) {
        const int w = checkWide();
        if (w == 2) return solveChunk(mode, wantPositions, wantPath, units, ua, ub, out, ring, ringH);     // (nothing of `out` was touched yet)
        if (w) return 1;
    }
    lap("chunk: kernels+D2H");

    // exact second pass for units with more end locations than kPosCap
    std::vector<int> ovf; std::vector<long long> ovfOff(1, 0); std::vector<int> ovfPos;
    if (wantPositions && mode != EDLIB_MODE_NW) {
        for (size_t i = 0; i < n; ++i)
            if (count[i] > kPosCap) { ovf.push_back((int)i); ovfOff.push_back(ovfOff.back() + count[i]); }
        if (!ovf.empty()) {
            std::vector<PairDesc> d2(ovf.size());
            // (ring units are rescanned on the strips, every row of every column: a unit of more than 64 blocks -- a banded
            // unit on a ring of 2- / 4-block lanes -- needs the strips' hand-off buffer, which its first scan did not)
            long long aux2 = 0;
            for (size_t j = 0; j < ovf.size(); ++j) {
                d2[j] = descs[ovf[j]];
                d2[j].kinit = score[ovf[j]]; d2[j].posCap = count[ovf[j]]; d2[j].posOff = ovfOff[j];
                if (ring != kWide) { d2[j].auxOff = aux2; if ((d2[j].qlen + 63) / 64 > 64) aux2 += d2[j].tlen; }
            }
            if (ring != kWide) EDLIB_AMD_HIP(d_aux_.ensure((size_t)aux2));
            DevBuf<PairDesc> dd; DevBuf<int> pool2, s2, c2, l2;
            EDLIB_AMD_HIP(dd.alloc(d2.size())); EDLIB_AMD_HIP(pool2.alloc((size_t)ovfOff.back()));
            EDLIB_AMD_HIP(s2.alloc(d2.size())); EDLIB_AMD_HIP(c2.alloc(d2.size())); EDLIB_AMD_HIP(l2.alloc(d2.size()));
            WidePlan wp2;
            if (ring == kWide && planWide(mode, d2.data(), d2.size(), wp2)) return 1;
            EDLIB_AMD_HIP(hipMemcpyAsync(dd.p, d2.data(), d2.size() * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
            PairScanArgs a2 = a;
            a2.aux = d_aux_.p;
            a2.descs = dd.p; a2.numUnits = (int)d2.size(); a2.posPool = pool2.p;
            a2.outScore = s2.p; a2.outCount = c2.p; a2.outLast = l2.p;
            scanTimerStart();
            if (ring == kWide) { if (launchWide(mode, a2, d2.data(), d2.size(), wp2)) return 1; }
            else EDLIB_AMD_HIP(launch_scan_pairs(mode, false, a2, stream_));
            scanTimerStop();
            ovfPos.resize((size_t)ovfOff.back());
            EDLIB_AMD_HIP(hipMemcpyAsync(ovfPos.data(), pool2.p, ovfPos.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            if (ring == kWide) {
                const int w = checkWide();
                if (w == 2) return solveChunk(mode, wantPositions, wantPath, units, ua, ub, out, ring, ringH);
                if (w) return 1;
            }
            stats.overflow_units += (int)ovf.size();
            for (size_t j = 0; j < ovf.size(); ++j) {
                const PairDesc& d = d2[j];
                stats.word_steps += 2LL * ((d.qlen + 63) / 64) * d.tlen;
            }
        }
    }
    size_t oj = 0;
    const bool lists = wantPositions && mode != EDLIB_MODE_NW;
    size_t w = out.posFlat.size();                                     // positions are written in place: one resize per chunk
    if (lists) {
        size_t tot = 0;
        for (size_t i = 0; i < n; ++i) if (score[i] >= 0) tot += (size_t)std::max(count[i], 0);
        out.posFlat.resize(w + tot);
    }
    int* pf = out.posFlat.data();
    for (size_t i = 0; i < n; ++i) {
        const size_t g = ua + i;
        out.score[g] = score[i]; out.count[g] = count[i]; out.last[g] = last[i];
        if (lists && score[i] >= 0) {
            const int* src = pool + i * kPosCap;
            if (oj < ovf.size() && ovf[oj] == (int)i) { src = ovfPos.data() + ovfOff[oj]; ++oj; }
            const int c = std::max(count[i], 0);
            for (int k = 0; k < c; ++k) pf[w + k] = src[k];
            w += (size_t)c;
        }
        out.posStart[g + 1] = (long long)w;
        if (wantPath && ops) {
            out.opsPtr[g] = ops->p + opsOff[i + 1] - opsLen[i];
            out.opsLen[g] = opsLen[i];
        }
    }
    lap("chunk: host gather");
    if (zeroCopy) {          // the views into this chunk's pinned block die with it
        d_out3_.release(); d_posPool_.release(); d_opsLen_.release(); d_opsOff_.release(); d_ops_.release();
        d_outScore_.release(); d_outCount_.release(); d_outLast_.release();
    }
    return 0;
}

// ------------------------------------------------------ one unit on many waves

// The strips of a unit run as a pipeline over `slots` single-wave workgroups whose hand-offs spin, so every workgroup of a
// launch has to be resident: slots * (units per launch) stays within what the device holds (wide_resident_waves).
int Batch::planWide(int mode, PairDesc* descs, size_t n, WidePlan& plan)
{
    if (wideCap_ < 0) wideCap_ = wide_resident_waves(tab_.sigmaT);
    if (wideCap_ <= 0) { set_error("wide kernel: no resident waves (occupancy query failed)"); return 1; }
    int want = 1;
    for (size_t i = 0; i < n; ++i) want = std::max(want, wide_slots_wanted(mode, descs[i].qlen, descs[i].tlen, descs[i].bandT, descs[i].kinit));
    if (const char* e = getenv("EDLIB_AMD_WIDE_SLOTS")) { if (atoi(e) > 0) want = atoi(e); }      // (tests: fewer slots than strips alive)
    // after an aborted launch of this run (workgroups not resident together, a stalled hand-off): one slot per unit -- a wave
    // then only reads granules it wrote itself and never waits, whatever else is on the device
    if (wideSerial_) want = 1;
    plan.slots = std::min(want, wideCap_);
    plan.perLaunch = (size_t)std::max(1, wideCap_ / plan.slots);
    long long words = 0, most = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i % plan.perLaunch == 0) words = 0;
        descs[i].auxOff = words;
        words += wide_stream_words(descs[i].tlen, plan.slots);
        most = std::max(most, words);
    }
    EDLIB_AMD_HIP(d_wide_.ensure((size_t)most));
    if (!d_wabort_.p) { EDLIB_AMD_HIP(d_wabort_.alloc(2)); EDLIB_AMD_HIP(h_wabort_.alloc(sizeof(unsigned))); }    // {abort word, workgroups arrived}
    EDLIB_AMD_HIP(hipMemsetAsync(d_wabort_.p, 0, 2 * sizeof(unsigned), stream_));
    return 0;
}

// The strip pipelines spin on each other, so the workgroups of a wide launch must all be resident -- which the sizing of
// ONE launch guarantees (planWide) and two launches from two host threads sharing the device would not: each could hold
// the slots the other is waiting for until the hand-off timeout.  So wide launches of a process take turns per device:
// the gate is taken before the first launch of a chunk and given back by checkWide() behind the stream synchronisation
// that follows it (or when the batch is reset / destroyed after a failure in between).  A gate, not a std::mutex: it may
// be released by another thread than the one that took it.
namespace {
struct WideGate { std::mutex m; std::condition_variable cv; bool busy = false; };
WideGate& wide_gate(int device) { static WideGate* g = new WideGate[kMaxDevices]; return g[(device >= 0 && device < kMaxDevices) ? device : 0]; }
}
void Batch::wideGateRelease()
{
    if (!wideGateHeld_) return;
    WideGate& g = wide_gate(device_);
    { std::lock_guard<std::mutex> l(g.m); g.busy = false; }
    g.cv.notify_one();
    wideGateHeld_ = false;
}

int Batch::launchWide(int mode, const PairScanArgs& a0, const PairDesc* hostDescs, size_t n, const WidePlan& plan)
{
    if (!wideGateHeld_) {
        WideGate& g = wide_gate(device_);
        std::unique_lock<std::mutex> l(g.m);
        g.cv.wait(l, [&] { return !g.busy; });
        g.busy = true;
        wideGateHeld_ = true;
    }
    for (size_t g0 = 0; g0 < n; g0 += plan.perLaunch) {
        const size_t g1 = std::min(n, g0 + plan.perLaunch);
        const long long words = hostDescs[g1 - 1].auxOff + wide_stream_words(hostDescs[g1 - 1].tlen, plan.slots);
        // every polled word starts at zero (tags are strip + 1): a granule of an earlier launch must never look fresh
        EDLIB_AMD_HIP(hipMemsetAsync(d_wide_.p, 0, (size_t)words * sizeof(unsigned long long), stream_));
        EDLIB_AMD_HIP(hipMemsetAsync(d_wabort_.p + 1, 0, sizeof(unsigned), stream_));     // the residency count of THIS launch
        PairScanArgs a = a0;
        a.descs = a0.descs + g0; a.numUnits = (int)(g1 - g0);
        a.outScore = a0.outScore + g0; a.outCount = a0.outCount + g0; a.outLast = a0.outLast + g0;
        a.wstream = d_wide_.p; a.wabort = d_wabort_.p;
        // (tests: the residency check of a pipelined launch waits for one workgroup more than there are -- it gives up after
        // 0.2 s as if part of the launch had not fitted the device, and the units run again with one slot each)
        a.wideExpect = (!wideSerial_ && getenv("EDLIB_AMD_WIDE_TEST_NOT_RESIDENT")) ? (unsigned)(plan.slots * (g1 - g0) + 1) : 0u;
        EDLIB_AMD_HIP(launch_scan_pairs_wide(mode, a, plan.slots, stream_));
    }
    EDLIB_AMD_HIP(hipMemcpyAsync(h_wabort_.p, d_wabort_.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
    return 0;
}

// 0 = the launches since planWide() ran to their end; 2 = one of them gave up (its workgroups were not on the device
// together, or a hand-off made no progress) and the caller runs its units again, which planWide() now gives one slot each;
// 1 = that second attempt failed as well (cannot happen by construction: reported, not retried)
int Batch::checkWide()
{
    wideGateRelease();
    const unsigned code = h_wabort_.p ? *reinterpret_cast<const unsigned*>(h_wabort_.p) : 0u;
    if (code == 0u) return 0;
    ++stats.wide_retries;
    if (!wideSerial_) { wideSerial_ = true; return 2; }
    set_error(code == 2u ? "wide kernel: the workgroups of a one-slot launch did not all start"
                         : "wide kernel: a hand-off stalled inside a one-slot launch");
    return 1;
}

// ------------------------------------------------- semi-global units on rings

// SHW / HW units of at most 4 (16) blocks share a wave 16 (4) at a time on the lane rings; longer ones take
// the strips.  Same outputs as solve().
// HW is shift-invariant (a scan that starts 2m-1 columns early from the fresh state reproduces the
// exact bottom-row scores of its own columns), and a unit of kernel W is one wave's serial walk over its target: a
// 1 kb query against a 5 Mb chromosome is 5M dependent steps (0.3 s) while 1023 SIMDs idle.  When a batch of HW units
// does not fill the chip, every unit with a long target is cut into target segments (each a unit of its own with a
// warm-up that records nothing: UnitSpec::skip) and the segments' answers are merged: minimum score, the end
// locations of the segments that attain it in order, the last of them.  Results never depend on the cut.
int Batch::solveSemiGlobal(int mode, bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out)
{
    const size_t n = units.size();
    if (mode == EDLIB_MODE_SHW && n > 0 && !(getenv("EDLIB_AMD_SHWBAND") && getenv("EDLIB_AMD_SHWBAND")[0] == '0')) {
        // (queries of up to four blocks sit whole on the smallest ring whatever their threshold: nothing to band)
        bool any = false;
        for (size_t i = 0; i < n && !any; ++i) any = units[i].qlen > 256;
        if (any) return solveShwBanded(wantPositions, units, out);
    }
    if (mode == EDLIB_MODE_HW && n > 0 && !(getenv("EDLIB_AMD_HWBAND") && getenv("EDLIB_AMD_HWBAND")[0] == '0')) {
        // a query in a window not much longer than itself, with a threshold well below its length: the band of solveHwBanded
        // (only the units that qualify: the others keep the path below, which cuts long targets into segments when the batch
        // alone does not fill the chip -- a narrow-window unit next to a few long-target ones used to pull those past it)
        std::vector<char> qual(n);
        size_t nq = 0;
        for (size_t i = 0; i < n; ++i) { qual[i] = hw_band_rows(units[i], std::min(units[i].kinit, 64)) + 64 <= 64LL * ((units[i].qlen + 63) / 64 - 1); nq += qual[i]; }
        if (nq == n) return solveHwBanded(wantPositions, units, out);
        if (nq > 0 && !hwBandSplit_) {
            std::vector<UnitSpec> part[2]; std::vector<size_t> where(n);
            for (size_t i = 0; i < n; ++i) { where[i] = part[qual[i]].size(); part[qual[i]].push_back(units[i]); }
            SolveOut so[2];
            if (solveHwBanded(wantPositions, part[1], so[1])) return 1;
            hwBandSplit_ = true;                                    // (the rest: this function again, without the band)
            const int rc = solveSemiGlobal(mode, wantPositions, part[0], so[0]);
            hwBandSplit_ = false;
            if (rc) return 1;
            out.score.resize(n); out.count.resize(n); out.last.resize(n);
            out.posStart.assign(n + 1, 0); out.posFlat.clear();
            out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
            for (size_t i = 0; i < n; ++i) {
                const SolveOut& p = so[qual[i]];
                const size_t q = where[i];
                out.score[i] = p.score[q]; out.count[i] = p.count[q]; out.last[i] = p.last[q];
                out.posFlat.insert(out.posFlat.end(), p.posFlat.begin() + p.posStart[q], p.posFlat.begin() + p.posStart[q + 1]);
                out.posStart[i + 1] = (long long)out.posFlat.size();
            }
            return 0;
        }
    }
    if (mode != EDLIB_MODE_HW || n == 0 || n >= 4096) return solveSemiGlobalUnits(mode, wantPositions, units, out);
    const long long smax = std::max<long long>(1, 8192 / (long long)n);
    std::vector<UnitSpec> sub; std::vector<int> firstSeg(n + 1, 0); std::vector<int> base;   // base: first recorded column of a segment
    bool any = false;
    for (size_t i = 0; i < n; ++i) {
        const UnitSpec& u = units[i];
        const long long segMin = std::max<long long>(4096, 8LL * u.qlen);
        const long long S = std::max<long long>(1, std::min<long long>(smax, u.tlen / segMin));
        const long long segLen = (u.tlen + S - 1) / S;
        for (long long sg = 0; sg < S; ++sg) {
            const long long c0 = sg * segLen, c1 = std::min<long long>(u.tlen, c0 + segLen);
            if (c0 >= c1) break;
            const long long cw = std::max<long long>(0, c0 - (2LL * u.qlen - 1));
            UnitSpec v = u;
            v.toff = u.toff + cw * u.tstep; v.tlen = (int)(c1 - cw); v.skip = (int)(c0 - cw);
            sub.push_back(v); base.push_back((int)cw);
        }
        firstSeg[i + 1] = (int)sub.size();
        any = any || firstSeg[i + 1] - firstSeg[i] > 1;
    }
    if (!any) return solveSemiGlobalUnits(mode, wantPositions, units, out);
    SolveOut so;
    if (solveSemiGlobalUnits(mode, wantPositions, sub, so)) return 1;
    out.score.assign(n, -1); out.count.assign(n, 0); out.last.assign(n, -1);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    for (size_t i = 0; i < n; ++i) {
        int best = -1;
        for (int q = firstSeg[i]; q < firstSeg[i + 1]; ++q)
            if (so.score[q] >= 0 && (best < 0 || so.score[q] < best)) best = so.score[q];
        out.score[i] = best;
        if (best >= 0)
            for (int q = firstSeg[i]; q < firstSeg[i + 1]; ++q) {
                if (so.score[q] != best) continue;
                out.count[i] += so.count[q];
                out.last[i] = so.last[q] + base[q];
                for (long long k = so.posStart[q]; k < so.posStart[q + 1]; ++k) out.posFlat.push_back(so.posFlat[k] + base[q]);
            }
        out.posStart[i + 1] = (long long)out.posFlat.size();
    }
    return 0;
}

// SHW with a threshold: D[i][j] >= |i - j|, so a scan with threshold K only needs the diagonals [-K, K] and the first
// m + K columns (the reference's band for SHW, edlib.cpp:562, 602-630, written for a fixed k).  A unit with a real
// threshold (the reverse scans of HW start locations run with k = the distance, :253-257; calls with k >= 0) is scanned
// inside that band once; an open unit (k = -1: threshold m) climbs levels K = 256, 1024, 4096 ... like the reference
// doubles k (:197-217), the answer being exact as soon as some column scores <= K.  What it buys: the smallest ring that
// holds the BAND instead of the whole query (a 10 kb reverse scan with k = 100 on an 8-lane... here 16-lane ring, four
// units per wave, instead of five 2048-row strips), and m + K columns instead of 2 m.
int Batch::solveShwBanded(bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out)
{
    const size_t n = units.size();
    out.score.assign(n, -1); out.count.assign(n, 0); out.last.assign(n, -1);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    std::vector<long long> kcur(n);
    std::vector<std::vector<int>> posOf(wantPositions ? n : 0);
    std::vector<size_t> rest;
    // a level K can only find a column when row m-1 is inside its band somewhere: T >= m - K (else the kernels would never
    // start the last block); levels that cannot are skipped, a unit whose own threshold cannot has no answer (-1)
    auto reachable = [&](const UnitSpec& u, long long K) { return (long long)u.tlen >= (long long)u.qlen - K; };
    for (size_t i = 0; i < n; ++i) {
        const UnitSpec& u = units[i];
        if (!reachable(u, std::min(u.kinit, u.qlen))) continue;
        kcur[i] = u.kinit < u.qlen ? u.kinit : 256;
        while (kcur[i] < u.kinit && !reachable(u, kcur[i])) kcur[i] *= 4;
        rest.push_back(i);
    }
    while (!rest.empty()) {
        std::vector<UnitSpec> sel; sel.reserve(rest.size());
        for (size_t i : rest) {
            UnitSpec u = units[i];
            const long long K = std::min<long long>(kcur[i], u.kinit);
            u.kinit = (int)K;
            u.band = K < u.qlen ? 1 : 0;
            u.tlen = (int)std::min<long long>(u.tlen, (long long)u.qlen + K);
            sel.push_back(u);
        }
        SolveOut so;
        if (solveSemiGlobalUnits(EDLIB_MODE_SHW, wantPositions, sel, so)) return 1;
        std::vector<size_t> again;
        for (size_t q = 0; q < sel.size(); ++q) {
            const size_t i = rest[q];
            if (so.score[q] >= 0 || sel[q].kinit >= units[i].kinit) {          // exact / the caller's own threshold found nothing
                out.score[i] = so.score[q]; out.count[i] = so.count[q]; out.last[i] = so.last[q];
                if (wantPositions) posOf[i].assign(so.posFlat.begin() + so.posStart[q], so.posFlat.begin() + so.posStart[q + 1]);
                continue;
            }
            kcur[i] = 4LL * sel[q].kinit;
            again.push_back(i);
        }
        rest.swap(again);
    }
    if (wantPositions)
        for (size_t i = 0; i < n; ++i) {
            out.posFlat.insert(out.posFlat.end(), posOf[i].begin(), posOf[i].end());
            out.posStart[i + 1] = (long long)out.posFlat.size();
        }
    return 0;
}

// HW with a threshold, as a STATIC band (the reference narrows HW with its first..lastBlock bookkeeping, edlib.cpp:562,
// 602-630, block 0 kept alive): an alignment of the whole query with cost <= K starts at a column j0 in [0, T - m + K] and
// stays within K diagonals of j0, so only the diagonals [-K, (T - m) + 2 K] are needed -- (T - m) + 3 K + 1 rows per column
// instead of m.  That pays for a query in a window not much longer than itself (the verification step of a mapper: 1 kb in
// 1.2 kb at k = 20 is 261 rows: an 8-lane ring, eight units per wave, instead of 16 lanes and four).  A unit with a real
// threshold is scanned inside it once; an open unit (k = -1: threshold m) climbs K = 64, 256, 1024 ... like the reference
// doubles k (:197-217), exact as soon as some column scores <= K; a level whose band is no narrower than the query is
// the plain scan, which ends the climb.  Units with long targets keep the plain path (target segments, the piece filter).
long long hw_band_rows(const UnitSpec& u, long long K) { return std::max(0, u.tlen - u.qlen) + 3 * K + 1; }

int Batch::solveHwBanded(bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out)
{
    const size_t n = units.size();
    out.score.assign(n, -1); out.count.assign(n, 0); out.last.assign(n, -1);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    std::vector<long long> kcur(n);
    std::vector<std::vector<int>> posOf(wantPositions ? n : 0);
    std::vector<size_t> rest;
    // a level K pays while its band is at least a block narrower than the query
    auto narrow = [&](const UnitSpec& u, long long K) { return hw_band_rows(u, K) + 64 <= 64LL * ((u.qlen + 63) / 64 - 1); };
    for (size_t i = 0; i < n; ++i) {
        const UnitSpec& u = units[i];
        kcur[i] = u.kinit < u.qlen ? u.kinit : 64;                    // the caller's threshold, or the first level of an open unit
        rest.push_back(i);
    }
    while (!rest.empty()) {
        std::vector<UnitSpec> sel; sel.reserve(rest.size());
        for (size_t i : rest) {
            UnitSpec u = units[i];
            const long long K = std::min<long long>(kcur[i], u.kinit);
            // (T < m - K: no alignment of cost <= K exists; the band would never reach row m - 1 -- the plain scan answers)
            const bool banded = K < u.kinit || u.kinit < u.qlen ? (narrow(u, K) && (long long)u.tlen >= (long long)u.qlen - K) : false;
            u.kinit = banded ? (int)K : units[i].kinit;
            u.band = banded ? 1 : 0;
            sel.push_back(u);
        }
        SolveOut so;
        if (solveSemiGlobalUnits(EDLIB_MODE_HW, wantPositions, sel, so)) return 1;
        std::vector<size_t> again;
        for (size_t q = 0; q < sel.size(); ++q) {
            const size_t i = rest[q];
            if (!sel[q].band || so.score[q] >= 0 || sel[q].kinit >= units[i].kinit) {   // plain / exact / the caller's own threshold found nothing
                out.score[i] = so.score[q]; out.count[i] = so.count[q]; out.last[i] = so.last[q];
                if (wantPositions) posOf[i].assign(so.posFlat.begin() + so.posStart[q], so.posFlat.begin() + so.posStart[q + 1]);
                continue;
            }
            kcur[i] = 4LL * sel[q].kinit;
            again.push_back(i);
        }
        rest.swap(again);
    }
    if (wantPositions)
        for (size_t i = 0; i < n; ++i) {
            out.posFlat.insert(out.posFlat.end(), posOf[i].begin(), posOf[i].end());
            out.posStart[i + 1] = (long long)out.posFlat.size();
        }
    return 0;
}

int Batch::solveSemiGlobalUnits(int mode, bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out)
{
    const size_t n = units.size();
    const bool ringsOff = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
    // units of up to 4 / 16 blocks on 4- / 16-lane rings, up to 32 / 64 blocks on 16-lane rings whose lanes hold 2 / 4
    // blocks (four units per wave, every lane busy: a 1025-base query on the strips uses 17 of a wave's 64 lanes), the
    // rest on the strips
    // more than 64 blocks: the strips as a pipeline over many waves (wide_kernels.hip) instead of one wave walking them
    // one after the other
    static const int rings[7] = {4, 8, 16, 16, 16, 0, kWide}, ringH[7] = {1, 1, 1, 2, 4, 1, 1};
    const int NG = 7;
    std::vector<int> grp(n, 5);
    size_t cnt[NG] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        const int nb = (units[i].qlen + 63) / 64;
        if (!ringsOff) grp[i] = nb <= 4 ? 0 : (nb <= 8 ? 1 : (nb <= 16 ? 2 : (nb <= 32 ? 3 : (nb <= 64 ? 4 : 5))));
        if (nb > 64) grp[i] = 6;
        // a banded SHW / HW unit (UnitSpec::band) needs the ring that holds its band, not its query
        if ((mode == EDLIB_MODE_SHW || mode == EDLIB_MODE_HW) && units[i].band && !ringsOff) {
            // (the SHW band [-K, K] is 2 K + 1 rows wide, twice the NW band of the same threshold: ring_max_k / 2; the HW band
            // [-K, (T - m) + 2 K] is (T - m) + 3 K + 1 rows wide)
            const long long K2 = mode == EDLIB_MODE_SHW ? 2LL * units[i].kinit : hw_band_rows(units[i], units[i].kinit) - 1;
            if (nb > 4 && K2 <= ring_max_k(4)) grp[i] = 0;
            else if (nb > 8 && K2 <= ring_max_k(8)) grp[i] = 1;
            else if (nb > 16 && K2 <= ring_max_k(16)) grp[i] = 2;
            else if (nb > 32 && K2 <= ring_max_k(16, 2)) grp[i] = 3;
            else if (nb > 64 && K2 <= ring_max_k(16, 4)) grp[i] = 4;
        }
        ++cnt[grp[i]];
    }
    for (int g = 0; g < NG; ++g)
        if (cnt[g] == n) return solve(mode, wantPositions, false, units, out, rings[g], ringH[g]);   // the usual case: one kind
    SolveOut part[NG];
    std::vector<size_t> where(n);
    for (int g = 0; g < NG; ++g) {
        if (!cnt[g]) continue;
        std::vector<UnitSpec> sel; sel.reserve(cnt[g]);
        for (size_t i = 0; i < n; ++i) if (grp[i] == g) { where[i] = sel.size(); sel.push_back(units[i]); }
        if (solve(mode, wantPositions, false, sel, part[g], rings[g], ringH[g])) return 1;
    }
    out.score.resize(n); out.count.resize(n); out.last.resize(n);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    for (size_t i = 0; i < n; ++i) {
        const SolveOut& p = part[grp[i]];
        const size_t q = where[i];
        out.score[i] = p.score[q]; out.count[i] = p.count[q]; out.last[i] = p.last[q];
        out.posFlat.insert(out.posFlat.end(), p.posFlat.begin() + p.posStart[q], p.posFlat.begin() + p.posStart[q + 1]);
        out.posStart[i + 1] = (long long)out.posFlat.size();
    }
    return 0;
}

// ------------------------------------------------------ a level that takes every unit

// A big batch whose units all start on one ring (config 4: 100,000 x 10 kb pairs on 21-lane rings) spent 1.6 ms of a 22 ms
// step around its scan: 0.3 ms selecting the level's units (all of them), 0.33 ms writing 100,000 descriptors, 0.35 ms
// sending them up, 0.75 ms building Peq -- behind a divergence probe that leaves the chip idle (16 waves, 0.5 ms).  So:
// the units' offsets and lengths stay on the device between runs (LevelSpec: input layout, like the offset arrays of
// init()); BEFORE the probe a third stream writes plain descriptors from them and builds the Peq of every unit, and a level
// that then takes every unit has its descriptors rewritten by a kernel (threshold, ring) and scans at once.  Units the
// level leaves open go on to the next one the usual way.  Returns 0 with levelAllReady_ set when the Peq is on its way.
int Batch::prepareLevelAll(const std::vector<UnitSpec>& units)
{
    levelAllReady_ = false;
    const size_t n = units.size();
    if (levelSpecsVersion_ != pairSpecsVersion_) {
        EDLIB_AMD_HIP(h_levelSpecs_.alloc(n * sizeof(LevelSpec)));
        LevelSpec* ls = reinterpret_cast<LevelSpec*>(h_levelSpecs_.p);
        long long words = 0;
        bool plain = true;
        for (size_t i = 0; i < n; ++i) {
            const UnitSpec& u = units[i];
            plain = plain && u.qstep == 1 && u.tstep == 1 && u.skip == 0 && u.band == 0;
            ls[i] = LevelSpec{u.qoff, u.toff, words, u.qlen, u.tlen};
            words += (long long)((u.qlen + 63) / 64) * tab_.sigmaT;
        }
        levelPeqWords_ = plain ? words : -1;
        levelSpecsVersion_ = pairSpecsVersion_;
        if (!plain) return 0;
        EDLIB_AMD_HIP(d_levelSpecs_.ensure(n));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_levelSpecs_.p, ls, n * sizeof(LevelSpec), hipMemcpyHostToDevice, stream_));
    }
    if (levelPeqWords_ < 0 || levelPeqWords_ * 8 > (4LL << 30)) return 0;
    EDLIB_AMD_HIP(d_descsAll_.ensure(n));
    EDLIB_AMD_HIP(d_peqAll_.ensure((size_t)levelPeqWords_));
    if (!aux_) {                                                    // (lowest priority: the probe's 16 waves are dispatched first)
        int least = 0, greatest = 0;
        EDLIB_AMD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        EDLIB_AMD_HIP(hipStreamCreateWithPriority(&aux_, hipStreamNonBlocking, least));
    }
    EDLIB_AMD_HIP(evLevelIn_.create()); EDLIB_AMD_HIP(evLevelPeq_.create());
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(hipEventRecord(evLevelIn_.e, stream_));           // the inputs, the specs and the equality table went up on stream_
    EDLIB_AMD_HIP(hipStreamWaitEvent(aux_, evLevelIn_.e, 0));
    EDLIB_AMD_HIP(launch_fill_level_descs(d_levelSpecs_.p, (int)n, 0, 0, 0, 0, d_descsAll_.p, aux_));
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_descsAll_.p, (int)n, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT, d_peqAll_.p, aux_));
    EDLIB_AMD_HIP(hipEventRecord(evLevelPeq_.e, aux_));
    levelAllReady_ = true;
    return 0;
}

// the scan of a level that takes every unit: scores into h_levelScore_ (pinned, units.size() ints)
int Batch::runLevelAll(const std::vector<UnitSpec>& units, int ring, int ringH, int ringBlocks, int cap, int kcap, int nbMax)
{
    const size_t n = units.size();
    Lap lap;
    stats.path |= 2;
    if (!d_out3_.owned) d_out3_.release();
    if (!d_posPool_.owned) d_posPool_.release();
    EDLIB_AMD_HIP(d_out3_.ensure(3 * n));
    EDLIB_AMD_HIP(d_posPool_.ensure(1));
    if (h_levelScore_.n < n * sizeof(int)) EDLIB_AMD_HIP(h_levelScore_.alloc(n * sizeof(int)));
    d_outScore_.alias(d_out3_.p, n); d_outCount_.alias(d_out3_.p + n, n); d_outLast_.alias(d_out3_.p + 2 * n, n);
    EDLIB_AMD_HIP(hipStreamWaitEvent(stream_, evLevelPeq_.e, 0));   // (the descriptors below replace the ones the Peq build read)
    EDLIB_AMD_HIP(launch_fill_level_descs(d_levelSpecs_.p, (int)n, kcap, ringBlocks, cap, ring, d_descsAll_.p, stream_));
    PairScanArgs a{};
    a.descs = d_descsAll_.p; a.numUnits = (int)n; a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peqAll_.p; a.aux = d_aux_.p;
    a.peqRowStride = peq_row_stride(nbMax);
    a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
    if (getenv("EDLIB_AMD_PEQFULL") && getenv("EDLIB_AMD_PEQFULL")[0] == '0') a.peqFullStride = 0;      // (as solveChunk)
    a.outScore = d_outScore_.p; a.outCount = d_outCount_.p; a.outLast = d_outLast_.p; a.posPool = d_posPool_.p;
    a.wordSteps = ringStepsCounter();
    scanTimerStart();
    EDLIB_AMD_HIP(launch_scan_pairs_ring(ring, EDLIB_MODE_NW, false, a, stream_, ringH));
    scanTimerStop();
    EDLIB_AMD_HIP(hipMemcpyAsync(h_levelScore_.p, d_outScore_.p, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
    if (whileScanning_) { auto f = std::move(whileScanning_); whileScanning_ = nullptr; f(); }
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    lap("nw level (every unit): kernels + D2H");
    return 0;
}

// ------------------------------------------------------ the lane-per-pair level

// Packs every unit of the batch for the lane-per-pair scan (lanepair.hpp) on the third stream, beside the divergence probe:
// query and target bit planes, the "not for this kernel" flags, and -- when the units the count is wanted for are exactly
// these pairs -- alphabetLength on the way (the count used to read the batch's 2 GB a second time).  Returns 0 with
// laneReady_ set when the pack is on its way.
int Batch::prepareLaneLevel(const std::vector<UnitSpec>& units)
{
    laneReady_ = false;
    // more than four target symbols in the batch (a genome with N, soft-masked stretches): every unit is coded from its own
    // target's bytes -- identity equality only; with additional equalities a byte may equal several symbols
    const bool perUnit = tab_.sigmaT > 4;
    if (perUnit && (!tab_.eq8.empty() || tab_.sigmaT > 16)) return 0;       // (protein-sized alphabets: hardly a unit would stay within four)
    const size_t n = units.size();
    if (laneSpecsVersion_ != pairSpecsVersion_) {
        EDLIB_AMD_HIP(h_laneUnits_.alloc(n * sizeof(lanepair::LaneUnit)));
        lanepair::LaneUnit* lu = reinterpret_cast<lanepair::LaneUnit*>(h_laneUnits_.p);
        long long pw = 0, tw = 0;
        bool plain = true;
        for (size_t i = 0; i < n; ++i) {
            const UnitSpec& u = units[i];
            plain = plain && u.qstep == 1 && u.tstep == 1 && u.skip == 0 && u.band == 0 && u.qlen > 0 && u.tlen > 0;
            lu[i] = lanepair::LaneUnit{u.qoff, u.toff, pw, tw, u.qlen, u.tlen};
            pw += (u.qlen + 31) / 32; tw += (u.tlen + 31) / 32;
        }
        plain = plain && pw < (1LL << 31) && tw < (1LL << 31);         // (32-bit offsets into the plane pools)
        lanePlaneWords_ = plain ? pw : -1; laneTgtWords_ = tw;
        laneSpecsVersion_ = pairSpecsVersion_;
        if (!plain) return 0;
        EDLIB_AMD_HIP(d_laneUnits_.ensure(n));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_laneUnits_.p, lu, n * sizeof(lanepair::LaneUnit), hipMemcpyHostToDevice, stream_));
    }
    if (lanePlaneWords_ < 0) return 0;
    EDLIB_AMD_HIP(d_lanePlanes_.ensure((size_t)lanePlaneWords_ + 1));
    EDLIB_AMD_HIP(d_laneTgts_.ensure((size_t)laneTgtWords_ + 1));
    EDLIB_AMD_HIP(d_laneFlags_.ensure(n));
    EDLIB_AMD_HIP(d_laneScore_.ensure(n));
    if (h_laneScore_.n < n * sizeof(int)) EDLIB_AMD_HIP(h_laneScore_.alloc(n * sizeof(int)));
    if (!aux_) {
        int least = 0, greatest = 0;
        EDLIB_AMD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        EDLIB_AMD_HIP(hipStreamCreateWithPriority(&aux_, hipStreamNonBlocking, least));
    }
    // alphabetLength on the way, when the count is wanted for exactly these units in this order
    if (alphaIsPairsVersion_ != pairSpecsVersion_) { alphaIsPairs_ = alphaUnits_ == pairUnits_; alphaIsPairsVersion_ = pairSpecsVersion_; }
    const bool fuseAlpha = alphaDeferred_ && alphaIsPairs_ && !alphaOnHost_ && !alphaUnits_.empty() && &units == &pairSpecs_;
    if (fuseAlpha && alphabetBuffers()) return 1;
    EDLIB_AMD_HIP(evLevelIn_.create()); EDLIB_AMD_HIP(evLanePack_.create());
    EDLIB_AMD_HIP(hipEventRecord(evLevelIn_.e, stream_));           // the inputs and the units went up on stream_
    EDLIB_AMD_HIP(hipStreamWaitEvent(aux_, evLevelIn_.e, 0));
    lanepair::PackArgs pa{};
    pa.qpool = d_qpool_.p; pa.tpool = d_tpool_.p; pa.tlut = d_tlut_.p; pa.eqtbl = d_eqtbl_.p; pa.sigmaT = tab_.sigmaT; pa.perUnit = perUnit ? 1 : 0;
    pa.units = d_laneUnits_.p; pa.numUnits = (int)n; pa.planes = d_lanePlanes_.p; pa.tgts = d_laneTgts_.p;
    pa.flags = d_laneFlags_.p; pa.alphaOut = fuseAlpha ? d_alphaOut_.p : nullptr;
    EDLIB_AMD_HIP(launch_lanepair_pack(pa, aux_));
    EDLIB_AMD_HIP(hipEventRecord(evLanePack_.e, aux_));
    if (fuseAlpha) {                                                // collected from the side stream like the count's own launch
        alphaDeferred_ = false;
        EDLIB_AMD_HIP(hipStreamWaitEvent(side_, evLanePack_.e, 0));
        EDLIB_AMD_HIP(evB_.create());
        EDLIB_AMD_HIP(hipEventRecord(evB_.e, side_));
        EDLIB_AMD_HIP(hipMemcpyAsync(alphaPin_.p, d_alphaOut_.p, n * sizeof(int), hipMemcpyDeviceToHost, side_));
        alphaPending_ = true;
    }
    laneReady_ = true;
    return 0;
}

// the scan of the lane-per-pair level: computed distances into h_laneScore_ (pinned, units.size() ints; exact iff <= K)
int Batch::runLaneLevel(const std::vector<UnitSpec>& units, double rate, int kcap, int W)
{
    const size_t n = units.size();
    Lap lap;
    stats.path |= 2;
    lanepair::ScanArgs a{};
    a.units = d_laneUnits_.p; a.flags = d_laneFlags_.p; a.numUnits = (int)n; a.planes = d_lanePlanes_.p; a.tgts = d_laneTgts_.p;
    a.rate = (float)rate; a.kcap = kcap; a.kmax = lanepair::window_max_k(W); a.outScore = d_laneScore_.p; a.wordSteps = ringStepsCounter(); a.denySeed = 0;
    if (const char* e = getenv("EDLIB_AMD_LANEPAIR")) { if (e[0] == 's') a.denySeed = 0xffffffffu; }           // ("static": A/B against the static band)
    EDLIB_AMD_HIP(hipStreamWaitEvent(stream_, evLanePack_.e, 0));
    scanTimerStart();
    EDLIB_AMD_HIP(launch_lanepair_scan(a, W, stream_));
    scanTimerStop();
    EDLIB_AMD_HIP(hipMemcpyAsync(h_laneScore_.p, d_laneScore_.p, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
    if (whileScanning_) { auto f = std::move(whileScanning_); whileScanning_ = nullptr; f(); }
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    lap("nw lane level: kernels + D2H");
    return 0;
}

// ------------------------------------------------------ NW distance levels

// The reference finds the NW distance by doubling k from 64 until the banded scan succeeds
// (edlib.cpp:197-217); any threshold >= the distance gives the same answer, so the levels here are the
// ring sizes of scan_pairs_ring_kernel: K = 196 on 4-lane rings (16 units per wave), 456 on 8, 976 on 16, 1301 on 21
// (three units per wave), 2016 on half waves, 4031 on whole waves (ring_max_k), then the unbanded strips.  A unit whose blocks
// all fit a ring is exact on it for any distance (threshold max(m, T)).  A failed level is pure waste when the whole
// batch is divergent, so larger batches first measure the divergence of 64 strided units on their 512-base prefixes
// (one small launch) and every unit starts at the level that holds its extrapolated distance.  The estimate only picks the starting level; results never depend on it.
int Batch::solveGlobalDistances(const std::vector<UnitSpec>& units, std::vector<int>& score, std::vector<OpsOut>* paths)
{
    const size_t n = units.size();
    score.assign(n, -1);
    if (paths) { paths->clear(); paths->resize(n); }
    if (n == 0) return 0;
    const bool bandOff = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
    // ring levels (lanes, blocks per lane); level nl = unbanded strips.  Rings whose lanes hold 2 / 4 blocks (16 x 2: four
    // units per wave, DPP carry) were measured here in round 3 and lost: a ring computes ALL its rows every step, and
    // 16 x 2 = 2048 rows for a band that needs ~1300 is 52 % more block updates than the 21-lane ring's 1344, which the
    // cheaper step (105 against 119 SIMD cycles per block) does not pay back: config 4 24.7 ms of scans against 20.7.
    // They serve the semi-global units of 17..64 blocks instead (solveSemiGlobalUnits), where the alternative is a strip
    // that uses 17 of 64 lanes.
    static const int ringOf[kNumRings] = {4, 8, 16, 21, 32, 64}, ringH[kNumRings] = {1, 1, 1, 1, 1, 1};
    auto cap_of = [&](int l) { return ring_max_k(ringOf[l], ringH[l]); };
    auto blocks_of = [&](int l) { return ringOf[l] * ringH[l]; };
    const int nl = kNumRings;                                           // ring levels; level nl = unbanded strips
    const int kInf = 0x3fffffff;
    const int kcap = cfg_.k >= 0 ? cfg_.k : kInf;                       // answers above the caller's k are all alike
    auto blocks = [&](size_t i) { return (units[i].qlen + 63) / 64; };

    double rate = 0.0;                                                  // edits per base, median of the sample
    // the shape of the batch (its extremes): a property of the inputs like the specs themselves, kept with them
    PairShape shapeNow;
    PairShape& shape = &units == &pairSpecs_ ? shape_ : shapeNow;
    if (&units != &pairSpecs_ || shapeVersion_ != pairSpecsVersion_) {
        shape = PairShape{};
        shape.minBlocks = (size_t)1 << 40; shape.minLenLo = 0x7fffffff;
        for (size_t i = 0; i < n; ++i) {
            const size_t nbI = (size_t)blocks(i);
            const int lo = std::min(units[i].qlen, units[i].tlen);
            shape.minBlocks = std::min(shape.minBlocks, nbI); shape.maxBlocks = std::max(shape.maxBlocks, nbI);
            shape.minLenLo = std::min(shape.minLenLo, lo); shape.minLenHi = std::max(shape.minLenHi, lo);
            shape.maxDiff = std::max(shape.maxDiff, std::abs(units[i].qlen - units[i].tlen));
        }
        if (&units == &pairSpecs_) shapeVersion_ = pairSpecsVersion_;
    }
    const size_t maxBlocks = shape.maxBlocks;
    // (units of at most 16 blocks climb cheap levels -- the 16-lane ring holds them whole -- and skip the probe)
    levelAllReady_ = false; laneReady_ = false;
    if (n >= 256 && maxBlocks > 16 && !bandOff && !getenv("EDLIB_AMD_NOPROBE")) {
        // (a big batch: the Peq of every unit is built next to the probe -- prepareLevelAll)
        // (only for units of like lengths: a level takes every unit when the batch's extremes land on the same ring)
        const bool bigAlike = paths == nullptr && n >= 8192 && &units == &pairSpecs_ && 4LL * shape.minLenHi <= 5LL * shape.minLenLo;
        // four target symbols at most: the lane-per-pair level (a lane owns a unit; prepareLaneLevel); else the rings
        const bool laneOff = getenv("EDLIB_AMD_LANEPAIR") && getenv("EDLIB_AMD_LANEPAIR")[0] == '0';
        if (bigAlike && !laneOff && prepareLaneLevel(units)) return 1;
        if (bigAlike && !laneReady_ && prepareLevelAll(units)) return 1;
        // 64 strided units, the first 512 bases of the query against the first 512 + 128 of the target in PREFIX mode
        // (the best end column is free: a global alignment of two equally cut prefixes would add the indel drift at
        // the cut to the count, about one edit in a hundred bases at ONT-like rates).  16 waves of ~650 dependent steps:
        // the median of 64 counts of ~60 edits is good to 2 %, and the estimate only picks the first level (1 kb prefixes,
        // rounds 2-4, took twice as long for 1.4 %).
        const int np = 64, cut = 512;
        std::vector<UnitSpec> probe(np);
        for (int i = 0; i < np; ++i) {
            UnitSpec u = units[(size_t)((long long)i * n / np)];
            u.qlen = std::min(u.qlen, cut); u.tlen = std::min(u.tlen, cut + 128);
            u.kinit = u.qlen;                                           // 16 blocks at most: the whole matrix on a 16-lane ring
            probe[i] = u;
        }
        SolveOut so;
        if (solve(EDLIB_MODE_SHW, false, false, probe, so, 16)) return 1;
        std::vector<double> r(np);
        for (int i = 0; i < np; ++i) r[i] = (double)std::max(so.score[i], 0) / std::max(1, probe[i].qlen);
        std::sort(r.begin(), r.end());
        rate = r[np / 2];
    }
    if (alphaDeferred_) {                                               // alphabetLength: behind the Peq build of the batch, if there is one
        alphaDeferred_ = false;
        if (alphabetLengthsBegin(levelAllReady_ ? evLevelPeq_.e : nullptr)) return 1;
    }
    // First level of a unit: the smallest ring that holds all its blocks or its extrapolated distance.  The distance
    // of a unit of length L at rate r scatters like a sum of L Bernoulli trials (sigma = sqrt(r L)).  A ring of G
    // lanes costs G / 64 of a wave per unit, so trying the smaller ring first pays as long as fewer than a quarter
    // to a half of the units fail on it and move up: the estimate is the mean plus half a sigma (10 kb pairs at
    // 11.4 % sit under the 21-lane ring's 1301 -- three units per wave instead of two).
    // (est = mean + sqrt(mean) / 2 + 8 <= cap is a bound on the mean: solved once per level, so that a unit costs a
    // multiply-add and a few compares -- the square root per unit was 2 ms of host time per 100,000 units)
    // A handful of LONG units (the reference's 1 Mb Chromosome pairs, test_data/perf_tests.sh:180-191): a level costs its
    // ~T dependent steps whether it succeeds or not (0.1 s per Mb), so each unit gets its own estimate from its first 4 kb
    // (PREFIX mode on a 16-lane ring of 4-block lanes: ~1 ms) instead of climbing.
    const bool wideLevel = paths == nullptr;                            // what follows the rings: the wide band (with the column store: the strips)
    // A handful of units (edlibAlign() on a long pair is one) are bound by DEPENDENT STEPS, not by work: a ring scan is
    // ~T steps of 0.12 us on one wave whether it succeeds or not, the wide kernel's two half scans are T / 2 steps of
    // 0.074 us on as many waves as the band is tall.  So when the strips of all units' whole matrices fit the resident
    // waves, units of 4 kb and more skip the rings: straight to two half scans, over the WHOLE matrix (no estimate, no
    // ladder, always exact) while that is at most 2e10 cells, inside a band from the unit's own first 4 kb beyond that
    // (the reference's 1 Mb Chromosome pairs, test_data/perf_tests.sh:180-191; PREFIX mode on a 16-lane ring of 4-block
    // lanes: ~1 ms against 40 ms per pass).
    std::vector<uint8_t> direct;
    std::vector<double> unitRate;
    if (wideLevel && rate == 0.0 && n <= 64 && !bandOff) {
        if (wideCap_ < 0) wideCap_ = wide_resident_waves(tab_.sigmaT);
        long long waves = 0;
        for (size_t i = 0; i < n; ++i)
            if (std::min(units[i].qlen, units[i].tlen) >= 4096)        // whole matrix: every strip is alive; a band: a few dozen
                waves += (double)units[i].qlen * (double)units[i].tlen <= 2e10 ? 2LL * ((units[i].qlen + 2047) / 2048) : 96;
        if (waves > 0 && waves <= wideCap_) {
            direct.assign(n, 0);
            for (size_t i = 0; i < n; ++i) direct[i] = std::min(units[i].qlen, units[i].tlen) >= 4096;
        }
    }
    auto whole_ok = [&](size_t i) { return (double)units[i].qlen * (double)units[i].tlen <= 2e10; };
    if (rate == 0.0 && n <= 512 && !bandOff && !getenv("EDLIB_AMD_NOPROBE")) {
        std::vector<UnitSpec> probe; std::vector<size_t> who;
        const int cut = 4096;
        for (size_t i = 0; i < n; ++i)
            if (direct.empty() ? std::min(units[i].qlen, units[i].tlen) >= 32768 : (direct[i] && !whole_ok(i))) {
                UnitSpec u = units[i];
                u.qlen = cut; u.tlen = std::min(u.tlen, cut + 512); u.kinit = cut;       // (never past the unit's own target)
                probe.push_back(u); who.push_back(i);
            }
        if (!probe.empty()) {
            SolveOut so;
            if (solve(EDLIB_MODE_SHW, false, false, probe, so, 16, 4)) return 1;
            unitRate.assign(n, 0.0);
            for (size_t q = 0; q < probe.size(); ++q) unitRate[who[q]] = (double)std::max(so.score[q], 0) / cut;
        }
    }
    // (the length difference of a pair is part of its edits -- 10 kb at 4 % insertions and 4 % deletions differ by 28 bases
    // sigma -- and already inside the rate; only what exceeds three sigma of such a drift counts as a tail to add: rounds 2-4
    // added all of it, which sent the 7 % of config 4's pairs with the largest drift past the ring that holds them)
    auto mean_of = [&](size_t i) {
        const UnitSpec& u = units[i];
        const double base = (unitRate.empty() ? rate : unitRate[i]) * std::min(u.qlen, u.tlen);
        const double diff = std::abs(u.qlen - u.tlen);
        return diff * diff <= 9.0 * base ? base : base + diff - 3.0 * std::sqrt(base);        // (no square root in the usual case)
    };
    double meanCap[kNumRings + 1];
    auto mean_cap = [](double cap) { if (cap < 8) return -1.0; const double r = (-0.5 + std::sqrt(0.25 + 4.0 * (cap - 8.0))) / 2.0; return r * r; };
    for (int l = 0; l < nl; ++l) meanCap[l] = mean_cap(std::min<double>(cap_of(l), kcap));
    meanCap[nl] = mean_cap(2.0 * ring_max_k(64));
    int levelOfKcap = nl;                                               // est = kcap when the caller's k is the smaller one
    for (int l = nl - 1; l >= 0; --l) if (kcap <= cap_of(l)) levelOfKcap = l;
    auto first_level = [&](size_t i) {
        const double mean = mean_of(i);
        const int nbI = blocks(i);
        for (int l = 0; l < nl; ++l)
            if (nbI <= blocks_of(l) || mean <= meanCap[l] || l >= levelOfKcap) return l;
        // above every ring: the band on many waves.  (With the column store -- fused PATH levels -- what follows the rings is
        // the unbanded strips, nstrips times the work: the last ring is still tried while the estimate is within twice its limit.)
        if (!wideLevel) return (mean <= meanCap[nl] || kcap <= 2.0 * ring_max_k(64)) ? nl - 1 : nl;
        return nl;
    };
    std::vector<int>& lvl = lvlScratch_;
    lvl.resize(n);
    // A few units do not fill the chip at any ring size: a level then costs its ~T dependent steps on one wave
    // whether it succeeds or not (a 10 kb pair: 1.7 ms per level), so units with more blocks than a ring holds
    // go straight to whole-wave rings (K = 4031) instead of climbing.
    const bool fewUnits = n <= 512 && rate == 0.0;
    std::vector<size_t> atLevel(nl + 2, 0);
    // A batch whose shortest and longest unit start on the same ring level (config 4: 100,000 pairs of 10 kb +- 1 %): the
    // level of a unit is monotone in its estimate, and no unit is small enough to sit whole on a smaller ring -- one
    // evaluation at each extreme instead of 100,000 (0.4 ms).
    int uniformLevel = -1;
    if (!bandOff && !fewUnits && rate > 0.0 && unitRate.empty() && direct.empty()) {
        auto by_mean = [&](double mean) {
            for (int l = 0; l < nl; ++l) if (mean <= meanCap[l] || l >= levelOfKcap) return l;
            return nl;
        };
        const double lo = rate * shape.minLenLo, base = rate * shape.minLenHi;
        const double hi = (double)shape.maxDiff * shape.maxDiff <= 9.0 * lo ? base : base + shape.maxDiff;
        const int L = by_mean(lo);
        if (L < nl && by_mean(hi) == L && (L == 0 || shape.minBlocks > (size_t)blocks_of(L - 1))) uniformLevel = L;
    }
    if (uniformLevel >= 0) {
        std::fill(lvl.begin(), lvl.end(), uniformLevel);
        atLevel[uniformLevel] = n;
    } else
    {
        int lastQ = -1, lastT = -1, lastL = 0;                           // batches of equal shapes: one evaluation
        for (size_t i = 0; i < n; ++i) {
            if (units[i].qlen != lastQ || units[i].tlen != lastT || !direct.empty()) {
                lastQ = units[i].qlen; lastT = units[i].tlen;
                lastL = bandOff ? nl : first_level(i);
                if (!bandOff && fewUnits && lastL < nl - 1 && blocks(i) > blocks_of(lastL)) lastL = nl - 1;
                if (!direct.empty() && direct[i]) lastL = nl;
            }
            lvl[i] = lastL;
            ++atLevel[lastL];
        }
    }
    Lap lap;
    // A handful of PATH units (edlibAlign() with TASK_PATH on a 1 kb pair is one): the whole-wave ring they would start on
    // takes 0.48 us per dependent step with its store, and the walk a cell at a time.  First the 16-lane rings of 32-row
    // words (K = 448, or the whole matrix up to 16 words: DESIGN.md 4d): 0.16 us per step, a walk of ~T / 32 trips; a unit
    // beyond that band goes on to the levels below.
    if (paths != nullptr && fewUnits && !bandOff) {
        std::vector<UnitSpec> sel; std::vector<size_t> who;
        for (size_t i = 0; i < n; ++i) {
            UnitSpec u = units[i];
            const int nw = (u.qlen + 31) / 32;
            u.kinit = std::min(kcap, nw <= 16 ? std::max(u.qlen, u.tlen) : ring32_max_k(16));
            if (std::abs(u.qlen - u.tlen) > u.kinit || u.qlen < 128) continue;   // (no path within this band; tiny units: their 4-lane level takes ring32 anyway)
            sel.push_back(u); who.push_back(i);
        }
        if (!sel.empty() && ring32_fits(sel, 0, sel.size(), 16)) {
            SolveOut& so = soLevel_;
            if (solve(EDLIB_MODE_NW, false, true, sel, so, kRing32)) return 1;
            opsKeep_.insert(opsKeep_.end(), so.opsBufs.begin(), so.opsBufs.end());
            for (size_t q = 0; q < sel.size(); ++q) {
                const size_t i = who[q];
                if (so.score[q] > sel[q].kinit) continue;                         // beyond the band: the levels below
                score[i] = so.score[q];
                (*paths)[i].p = so.opsPtr[q]; (*paths)[i].len = so.opsLen[q];
                --atLevel[lvl[i]]; lvl[i] = -1;
            }
            lap("nw level: ring32");
        }
    }
    if (levelAllReady_) EDLIB_AMD_HIP(hipStreamWaitEvent(stream_, evLevelPeq_.e, 0));      // whatever follows: the side work is over before stream_'s next synchronisation
    // ---- the lane-per-pair level: every unit at its own threshold from the probe's rate (lanepair::unit_threshold; the
    // reference doubles k from 64, edlib.cpp:197-217: any threshold >= the distance gives the same answer).  A lane's band is
    // K + 1 diagonals at the start and narrows as the scores use the threshold up.  Units it leaves open (distance above the
    // threshold, foreign symbols, a band beyond the window) climb the rings.
    if (laneReady_) {
        EDLIB_AMD_HIP(hipStreamWaitEvent(stream_, evLanePack_.e, 0));
        // the window: what the threshold of a unit of the batch's longest length needs (its own length difference aside: a
        // unit whose threshold the window caps is scanned with the cap, and climbs the rings if that was too little)
        const int kLong = lanepair::unit_threshold(shape.minLenHi, shape.minLenHi, (float)rate, kcap, 0x3fffffff);
        int W = lanepair::window_for_k(kLong);
        if (W == 0 && kcap > lanepair::window_max_k(48)) W = 48;           // (what does not fit 48 words is left to the rings)
        if (rate > 0.0 && W != 0) {
            if (runLaneLevel(units, rate, kcap, W)) return 1;
            const int* got = reinterpret_cast<const int*>(h_laneScore_.p);
            size_t open = 0;
            for (size_t i = 0; i < n; ++i) {
                if (lvl[i] < 0) continue;
                const int l = lvl[i], g = got[i];
                if (g < lanepair::kAboveFinal) { score[i] = g; --atLevel[l]; lvl[i] = -1; }          // exact
                else if (g == lanepair::kAboveFinal) { score[i] = kInf; --atLevel[l]; lvl[i] = -1; } // > k: final
                else {
                    if (g == lanepair::kAboveOpen) {                                                  // the rings, above its threshold
                        const int Kl = lanepair::unit_threshold(units[i].qlen, units[i].tlen, (float)rate, kcap, lanepair::window_max_k(W));
                        int up = l;
                        while (up < nl && cap_of(up) <= Kl && blocks(i) > blocks_of(up)) ++up;
                        if (up != l) { --atLevel[l]; lvl[i] = up; ++atLevel[up]; }
                    }
                    ++open;
                }
            }
            if (getenv("EDLIB_AMD_DEBUG")) fprintf(stderr, "[edlib_amd] lane level: %d words, threshold of the longest unit %d, %zu of %zu units open\n", W, kLong, open, n);
            lap("nw lane level: scores");
        }
    }
    for (int l = 0; l <= nl; ++l) {
        if (atLevel[l] == 0) continue;
        if (l == nl && wideLevel) break;
        if (levelAllReady_ && l < nl && atLevel[l] == n) {                // ---- the level takes every unit
            if (runLevelAll(units, ringOf[l], ringH[l], blocks_of(l), cap_of(l), kcap, (int)maxBlocks)) return 1;
            const int* got = reinterpret_cast<const int*>(h_levelScore_.p);
            atLevel[l] = 0;
            const int capL = std::min(kcap, cap_of(l));
            const bool banded = shape.minBlocks > (size_t)blocks_of(l);           // every unit inside the ring's band limit: one threshold
            for (size_t i = 0; i < n; ++i) {
                const int kin = banded ? capL : std::min(kcap, blocks(i) <= blocks_of(l) ? std::max(units[i].qlen, units[i].tlen) : cap_of(l));
                if (got[i] <= kin) score[i] = got[i];                             // exact
                else if (kin >= kcap) score[i] = kInf;                             // > k: final
                else { lvl[i] = l + 1; ++atLevel[l + 1]; }                         // next level
            }
            lap("nw level (every unit): scores");
            continue;
        }
        std::vector<UnitSpec>& sel = selScratch_; std::vector<size_t>& who = whoScratch_;
        sel.clear(); who.clear();
        sel.reserve(atLevel[l]); who.reserve(atLevel[l]);
        for (size_t i = 0; i < n; ++i) {
            if (lvl[i] != l) continue;
            UnitSpec u = units[i];
            if (l < nl) u.kinit = std::min(kcap, blocks(i) <= blocks_of(l) ? std::max(u.qlen, u.tlen) : cap_of(l));
            sel.push_back(u); who.push_back(i);
        }
        if (sel.empty()) continue;
        lap("nw level: select");
        SolveOut& so = soLevel_;
        const bool store = paths != nullptr && (l == nl || ringH[l] == 1);
        // A level that only reruns what the one before left open is a handful of waves: its launch takes the T + blocks
        // DEPENDENT steps of one wave however few units there are (config 4: 7 % of the batch, 2.6 ms of a 23 ms step).  The
        // two halves of a target are independent (edlib.cpp:1246-1260): forward over the left half, reverse over the right
        // half, min over the rows of L[i] + R[i+1] -- half the steps, while twice the waves still fit the chip at once.
        bool halves = !store && l < nl && ringH[l] == 1 && sel.size() < n && 2 * ((sel.size() + 64 / ringOf[l] - 1) / (64 / ringOf[l])) <= 8192;
        for (size_t q = 0; halves && q < sel.size(); ++q)
            halves = sel[q].tlen >= 2048 && sel[q].qstep == 1 && sel[q].tstep == 1 && sel[q].kinit <= cap_of(l) && blocks(who[q]) > blocks_of(l);
        if (halves) {
            std::vector<int> sp;
            if (solveWideSplit(sel, sp, ringOf[l])) return 1;
            so.score.resize(sel.size());
            for (size_t q = 0; q < sel.size(); ++q) so.score[q] = sp[4 * q];
        } else
        if (solve(EDLIB_MODE_NW, false, store, sel, so, l < nl ? ringOf[l] : 0, l < nl ? ringH[l] : 1)) return 1;
        lap("nw level: solve");
        if (store) opsKeep_.insert(opsKeep_.end(), so.opsBufs.begin(), so.opsBufs.end());
        for (size_t q = 0; q < sel.size(); ++q) {
            const size_t i = who[q];
            if (store && (l == nl || so.score[q] <= sel[q].kinit)) { (*paths)[i].p = so.opsPtr[q]; (*paths)[i].len = so.opsLen[q]; }
            if (l == nl || so.score[q] <= sel[q].kinit) score[i] = so.score[q];         // exact
            else if (sel[q].kinit >= kcap) score[i] = kInf;                              // > k: final
            else { lvl[i] = l + 1; ++atLevel[l + 1]; }                                   // next level
        }
        lap("nw level: scores");
    }
    // ---- beyond the rings: Ukkonen's band of ANY width on many waves (wide_kernels.hip).  The reference keeps doubling k
    // (edlib.cpp:197-217); a pass here costs about T dependent steps whatever its K, so the first K is generous (1.5 x the
    // estimate) and a failed pass doubles it.  K = max(m, T) is the whole matrix and always exact.
    if (wideLevel && atLevel[nl] > 0) {
        std::vector<size_t> rest;
        std::vector<long long> kcur(n, 0);
        for (size_t i = 0; i < n; ++i)
            if (lvl[i] == nl) {
                rest.push_back(i);
                const double est = mean_of(i);
                const bool dir = !direct.empty() && direct[i];
                kcur[i] = std::max<long long>(dir ? 1024 : 2LL * (ring_max_k(64) + 128), (long long)(1.5 * est + 4.0 * std::sqrt(est) + 64.0));
                if (dir && whole_ok(i)) kcur[i] = std::max(units[i].qlen, units[i].tlen);
                if (const char* e = getenv("EDLIB_AMD_WIDE_K0")) { if (atoi(e) > 0) kcur[i] = atoi(e); }     // (tests: the ladder from a small K)
            }
        // long units: two half scans that meet in the middle (solveWideSplit: half the dependent steps); a unit of one
        // target column has no two halves
        const int splitMin = direct.empty() ? 16384 : 4096;
        while (!rest.empty()) {
            std::vector<UnitSpec>& sel = selScratch_;
            sel.clear();
            std::vector<UnitSpec> halves; std::vector<size_t> whoWhole, whoHalves;
            for (size_t i : rest) {
                UnitSpec u = units[i];
                u.kinit = (int)std::min<long long>(std::min<long long>(kcap, kcur[i]), std::max(u.qlen, u.tlen));
                if (splitMin > 0 && std::min(u.qlen, u.tlen) >= splitMin && u.tlen >= 2) { halves.push_back(u); whoHalves.push_back(i); }
                else { sel.push_back(u); whoWhole.push_back(i); }
            }
            std::vector<size_t> again;
            auto settle = [&](size_t i, const UnitSpec& u, int got) -> int {
                if (got >= 0 && got <= u.kinit) score[i] = got;                                   // exact
                else if (u.kinit >= kcap) score[i] = kInf;                                          // > k: final
                else if (u.kinit >= std::max(u.qlen, u.tlen)) { set_error("wide band: no score inside the whole matrix"); return 1; }
                else { kcur[i] = 2LL * u.kinit; again.push_back(i); }
                return 0;
            };
            if (!sel.empty()) {
                SolveOut& so = soLevel_;
                if (solve(EDLIB_MODE_NW, false, false, sel, so, kWide)) return 1;
                for (size_t q = 0; q < sel.size(); ++q) if (settle(whoWhole[q], sel[q], so.score[q])) return 1;
            }
            if (!halves.empty()) {
                std::vector<int> sp;
                if (solveWideSplit(halves, sp)) return 1;
                for (size_t q = 0; q < halves.size(); ++q) {
                    const UnitSpec& u = halves[q];
                    if (sp[4 * q] <= u.kinit && u.qstep == 1 && u.tstep == 1)
                        knownSplits_.push_back(KnownSplit{u.qoff, u.qlen, u.toff, u.tlen, sp[4 * q], sp[4 * q + 1], sp[4 * q + 2], sp[4 * q + 3]});
                    if (settle(whoHalves[q], u, sp[4 * q])) return 1;
                }
            }
            lap("nw wide level");
            rest.swap(again);
        }
    }
    return 0;
}


}  // namespace edlib_amd
