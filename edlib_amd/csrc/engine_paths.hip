// engine_paths.hip -- alignment paths of NW jobs of any size (reference obtainAlignment / obtainAlignmentHirschberg,
// edlib.cpp:1161-1213, 1231-1396): the 1 MiB rule, Hirschberg levels flattened over the batch, the distance of a long unit
// as two half scans that meet in the middle, leaves through store + traceback.
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace edlib_amd {

// ------------------------------------------------------------- Hirschberg

bool needs_hirschberg(int m, int T) {
    const long long nb = (m + 63) / 64;
    return (2LL * 8 + 4) * nb * T + 8LL * T >= 1024 * 1024;              // edlib.cpp:1188-1190
}

// One level of the divide step for a set of pieces: forward scan of (query, left half) and reverse
// scan of (reversed query, reversed right half), both NW and dumped at their last column
// (edlib.cpp:1246-1260), then the split search on the device (edlib.cpp:1314-1353).
int Batch::hirschbergLevel(const std::vector<PathPiece>& big, std::vector<int>& splitRow,
                           std::vector<int>& leftScore, std::vector<int>& rightScore)
{
    const size_t np = big.size();
    // Each piece scans inside the band of the WHOLE piece with k = its distance, stopped at the half's last
    // column -- exactly the reference's two calls, edlib.cpp:1252-1260 -- on the smallest lane ring that holds
    // that band (or all its blocks); pieces no ring holds take the unbanded strips.  Both dump their last column.
    static const int rings[kNumRings + 1] = {4, 8, 16, 21, 32, 64, 0};
    // (packing only pays with enough pieces to fill the chip: a handful of long pieces runs faster one per wave)
    const bool packed = np >= 256;
    auto ring_of = [&](const PathPiece& pc) {
        const bool off = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
        if (off) return kNumRings;
        // (a few long pieces are bound by dependent steps: 0.074 us on the wide kernel's waves against 0.12 on a ring's)
        if (!packed) return (pc.m > 64 * 64 && pc.score <= kMaxBandK && pc.T < 4096) ? kNumRings - 1 : kNumRings;
        for (int g = 0; g < kNumRings; ++g) if (pc.score <= ring_max_k(rings[g])) return g;
        return (pc.m + 63) / 64 <= 64 ? kNumRings - 1 : kNumRings;
    };
    std::vector<size_t> order; order.reserve(np);
    size_t groupCount[kNumRings + 1] = {0};
    std::vector<int> groupOf(np);
    for (size_t p = 0; p < np; ++p) { groupOf[p] = ring_of(big[p]); ++groupCount[groupOf[p]]; }
    for (int g = 0; g <= kNumRings; ++g) for (size_t p = 0; p < np; ++p) if (groupOf[p] == g) order.push_back(p);
    std::vector<PairDesc> descs(2 * np);
    std::vector<int> best(np);
    // what no ring holds: the band on many waves (wide_kernels.hip)
    const bool wideOff = false;
    long long peqWords = 0, auxInts = 0, colBlocks = 0;
    for (size_t q = 0; q < np; ++q) {
        const PathPiece& pc = big[order[q]];
        const int ring = rings[groupOf[order[q]]];
        const bool wide = ring == 0 && !wideOff;
        const bool banded = ring != 0 || wide;
        const int lw = pc.T / 2, rw = pc.T - lw;                         // :1247-1248
        const long long nb = (pc.m + 63) / 64;
        best[q] = pc.score;
        for (int side = 0; side < 2; ++side) {
            PairDesc& d = descs[2 * q + side];
            d.qlen = pc.m; d.kinit = banded ? pc.score : 0; d.posCap = 0; d.posOff = 0; d.storeOff = 0;
            d.bandT = banded ? pc.T : 0; d.ring = 0; d.skip = 0;
            if (side == 0) { d.qoff = pc.qoff; d.qstep = 1; d.toff = pc.toff; d.tstep = 1; d.tlen = lw; }
            else { d.qoff = pc.qoff + pc.m - 1; d.qstep = -1; d.toff = pc.toff + pc.T - 1; d.tstep = -1; d.tlen = rw; }
            d.peqOff = peqWords; peqWords += nb * tab_.sigmaT;
            d.auxOff = auxInts; if (nb > 64 && !banded) auxInts += d.tlen;
            d.colOff = colBlocks; colBlocks += nb;
            if (!banded) stats.word_steps += 2 * nb * (long long)d.tlen;
            else if (wide) stats.word_steps += wide_word_steps(0, d.qlen, d.tlen, d.bandT, d.kinit);
        }
    }
    const size_t firstWide = np - groupCount[kNumRings];               // (the groups are laid out in ring order, this one last)
    WidePlan wplan;
    const bool anyWide = groupCount[kNumRings] > 0 && !wideOff;
    if (anyWide && planWide(0, descs.data() + 2 * firstWide, 2 * groupCount[kNumRings], wplan)) return 1;
    const size_t n = descs.size();
    DevBuf<unsigned long long> colP, colM; DevBuf<int> colS, d_best, d_out;
    EDLIB_AMD_HIP(colP.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colM.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colS.alloc((size_t)colBlocks));
    EDLIB_AMD_HIP(d_best.alloc(np)); EDLIB_AMD_HIP(d_out.alloc(3 * np));
    // blocks outside the band at the stop column: P = M = 0 and a score no sum can reach
    EDLIB_AMD_HIP(hipMemsetAsync(colP.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colM.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colS.p, 0x3f, (size_t)colBlocks * 4, stream_));
    EDLIB_AMD_HIP(d_descs_.ensure(n)); EDLIB_AMD_HIP(d_peq64_.ensure((size_t)peqWords)); EDLIB_AMD_HIP(d_aux_.ensure((size_t)auxInts));
    EDLIB_AMD_HIP(d_out3_.ensure(3 * n));
    d_outScore_.alias(d_out3_.p, n); d_outCount_.alias(d_out3_.p + n, n); d_outLast_.alias(d_out3_.p + 2 * n, n);
    EDLIB_AMD_HIP(d_posPool_.ensure(1));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_descs_.p, descs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_best.p, best.data(), np * sizeof(int), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_descs_.p, (int)n, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT,
                                         d_peq64_.p, stream_));
    PairScanArgs a{};
    a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = d_aux_.p;
    a.posPool = d_posPool_.p;
    {
        long long nbMax = 0;
        for (const PathPiece& pc : big) nbMax = std::max<long long>(nbMax, (pc.m + 63) / 64);
        a.peqRowStride = peq_row_stride(nbMax);
        a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
    }
    a.colP = colP.p; a.colM = colM.p; a.colS = colS.p;
    a.wordSteps = ringStepsCounter();
    size_t first = 0;
    for (int g = 0; g <= kNumRings; ++g) {
        if (!groupCount[g]) continue;
        a.descs = d_descs_.p + 2 * first; a.numUnits = (int)(2 * groupCount[g]);
        a.outScore = d_outScore_.p + 2 * first; a.outCount = d_outCount_.p + 2 * first; a.outLast = d_outLast_.p + 2 * first;
        scanTimerStart();
        if (rings[g]) EDLIB_AMD_HIP(launch_scan_pairs_ring(rings[g], 0, false, a, stream_));
        else if (anyWide) { if (launchWide(0, a, descs.data() + 2 * first, 2 * groupCount[g], wplan)) return 1; }
        else EDLIB_AMD_HIP(launch_scan_pairs(EDLIB_MODE_NW, false, a, stream_));
        scanTimerStop();
        first += groupCount[g];
    }
    SplitArgs sa{};
    sa.descs = d_descs_.p; sa.numPieces = (int)np; sa.best = d_best.p;
    sa.colP = colP.p; sa.colM = colM.p; sa.colS = colS.p; sa.out = d_out.p;
    EDLIB_AMD_HIP(launch_hirschberg_split(sa, stream_));
    std::vector<int> out(3 * np);
    EDLIB_AMD_HIP(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    if (anyWide) {
        const int w = checkWide();
        if (w == 2) return hirschbergLevel(big, splitRow, leftScore, rightScore);
        if (w) return 1;
    }
    splitRow.resize(np); leftScore.resize(np); rightScore.resize(np);
    for (size_t q = 0; q < np; ++q) {
        const size_t p = order[q];
        splitRow[p] = out[3 * q]; leftScore[p] = out[3 * q + 1]; rightScore[p] = out[3 * q + 2];
    }
    return 0;
}

// A scan of T columns is T dependent steps however many waves share its band; the two halves of the target are independent
// of each other.  So the distance of a long unit is found like the first Hirschberg level finds its split
// (edlib.cpp:1246-1260, 1314-1353): forward scan of (query, left half) and reverse scan of (reversed query, reversed right
// half), both inside the band of the whole problem and dumped at their last column, then min over the query rows of
// L[i] + R[i+1].  Half the dependent steps; exact iff the minimum is within the threshold (cells outside the band are
// upper bounds).
// ring > 0: the two halves of every unit on lane rings of that many lanes instead (band of UnitSpec::kinit <= ring_max_k(ring)):
// what a HANDFUL of ring units -- the rerun of the units a level left open -- takes, whose launch is bound by the
// T + blocks dependent steps of its waves, not by work.
int Batch::solveWideSplit(const std::vector<UnitSpec>& units, std::vector<int>& out, int ring)
{
    const size_t np = units.size();
    out.assign(4 * np, 0);
    if (np == 0) return 0;
    std::vector<PairDesc> descs(2 * np);
    long long peqWords = 0, colBlocks = 0;
    int maxRows = 1;
    for (size_t q = 0; q < np; ++q) {
        const UnitSpec& u = units[q];
        const int lw = u.tlen / 2, rw = u.tlen - lw;
        const long long nb = (u.qlen + 63) / 64;
        maxRows = std::max(maxRows, u.qlen);
        for (int side = 0; side < 2; ++side) {
            PairDesc& d = descs[2 * q + side];
            d.qlen = u.qlen; d.kinit = u.kinit; d.posCap = 0; d.posOff = 0; d.storeOff = 0; d.bandT = u.tlen; d.ring = 0; d.skip = 0;
            if (side == 0) { d.qoff = u.qoff; d.qstep = u.qstep; d.toff = u.toff; d.tstep = u.tstep; d.tlen = lw; }
            else {
                d.qoff = u.qoff + (long long)(u.qlen - 1) * u.qstep; d.qstep = -u.qstep;
                d.toff = u.toff + (long long)(u.tlen - 1) * u.tstep; d.tstep = -u.tstep; d.tlen = rw;
            }
            d.peqOff = peqWords; peqWords += nb * tab_.sigmaT;
            d.auxOff = 0;
            d.colOff = colBlocks; colBlocks += nb;
            if (ring <= 0) stats.word_steps += wide_word_steps(0, d.qlen, d.tlen, d.bandT, d.kinit);
        }
    }
    WidePlan wplan;
    if (ring <= 0 && planWide(0, descs.data(), descs.size(), wplan)) return 1;
    const size_t n = descs.size();
    DevBuf<unsigned long long> colP, colM, packed; DevBuf<int> colS, d_out;
    EDLIB_AMD_HIP(colP.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colM.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colS.alloc((size_t)colBlocks));
    EDLIB_AMD_HIP(packed.alloc(np)); EDLIB_AMD_HIP(d_out.alloc(4 * np));
    // blocks outside the band at the stop column: P = M = 0 and a score no sum can reach
    EDLIB_AMD_HIP(hipMemsetAsync(colP.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colM.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colS.p, 0x3f, (size_t)colBlocks * 4, stream_));
    EDLIB_AMD_HIP(d_descs_.ensure(n)); EDLIB_AMD_HIP(d_peq64_.ensure((size_t)peqWords));
    EDLIB_AMD_HIP(d_out3_.ensure(3 * n));
    d_outScore_.alias(d_out3_.p, n); d_outCount_.alias(d_out3_.p + n, n); d_outLast_.alias(d_out3_.p + 2 * n, n);
    EDLIB_AMD_HIP(d_posPool_.ensure(1));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_descs_.p, descs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_descs_.p, (int)n, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT, d_peq64_.p, stream_));
    PairScanArgs a{};
    a.descs = d_descs_.p; a.numUnits = (int)n; a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = d_aux_.p; a.posPool = d_posPool_.p;
    a.outScore = d_outScore_.p; a.outCount = d_outCount_.p; a.outLast = d_outLast_.p;
    a.colP = colP.p; a.colM = colM.p; a.colS = colS.p;
    scanTimerStart();
    if (ring > 0) {
        long long nbMax = 0;
        for (const UnitSpec& u : units) nbMax = std::max<long long>(nbMax, (u.qlen + 63) / 64);
        a.peqRowStride = peq_row_stride(nbMax);
        a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
        a.wordSteps = ringStepsCounter();
        EDLIB_AMD_HIP(launch_scan_pairs_ring(ring, 0, false, a, stream_));
    } else if (launchWide(0, a, descs.data(), n, wplan)) return 1;
    scanTimerStop();
    SplitArgs sa{};
    sa.descs = d_descs_.p; sa.numPieces = (int)np; sa.best = nullptr;
    sa.colP = colP.p; sa.colM = colM.p; sa.colS = colS.p; sa.out = d_out.p;
    EDLIB_AMD_HIP(launch_split_min(sa, packed.p, maxRows, stream_));
    EDLIB_AMD_HIP(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    if (ring <= 0) {
        const int w = checkWide();
        if (w == 2) return solveWideSplit(units, out);
        if (w) return 1;
    }
    return 0;
}

// Alignment paths of NW jobs of any size (reference obtainAlignment, edlib.cpp:1161-1213): pieces at
// or above the 1 MiB column-store estimate are halved Hirschberg-style, level by level across the
// whole batch, until every piece fits the traceback branch; the pieces' op strings concatenate.
int Batch::solvePaths(const std::vector<PathPiece>& jobs, std::vector<OpsOut>& opsOut, std::vector<int>& status)
{
    const size_t nj = jobs.size();
    Lap lap;
    opsOut.clear(); opsOut.resize(nj); status.assign(nj, EDLIB_STATUS_OK);
    // only jobs at or above the 1 MiB rule are ever split: the others stay a single implicit piece
    std::vector<std::vector<PathPiece>> pieces(nj);
    std::vector<size_t> bigJobs;
    for (size_t j = 0; j < nj; ++j)
        if (needs_hirschberg(jobs[j].m, jobs[j].T)) { pieces[j].push_back(jobs[j]); bigJobs.push_back(j); }
    auto npieces = [&](size_t j) { return pieces[j].empty() ? (size_t)1 : pieces[j].size(); };
    auto piece = [&](size_t j, size_t i) -> const PathPiece& { return pieces[j].empty() ? jobs[j] : pieces[j][i]; };
    for (int level = 0; level < 64 && !bigJobs.empty(); ++level) {
        std::vector<PathPiece> big; std::vector<std::pair<size_t, size_t>> where;
        for (size_t j : bigJobs) {
            if (status[j] != EDLIB_STATUS_OK) continue;
            for (size_t i = 0; i < pieces[j].size(); ++i) {
                const PathPiece& pc = pieces[j][i];
                if (pc.m > 0 && pc.T > 0 && needs_hirschberg(pc.m, pc.T)) {
                    if (pc.T < 2) { status[j] = EDLIB_STATUS_ERROR; break; }   // the reference has no answer here either
                    big.push_back(pc); where.push_back({j, i});
                }
            }
        }
        if (big.empty()) break;
        std::vector<int> row, ls, rs;
        // pieces whose two half scans already ran for the distance (solveWideSplit) bring their split along
        std::vector<PathPiece> todo; std::vector<size_t> todoAt;
        row.assign(big.size(), -2); ls.assign(big.size(), 0); rs.assign(big.size(), 0);
        for (size_t b = 0; b < big.size(); ++b) {
            const KnownSplit* ks = nullptr;
            if (level == 0)
                for (const KnownSplit& k : knownSplits_)
                    if (k.qoff == big[b].qoff && k.m == big[b].m && k.toff == big[b].toff && k.T == big[b].T && k.score == big[b].score) { ks = &k; break; }
            if (ks) { row[b] = ks->row; ls[b] = ks->left; rs[b] = ks->right; }
            else { todo.push_back(big[b]); todoAt.push_back(b); }
        }
        if (!todo.empty()) {
            std::vector<int> r2, l2, s2;
            if (hirschbergLevel(todo, r2, l2, s2)) return 1;
            for (size_t q = 0; q < todo.size(); ++q) { row[todoAt[q]] = r2[q]; ls[todoAt[q]] = l2[q]; rs[todoAt[q]] = s2[q]; }
        }
        // replace pieces back to front so the recorded indices stay valid
        for (size_t b = big.size(); b-- > 0;) {
            const size_t j = where[b].first, i = where[b].second;
            if (status[j] != EDLIB_STATUS_OK) continue;
            if (row[b] == -2) { status[j] = EDLIB_STATUS_ERROR; continue; }          // edlib.cpp:1358-1362
            const PathPiece pc = pieces[j][i];
            const int lw = pc.T / 2, ulH = row[b] + 1;                                // :1367-1370
            const PathPiece ul{pc.qoff, ulH, pc.toff, lw, ls[b]};
            const PathPiece lr{pc.qoff + ulH, pc.m - ulH, pc.toff + lw, pc.T - lw, rs[b]};
            pieces[j][i] = ul;
            pieces[j].insert(pieces[j].begin() + i + 1, lr);
        }
    }
    // leaves: trivial pieces on the host (edlib.cpp:1168-1175), the rest through store + traceback
    std::vector<UnitSpec> units;
    units.reserve(nj);
    for (size_t j = 0; j < nj; ++j) {
        if (status[j] != EDLIB_STATUS_OK) continue;
        for (size_t i = 0; i < npieces(j); ++i) {
            const PathPiece& pc = piece(j, i);
            // kinit = the piece's distance: the storing scan runs inside exactly that band (the reference's
            // second call with k = bestScore, edlib.cpp:1196-1199)
            if (pc.m > 0 && pc.T > 0) units.push_back(UnitSpec{pc.qoff, pc.m, 1, pc.toff, pc.T, 1, pc.score});
        }
    }
    lap("paths: levels+units");
    // smallest ring that holds the band (or all blocks) of each leaf; strips when none does
    std::vector<const uint8_t*> leafPtr(units.size(), nullptr); std::vector<int> leafLen(units.size(), 0);
    {
        const bool bandOff = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
        static const int rings[kNumRings + 1] = {4, 8, 16, 21, 32, 64, 0};
        std::vector<int> ringOfUnit(units.size(), 0);
        for (size_t u = 0; u < units.size(); ++u) {
            const int nb = (units[u].qlen + 63) / 64;
            for (int g = 0; g < kNumRings && !bandOff; ++g)
                if (nb <= rings[g] || units[u].kinit <= ring_max_k(rings[g])) { ringOfUnit[u] = rings[g]; break; }
        }
        size_t perRing[kNumRings + 1] = {0};
        for (size_t u = 0; u < units.size(); ++u) for (int g = 0; g <= kNumRings; ++g) if (ringOfUnit[u] == rings[g]) ++perRing[g];
        for (int g = 0; g <= kNumRings; ++g) {
            if (!perRing[g]) continue;
            SolveOut so;
            if (perRing[g] == units.size()) {                       // the usual case: one kind of leaf
                if (solve(EDLIB_MODE_NW, false, true, units, so, rings[g])) return 1;
                leafPtr.swap(so.opsPtr); leafLen.swap(so.opsLen);
            } else {
                std::vector<UnitSpec> sel; std::vector<size_t> who;
                sel.reserve(perRing[g]); who.reserve(perRing[g]);
                for (size_t u = 0; u < units.size(); ++u) if (ringOfUnit[u] == rings[g]) { sel.push_back(units[u]); who.push_back(u); }
                if (solve(EDLIB_MODE_NW, false, true, sel, so, rings[g])) return 1;
                for (size_t q = 0; q < sel.size(); ++q) { leafPtr[who[q]] = so.opsPtr[q]; leafLen[who[q]] = so.opsLen[q]; }
            }
            opsKeep_.insert(opsKeep_.end(), so.opsBufs.begin(), so.opsBufs.end());
        }
    }
    lap("paths: solve");
    // a job that was never split is its single leaf: hand out the view; split jobs concatenate their pieces
    std::vector<size_t> firstLeaf(nj + 1, 0);                 // leaves are listed job by job, piece by piece
    {
        size_t u = 0;
        for (size_t j = 0; j < nj; ++j) {
            firstLeaf[j] = u;
            if (status[j] != EDLIB_STATUS_OK) continue;
            for (size_t i = 0; i < npieces(j); ++i) if (piece(j, i).m > 0 && piece(j, i).T > 0) ++u;
        }
        firstLeaf[nj] = u;
    }
    for (size_t j = 0; j < nj; ++j) {
        if (status[j] != EDLIB_STATUS_OK) continue;
        OpsOut& o = opsOut[j];
        size_t u = firstLeaf[j];
        if (npieces(j) == 1 && piece(j, 0).m > 0 && piece(j, 0).T > 0) {
            o.p = leafPtr[u]; o.len = leafLen[u];
            continue;
        }
        for (size_t i = 0; i < npieces(j); ++i) {
            const PathPiece& pc = piece(j, i);
            if (pc.m == 0) o.own.insert(o.own.end(), (size_t)pc.T, (uint8_t)EDLIB_EDOP_DELETE);
            else if (pc.T == 0) o.own.insert(o.own.end(), (size_t)pc.m, (uint8_t)EDLIB_EDOP_INSERT);
            else { o.own.insert(o.own.end(), leafPtr[u], leafPtr[u] + leafLen[u]); ++u; }
        }
        o.p = o.own.data(); o.len = (int)o.own.size();
    }
    lap("paths: assemble");
    return 0;
}


}  // namespace edlib_amd
