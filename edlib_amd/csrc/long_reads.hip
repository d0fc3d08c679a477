// long_reads.hip -- long HW queries against the shared target (385 rows and more by default: engine.hip, kFilterFromWords;
// 257..384 rows run on kernel A's 12-word group, which is as fast or faster there).
//
// What the reference does for such a query (edlib.cpp:197-217, 550-704): myersCalcEditDistanceSemiGlobal over the whole
// target inside Ukkonen's band for k = 64, 128, ... until the distance fits.  The band of a semi-global scan is anchored
// at the top: away from a match it holds the rows whose score can still be <= k (roughly the first 2k rows), and only
// around a match it reaches the bottom row.  Kernel A's lane-per-read shape is the efficient one for a short top-anchored
// band but holds at most 256 rows per lane; kernel W's lane-per-block shape holds any query but wastes its
// lanes on a short band.  So the scan is split where the band itself splits:
//
//   1. FILTER (kernel A, `ReadScanArgs::filter`): the query is cut into p parts and the first <= 256 rows of every part
//      ("piece") are scanned against the whole target with the fixed threshold kp = floor(k / p) <= 56.  An alignment of the
//      whole query with cost <= k spends at most kp on one of the p parts (pigeonhole), hence at most kp on that
//      part's piece, and the piece's HW score at the column where the alignment leaves the piece's last row is <= kp:
//      every end column j of the query with D[m][j] <= k lies within k of (candidate column) + (rows below the piece).
//      The scan lists the 16-column blocks that hold candidate columns.  This is the top of the band: p lanes of at
//      most 8 words each instead of one column of ceil(m / 64) blocks.
//   2. VERIFY (kernel W, HW units with `skip`): the candidates of a query give windows of end columns; each merged
//      window is scanned with the WHOLE query from m + k columns before it (a cell <= k has its alignment start within
//      m + k columns, so restarted scores are exact where they are <= k and upper bounds elsewhere), recording only the
//      window's own columns.  This is the band around a match: full height, a few hundred columns.
//   3. the smallest score over the windows, if <= k, is the distance and its columns are the end locations (every
//      column with D[m][j] <= k is inside a window and exact there); else k doubles -- the reference's k-doubling.
//
// Thresholds only steer work: whatever level resolves a query, its result is the function of the full DP matrix that
// SURVEY.md §8a spells out.  Queries the filter cannot narrow (threshold above a quarter of the piece, candidate lists
// that overflow: low-complexity sequence) are handed back and run on kernel W over the whole target as before.
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace edlib_amd {

namespace {

const int kPieceRows = 32 * kMaxReadWords;       // 256
const int kCandCap = 16;                          // candidate blocks kept per (piece, target segment)
const int kMaxWindows = 512;                      // merged windows per query before it is handed back

struct LevelPlan { int p, partLen, rows, kp; bool ok; };

// parts and piece threshold of a query of m rows at level k: the fewest parts whose threshold stays <= kpMax (a
// piece costs its band height, which grows with kp by about a word per 8; parts cost a lane each)
LevelPlan plan_level(int m, int k, int kpMax)
{
    LevelPlan L{};
    L.p = k / (kpMax + 1) + 1;                    // smallest p with floor(k / p) <= kpMax
    L.partLen = m / L.p;
    L.rows = std::min(kPieceRows, L.partLen);
    L.kp = k / L.p;
    // a piece must stay specific: at most a quarter of its rows in errors, at least 48 rows
    L.ok = L.rows >= 48 && 4 * L.kp <= L.rows;
    return L;
}

}  // namespace

// One scan of a set of pieces on the reads-per-lane kernel.  filter: fixed thresholds, `cand` receives (piece, block)
// pairs and overflow[piece] = 1 where a list did not fit; else the banded scan with k tightening, best[piece] = the
// piece's HW distance (-1 if above its threshold).
int Batch::scanPieces(const std::vector<Piece>& pieces, bool filter, std::vector<std::pair<int, int>>* cand,
                      std::vector<uint8_t>* overflow, std::vector<int>* best)
{
    const size_t np = pieces.size();
    if (cand) cand->clear();
    if (overflow) overflow->assign(np, 0);
    if (best) best->assign(np, -1);
    if (!np) return 0;
    const int T = tlen(0);
    const int kNoCap = 0x3fffffff;
    // piece bounds in the form the Peq builder reads query offsets: slot -> entry 2j, rows [pb[2j], pb[2j + 1])
    PinBuf pbPin;
    EDLIB_AMD_HIP(pbPin.alloc(2 * np * sizeof(long long)));
    long long* pb = reinterpret_cast<long long*>(pbPin.p);
    std::vector<std::vector<int>> byWords(kMaxReadWords + 1);
    for (size_t j = 0; j < np; ++j) {
        pb[2 * j] = pieces[j].off; pb[2 * j + 1] = pieces[j].off + pieces[j].len;
        byWords[(pieces[j].len + 31) / 32].push_back((int)j);
    }
    DevBuf<long long> d_pb;
    EDLIB_AMD_HIP(d_pb.alloc(2 * np));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_pb.p, pb, 2 * np * sizeof(long long), hipMemcpyHostToDevice, stream_));
    for (int w = 1; w <= kMaxReadWords; ++w) {
        const std::vector<int>& who = byWords[w];
        if (who.empty()) continue;
        ReadGroup g;
        g.nwords = w;
        g.nslots = roundup((int)who.size(), 64);
        const size_t ns = (size_t)g.nslots;
        PinBuf slotPin;                                        // perm | thr
        EDLIB_AMD_HIP(slotPin.alloc(2 * ns * sizeof(int)));
        int* perm = reinterpret_cast<int*>(slotPin.p); int* thr = perm + ns;
        for (size_t s = 0; s < ns; ++s) {
            const bool real = s < who.size();
            perm[s] = real ? 2 * who[s] : -1;
            thr[s] = real ? pieces[who[s]].thr : -1;           // padding lanes: nothing ever scores <= -1
        }
        DevBuf<int> d_thr, d_best, d_total, d_pos, d_flags;
        EDLIB_AMD_HIP(g.d_perm.alloc(ns)); EDLIB_AMD_HIP(d_thr.alloc(ns));
        EDLIB_AMD_HIP(hipMemcpyAsync(g.d_perm.p, perm, ns * sizeof(int), hipMemcpyHostToDevice, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_thr.p, thr, ns * sizeof(int), hipMemcpyHostToDevice, stream_));
        EDLIB_AMD_HIP(g.d_qlen.alloc(ns)); EDLIB_AMD_HIP(g.d_kinit.alloc(ns)); EDLIB_AMD_HIP(g.d_alphaExtra.alloc(ns));
        EDLIB_AMD_HIP(g.d_peq.alloc(ns * (size_t)syms_ * w));
        g.warm = 2 * 32 * w - 1;
        plan_segments(g.nslots, T, EDLIB_MODE_HW, g.warm, 131072, g.numSegments, g.segLen, g.warm);
        const size_t S = (size_t)g.numSegments, cap = filter ? kCandCap : 8;
        EDLIB_AMD_HIP(g.d_segBest.alloc(ns * S)); EDLIB_AMD_HIP(g.d_segCnt.alloc(ns * S)); EDLIB_AMD_HIP(g.d_segPos.alloc(ns * S * cap));
        EDLIB_AMD_HIP(launch_build_peq_reads(w, syms_, d_qpool_.p, d_pb.p, g.d_perm.p, g.nslots, d_eqtbl_.p, d_presence_.p,
                                             -1, g.d_peq.p, g.d_qlen.p, g.d_kinit.p, g.d_alphaExtra.p, stream_));
        filterScan_ = filter;
        const int rc = scanGroup(g, EDLIB_MODE_HW, nullptr, g.nslots, kNoCap, d_thr.p, g.numSegments, g.segLen, g.warm,
                                 g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, (int)cap, nullptr, nullptr);
        filterScan_ = false;
        if (rc) return 1;
        if (filter) {
            // gather the (slot, block) candidates; the list is small (a handful per piece that hits at all)
            DevBuf<int> d_out, d_ctl;                          // d_ctl: [0] = counter, [1 .. ns] = overflow flags
            size_t maxOut = std::max<size_t>(4096, 4 * ns);
            EDLIB_AMD_HIP(d_ctl.alloc(ns + 1));
            PinBuf ctlPin; EDLIB_AMD_HIP(ctlPin.alloc((ns + 1) * sizeof(int)));
            int* ctl = reinterpret_cast<int*>(ctlPin.p);
            for (int attempt = 0; attempt < 2; ++attempt) {
                EDLIB_AMD_HIP(d_out.alloc(2 * maxOut));
                EDLIB_AMD_HIP(hipMemsetAsync(d_ctl.p, 0, (ns + 1) * sizeof(int), stream_));
                EDLIB_AMD_HIP(launch_collect_candidates(g.d_segCnt.p, g.d_segPos.p, g.numSegments, (int)cap, g.nslots,
                                                        d_out.p, (int)maxOut, d_ctl.p, d_ctl.p + 1, stream_));
                EDLIB_AMD_HIP(hipMemcpyAsync(ctl, d_ctl.p, (ns + 1) * sizeof(int), hipMemcpyDeviceToHost, stream_));
                EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
                if ((size_t)ctl[0] <= maxOut) break;
                maxOut = (size_t)ctl[0];                       // rare: a second gather into a list of the right size
            }
            const size_t nc = std::min<size_t>((size_t)ctl[0], maxOut);
            PinBuf outPin; EDLIB_AMD_HIP(outPin.alloc(std::max<size_t>(1, 2 * nc) * sizeof(int)));
            if (nc) {
                EDLIB_AMD_HIP(hipMemcpyAsync(outPin.p, d_out.p, 2 * nc * sizeof(int), hipMemcpyDeviceToHost, stream_));
                EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            }
            const int* o = reinterpret_cast<const int*>(outPin.p);
            for (size_t i = 0; i < nc; ++i)
                if ((size_t)o[2 * i] < who.size()) cand->push_back({who[o[2 * i]], o[2 * i + 1]});
            for (size_t s = 0; s < who.size(); ++s) if (ctl[1 + s]) (*overflow)[who[s]] = 1;
        } else {
            EDLIB_AMD_HIP(d_best.alloc(ns)); EDLIB_AMD_HIP(d_total.alloc(ns)); EDLIB_AMD_HIP(d_pos.alloc(ns * 16)); EDLIB_AMD_HIP(d_flags.alloc(ns + 1));
            EDLIB_AMD_HIP(launch_merge_segments(g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, g.numSegments, (int)cap, g.nslots, nullptr,
                                                16, d_best.p, d_total.p, d_pos.p, d_flags.p, stream_));
            PinBuf pin; EDLIB_AMD_HIP(pin.alloc(2 * ns * sizeof(int)));
            int* h = reinterpret_cast<int*>(pin.p);
            EDLIB_AMD_HIP(hipMemcpyAsync(h, d_best.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipMemcpyAsync(h + ns, d_total.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            for (size_t s = 0; s < who.size(); ++s) (*best)[who[s]] = h[ns + s] > 0 ? h[s] : -1;
        }
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));          // the group's buffers die here
    }
    return 0;
}

int Batch::solveLongReads(std::vector<UnitResult>& res, std::vector<int>& fallback)
{
    const size_t n = longUnits_.size();
    fallback.clear();
    if (!n) return 0;
    stats.path |= 4;
    const int T = tlen(0);
    const int kpMax = 56;   // (24 / 40 / 56 measured: 494 / 446 / 427 ms per 2,048 ONT-like 10 kb reads)
    static const bool dbg = getenv("EDLIB_AMD_DEBUG") != nullptr;
    Lap lap;
    auto kmax_of = [&](int m) { return (cfg_.k < 0 || cfg_.k > m) ? m : cfg_.k; };     // HW clamps k to m (edlib.cpp:566-568)

    // ---- first level: divergence of a sample (the HW distance of the first 256 rows of up to 128 strided queries,
    // one small banded scan), extrapolated to every query's length -- the reference starts every query at k = 64
    // and doubles (edlib.cpp:197-217); any threshold >= the distance gives the same answer
    double rate = 0.0;
    {
        const size_t ns = std::min<size_t>(n, 128);
        std::vector<Piece> probe(ns);
        for (size_t i = 0; i < ns; ++i) {
            const int u = longUnits_[(size_t)((long long)i * n / ns)];
            const int len = std::min(kPieceRows, qlen(u));
            probe[i] = Piece{qoff_[u], len, std::min(len, 80)};
        }
        std::vector<int> best;
        if (scanPieces(probe, false, nullptr, nullptr, &best)) return 1;
        std::vector<double> r(ns);
        for (size_t i = 0; i < ns; ++i) r[i] = best[i] < 0 ? 1.0 : (double)best[i] / probe[i].len;
        std::sort(r.begin(), r.end());
        rate = r[ns / 2];
    }
    lap("long: probe");
    std::vector<int> level(n);                     // current threshold of every query; -1 resolved, -2 handed back
    for (size_t i = 0; i < n; ++i) {
        const int m = qlen(longUnits_[i]);
        // expected distance D = rate * m of a query like the sample's median, two standard deviations of a sum of m
        // Bernoulli trials above it (a threshold that is too generous costs band height in every piece scan:
        // about a word per 8 units of the piece threshold)
        const double D = rate * m;
        const int k0 = (int)std::ceil(D + 2.0 * std::sqrt(D) + 3.0);
        level[i] = std::min(kmax_of(m), k0);
    }
    std::vector<size_t> pending(n);
    for (size_t i = 0; i < n; ++i) pending[i] = i;

    std::vector<Piece> pieces; std::vector<int> pieceUnit, pieceEnd;      // piece -> index into longUnits_, end row (exclusive)
    std::vector<std::pair<int, int>> cand; std::vector<uint8_t> ovf;
    std::vector<UnitSpec> vunits; std::vector<size_t> vwho; std::vector<int> vbase;
    for (int round = 0; round < 40 && !pending.empty(); ++round) {
        std::vector<size_t> next;
        // chunks of at most ~2M pieces (Peq rows and segment records of a scan stay within a few GB)
        size_t at = 0;
        while (at < pending.size()) {
            pieces.clear(); pieceUnit.clear(); pieceEnd.clear();
            std::vector<size_t> chunk;
            while (at < pending.size() && pieces.size() < (2u << 20)) {
                const size_t i = pending[at++];
                const int u = longUnits_[i], m = qlen(u), k = level[i];
                const LevelPlan L = plan_level(m, k, kpMax);
                if (!L.ok) { level[i] = -2; fallback.push_back(u); continue; }
                chunk.push_back(i);
                for (int q = 0; q < L.p; ++q) {
                    pieces.push_back(Piece{qoff_[u] + (long long)q * L.partLen, L.rows, L.kp});
                    pieceUnit.push_back((int)i); pieceEnd.push_back(q * L.partLen + L.rows);
                }
            }
            if (chunk.empty()) continue;
            if (scanPieces(pieces, true, &cand, &ovf, nullptr)) return 1;
            lap("long: filter scan");
            // ---- windows of end columns per query
            std::vector<std::vector<std::pair<int, int>>> win(n > 0 ? chunk.size() : 0);
            std::vector<int> slotOf(n, -1);                       // index into longUnits_ -> index into chunk
            for (size_t c = 0; c < chunk.size(); ++c) slotOf[chunk[c]] = (int)c;
            std::vector<uint8_t> bad(chunk.size(), 0);
            for (size_t j = 0; j < pieces.size(); ++j) if (ovf[j]) bad[slotOf[pieceUnit[j]]] = 1;
            for (const auto& cb : cand) {
                const int j = cb.first, i = pieceUnit[j], c = slotOf[i];
                if (bad[c]) continue;
                const int m = qlen(longUnits_[i]), k = level[i], below = m - pieceEnd[j];
                const long long lo = 16LL * cb.second + below - k, hi = 16LL * cb.second + 15 + below + k;
                if (hi < 0 || lo > T - 1) continue;
                win[c].push_back({(int)std::max<long long>(lo, 0), (int)std::min<long long>(hi, T - 1)});
            }
            vunits.clear(); vwho.clear(); vbase.clear();
            for (size_t c = 0; c < chunk.size(); ++c) {
                const size_t i = chunk[c];
                const int u = longUnits_[i], m = qlen(u), k = level[i];
                if (bad[c]) { level[i] = -2; fallback.push_back(u); continue; }
                auto& w = win[c];
                std::sort(w.begin(), w.end());
                // windows closer than a warm-up apart are one window (the second scan would cover the first's columns anyway)
                size_t nw = 0;
                for (size_t q = 0; q < w.size(); ++q) {
                    if (nw && (long long)w[q].first <= (long long)w[nw - 1].second + m + k) w[nw - 1].second = std::max(w[nw - 1].second, w[q].second);
                    else w[nw++] = w[q];
                }
                w.resize(nw);
                // windows that together cover most of the target cost more than one scan of it (each drags m + k columns of warm-up)
                long long cover = 0;
                for (size_t q = 0; q < nw; ++q) cover += (long long)w[q].second - w[q].first + 1 + m + k;
                if (nw > (size_t)kMaxWindows || cover > (long long)T) { level[i] = -2; fallback.push_back(u); continue; }
                for (size_t q = 0; q < nw; ++q) {
                    const int start = (int)std::max<long long>(0, (long long)w[q].first - m - k);
                    UnitSpec v{qoff_[u], m, 1, tbase(u) + start, w[q].second - start + 1, 1, k};
                    v.skip = w[q].first - start;
                    vunits.push_back(v); vwho.push_back(c); vbase.push_back(start);
                }
            }
            lap("long: windows");
            SolveOut so;
            if (!vunits.empty() && solveSemiGlobalUnits(EDLIB_MODE_HW, true, vunits, so)) return 1;
            lap("long: verification");
            // ---- the smallest score over a query's windows, its columns in ascending order (windows are disjoint and sorted)
            std::vector<int> bestOf(chunk.size(), -1);
            for (size_t v = 0; v < vunits.size(); ++v)
                if (so.score[v] >= 0 && (bestOf[vwho[v]] < 0 || so.score[v] < bestOf[vwho[v]])) bestOf[vwho[v]] = so.score[v];
            std::vector<std::vector<int>> posOf(chunk.size());
            for (size_t v = 0; v < vunits.size(); ++v) {
                const size_t c = vwho[v];
                if (so.score[v] < 0 || so.score[v] != bestOf[c]) continue;
                for (long long q = so.posStart[v]; q < so.posStart[v + 1]; ++q) posOf[c].push_back(so.posFlat[q] + vbase[v]);
            }
            for (size_t c = 0; c < chunk.size(); ++c) {
                const size_t i = chunk[c];
                if (level[i] < 0) continue;
                const int u = longUnits_[i], m = qlen(u), k = level[i];
                if (bestOf[c] >= 0) {                                  // <= k by construction: exact
                    finalize_semiglobal(res[u], cfg_.k, m, bestOf[c], posOf[c].data(), (long long)posOf[c].size());
                    level[i] = -1;
                } else if (k >= kmax_of(m)) {                          // nothing within the caller's k (k < m here: level m is never filterable)
                    finalize_semiglobal(res[u], cfg_.k, m, -1, nullptr, 0);
                    level[i] = -1;
                } else {
                    // the reference doubles (edlib.cpp:197-217); what fails the first level here is mostly unrelated sequence
                    // that will fail every level, so the steps are larger
                    level[i] = std::min(kmax_of(m), 4 * k);
                    next.push_back(i);
                }
            }
            lap("long: finalize");
            if (dbg) fprintf(stderr, "[edlib_amd] long reads round %d: %zu queries, %zu pieces, %zu candidates, %zu windows\n",
                             round, chunk.size(), pieces.size(), cand.size(), vunits.size());
        }
        pending.swap(next);
    }
    for (size_t i : pending) { level[i] = -2; fallback.push_back(longUnits_[i]); }
    std::sort(fallback.begin(), fallback.end());
    return 0;
}

}  // namespace edlib_amd

namespace edlib_amd {

// Queries the piece filter handed back (mostly unrelated sequence: every row of every column is needed) that are taller
// than the lane kernel's 32 words: cut into strips of `stripRows` rows, one launch of scan_reads_full_kernel per strip
// level, the horizontal deltas of a strip's bottom row handed to the strip below through HBM (two bits per column).
// Queries with the same number of strips share the target segmentation; the last strip of a query follows row m - 1
// and records best / count / positions like any read.  Queries whose end-location list overflows go on to kernel W.
int Batch::solveTallFull(const std::vector<int>& units, std::vector<UnitResult>& res, std::vector<int>& handBack)
{
    handBack.clear();
    const int T = tlen(0);
    const int kNoCap = 0x3fffffff;
    const int stripWords = syms_ == 4 ? kMaxLongReadWords4 : kMaxLongReadWords, stripRows = 32 * stripWords;
    std::vector<std::vector<int>> byStrips;                       // units by number of strips
    for (int u : units) {
        const size_t ns = (size_t)((qlen(u) + stripRows - 1) / stripRows);
        if (byStrips.size() <= ns) byStrips.resize(ns + 1);
        byStrips[ns].push_back(u);
    }
    for (size_t ns = 2; ns < byStrips.size(); ++ns) {
        const std::vector<int>& set = byStrips[ns];
        if (set.empty()) continue;
        const int nl = (int)set.size();
        int mmax = 0;
        for (int u : set) mmax = std::max(mmax, qlen(u));
        // one segmentation for every launch of the set: warm-up 2m - 1 of its tallest query
        const int warm = 2 * mmax - 1;
        const long long nrblk = (nl + 63) / 64;
        // as many segments as it takes to fill the chip (a wave of this kernel holds 32 KB of LDS rows: ~1280 resident
        // waves), none shorter than ONE warm-up.  (Rounds 2-3 asked for four: 559 unrelated reads of 6,000 bases then are 9 x 104
        // waves, fewer than the chip has SIMDs, and 410 of 8,192 bases fell below the floor underneath and went to kernel W.
        // Measured per 16,384-read-equivalents batch, the round-4 sweep: 6,000 bases 860 -> 745 ms, 8,192: 959 -> 803,
        // 10,000: 969 -> 933; a wave that recomputes as many columns as it owns still beats an idle SIMD.)
        const long long warms = 1;                                    // (shortest segment in warm-ups)
        const long long maxS = std::max<long long>(1, std::min<long long>(std::min(65535, std::max(1, T / 4096)), T / (warms * warm)));
        // ONE round of waves, at most one per SIMD (tall_round_waves): measured per 16,384-read-equivalents batch with
        // the round-4 sweep, 6,000-base reads (9 read blocks) at 768 / 896 / 1024 / 1152 / 1792 / 2043 waves: 672 / 628 /
        // 593 / 822 / 679 / 645 ms -- the kernel is bound by VALU issue, a SIMD with two of its waves takes twice as long and
        // the launch waits for it, and every extra segment is another warm-up of 2m - 1 columns.  Rounds 2-3 aimed at 2048
        // waves and four warm-ups per segment: 860 ms there (and 158 x 13 = 2054 waves for 4,096-base reads: 566 ms against
        // 438 at 78 x 13).
        const long long S = std::max<long long>(1, std::min<long long>(std::max(1LL, tall_round_waves() / nrblk), maxS));
        // The strip levels of a set run one after the other, and a segment cannot be shorter than a few warm-ups (2m - 1
        // columns each): a handful of very tall queries does not fill the chip this way (335 queries of 10 kb: 6 x 62
        // waves).  Those stay on kernel W, which cuts the target of each unit on its own.
        const char* mw = getenv("EDLIB_AMD_TALL_MIN_WAVES");          // (read per call: the tests lower it)
        // (512: the floor of 1024 / nrblk alone never goes below 513 waves; 410 reads of 8,192 bases on 532 waves: 807 ms, on kernel W: 959)
        if (nrblk * S < (mw ? atoll(mw) : 512)) { handBack.insert(handBack.end(), set.begin(), set.end()); continue; }
        const int segLen = roundup((int)((T + S - 1) / S), 16);
        const int numSegments = (T + segLen - 1) / segLen;
        const int chainBlocks = (segLen + warm) / 16 + 3;
        const size_t streamWords = (size_t)numSegments * chainBlocks * nl;
        DevBuf<uint32_t> streamA, streamB;
        EDLIB_AMD_HIP(streamA.alloc(streamWords));
        if (ns > 2) EDLIB_AMD_HIP(streamB.alloc(streamWords));
        // lane of a unit in the launches of the full strips: its index in the set
        PinBuf srcPin; EDLIB_AMD_HIP(srcPin.alloc((size_t)nl * sizeof(int)));
        DevBuf<int> d_src;
        EDLIB_AMD_HIP(d_src.alloc(nl));
        for (size_t level = 0; level < ns; ++level) {
            const bool last = level + 1 == ns;
            // pieces of this level, by word count (full strips: one group)
            std::vector<std::vector<int>> byWords(kMaxLongReadWords4 + 1);
            for (int i = 0; i < nl; ++i) {
                const int rows = last ? qlen(set[i]) - (int)level * stripRows : stripRows;
                byWords[read_group_words(rows)].push_back(i);
            }
            uint32_t* out = last ? nullptr : ((level & 1) ? streamB.p : streamA.p);
            const uint32_t* in = level == 0 ? nullptr : (((level - 1) & 1) ? streamB.p : streamA.p);
            for (int w = 1; w <= kMaxLongReadWords4; ++w) {
                const std::vector<int>& who = byWords[w];
                if (who.empty()) continue;
                ReadGroup g;
                g.nwords = w; g.nslots = roundup((int)who.size(), 64);
                const size_t nsl = (size_t)g.nslots, nreal = who.size();
                PinBuf pin;                                   // bounds (2 long long per slot) | perm | thr | rowBase | src
                EDLIB_AMD_HIP(pin.alloc(nsl * (2 * sizeof(long long) + 4 * sizeof(int))));
                long long* pb = reinterpret_cast<long long*>(pin.p);
                int* perm = reinterpret_cast<int*>(pb + 2 * nsl); int* thr = perm + nsl; int* rowBase = thr + nsl; int* src = rowBase + nsl;
                for (size_t sl = 0; sl < nsl; ++sl) {
                    const bool real = sl < nreal;
                    const int u = real ? set[who[sl]] : 0, m = real ? qlen(u) : 0;
                    const int r0 = (int)level * stripRows, rows = real ? (last ? m - r0 : stripRows) : 0;
                    pb[2 * sl] = real ? qoff_[u] + r0 : 0; pb[2 * sl + 1] = pb[2 * sl] + rows;
                    perm[sl] = real ? (int)(2 * sl) : -1;
                    thr[sl] = (real && last) ? ((cfg_.k < 0 || cfg_.k > m) ? m : cfg_.k) : -1;       // strips above the last follow no score
                    rowBase[sl] = r0; src[sl] = real ? who[sl] : 0;
                }
                DevBuf<long long> d_pb; DevBuf<int> d_meta;      // d_meta: perm | thr | rowBase | src
                EDLIB_AMD_HIP(d_pb.alloc(2 * nsl)); EDLIB_AMD_HIP(d_meta.alloc(4 * nsl));
                EDLIB_AMD_HIP(hipMemcpyAsync(d_pb.p, pb, 2 * nsl * sizeof(long long), hipMemcpyHostToDevice, stream_));
                EDLIB_AMD_HIP(hipMemcpyAsync(d_meta.p, perm, 4 * nsl * sizeof(int), hipMemcpyHostToDevice, stream_));
                g.d_perm.alias(d_meta.p, nsl);
                EDLIB_AMD_HIP(g.d_qlen.alloc(nsl)); EDLIB_AMD_HIP(g.d_kinit.alloc(nsl)); EDLIB_AMD_HIP(g.d_alphaExtra.alloc(nsl));
                EDLIB_AMD_HIP(g.d_peq.alloc(nsl * (size_t)syms_ * w));
                const size_t S2 = (size_t)numSegments;
                EDLIB_AMD_HIP(g.d_segBest.alloc(nsl * S2)); EDLIB_AMD_HIP(g.d_segCnt.alloc(nsl * S2));
                EDLIB_AMD_HIP(g.d_segPos.alloc(last ? nsl * S2 * 8 : 1));
                EDLIB_AMD_HIP(launch_build_peq_reads(w, syms_, d_qpool_.p, d_pb.p, g.d_perm.p, g.nslots, d_eqtbl_.p, d_presence_.p,
                                                     -1, g.d_peq.p, g.d_qlen.p, g.d_kinit.p, g.d_alphaExtra.p, stream_));
                // a full strip's launch has the set's lanes in set order (who[sl] == sl): its stream is indexed by that lane
                chain_ = ChainArgs{in, out, d_meta.p + 3 * nsl, nl, chainBlocks, d_meta.p + 2 * nsl};
                const int rc = scanGroup(g, EDLIB_MODE_HW, nullptr, (int)nreal, kNoCap, d_meta.p + nsl, numSegments, segLen, warm,
                                         g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, last ? 8 : 0, nullptr, nullptr, /*unbanded=*/true);
                chain_ = ChainArgs{};
                if (rc) return 1;
                if (last) {
                    DevBuf<int> d_best, d_total, d_pos, d_flags;
                    EDLIB_AMD_HIP(d_best.alloc(nsl)); EDLIB_AMD_HIP(d_total.alloc(nsl)); EDLIB_AMD_HIP(d_pos.alloc(nsl * 16)); EDLIB_AMD_HIP(d_flags.alloc(nsl + 1));
                    EDLIB_AMD_HIP(launch_merge_segments(g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, numSegments, 8, (int)nreal, nullptr, 16,
                                                        d_best.p, d_total.p, d_pos.p, d_flags.p, stream_));
                    PinBuf outPin; EDLIB_AMD_HIP(outPin.alloc(nsl * 19 * sizeof(int)));
                    int* h = reinterpret_cast<int*>(outPin.p);
                    EDLIB_AMD_HIP(hipMemcpyAsync(h, d_best.p, nreal * sizeof(int), hipMemcpyDeviceToHost, stream_));
                    EDLIB_AMD_HIP(hipMemcpyAsync(h + nsl, d_total.p, nreal * sizeof(int), hipMemcpyDeviceToHost, stream_));
                    EDLIB_AMD_HIP(hipMemcpyAsync(h + 2 * nsl, d_flags.p, nreal * sizeof(int), hipMemcpyDeviceToHost, stream_));
                    EDLIB_AMD_HIP(hipMemcpyAsync(h + 3 * nsl, d_pos.p, nreal * 16 * sizeof(int), hipMemcpyDeviceToHost, stream_));
                    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
                    for (size_t sl = 0; sl < nreal; ++sl) {
                        const int u = set[who[sl]];
                        if (h[2 * nsl + sl]) { handBack.push_back(u); continue; }      // more end locations than a slot keeps
                        finalize_semiglobal(res[u], cfg_.k, qlen(u), h[sl], h + 3 * nsl + sl * 16, h[sl] < 0 ? 0 : h[nsl + sl]);
                    }
                }
                EDLIB_AMD_HIP(hipStreamSynchronize(stream_));            // the group's buffers die here
            }
        }
    }
    return 0;
}

}  // namespace edlib_amd
