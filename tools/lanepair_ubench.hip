// lanepair_ubench.hip -- VERDICT r5 item 1, step 1: the lane-per-pair NW scan (edlib_amd/csrc/lanepair_kernels.hpp, the
// product's own kernels) on a config-4-like batch, before it is wired into the engine: ms per launch with the
// data-dependent trims and as a static band, word-columns computed, and a sample of units checked against the SAME
// code run on the host (which tests/test_lanepair_model.py pins against the oracle).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iedlib_amd/csrc tools/lanepair_ubench.hip -o build/lanepair_ubench
//   build/lanepair_ubench [--units 100000] [--len 10000] [--k 1280] [--reps 5] [--sub 0.04 --ins 0.04 --del 0.04]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "lanepair_kernels.hpp"

using namespace edlib_amd::lanepair;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

struct Rng { uint64_t s; uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); } double u() { return next() / 2147483648.0; } };

int main(int argc, char** argv)
{
    int units = 100000, len = 10000, K = 1280, reps = 5;
    double sub = 0.04, ins = 0.04, del = 0.04;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--units")) units = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--len")) len = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--k")) K = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--sub")) sub = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--ins")) ins = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--del")) del = atof(argv[i + 1]);
    }
    int W = window_for_k(K);
    for (int i = 1; i + 1 < argc; i += 2) if (!strcmp(argv[i], "--w")) W = atoi(argv[i + 1]);
    if (!W) { fprintf(stderr, "K too large\n"); return 2; }
    // ---- the batch: targets i.i.d. ACGT, queries = target with edits (config 4's recipe), bytes as the engine holds them
    std::vector<long long> qoff(units + 1), toff(units + 1);
    std::vector<std::vector<uint8_t>> qs(units);
    std::vector<uint8_t> tpool((size_t)units * len);
    {
        const int nt = std::max(1u, std::thread::hardware_concurrency());
        std::vector<std::thread> th;
        for (int w = 0; w < nt; ++w) th.emplace_back([&, w] {
            for (int i = w; i < units; i += nt) {
                Rng r{0x9e3779b97f4a7c15ull * (uint64_t)(i + 1) + 12349};
                uint8_t* t = &tpool[(size_t)i * len];
                std::vector<uint8_t>& q = qs[i];
                q.reserve(len + len / 8);
                for (int c = 0; c < len; ++c) {
                    const uint8_t b = "ACGT"[r.next() & 3];
                    t[c] = b;
                    const double x = r.u();
                    if (x < del) continue;
                    if (x < del + sub) q.push_back("ACGT"[(((b >> 1) & 3) + 1 + r.next() % 3) & 3]); else q.push_back(b);
                    if (r.u() < ins) q.push_back("ACGT"[r.next() & 3]);
                }
            }
        });
        for (auto& t : th) t.join();
    }
    std::vector<LaneUnit> lu(units);
    long long qb = 0, pw = 0, tw = 0;
    for (int i = 0; i < units; ++i) {
        qoff[i] = qb; toff[i] = (long long)i * len;
        lu[i] = LaneUnit{qoff[i], toff[i], pw, tw, (int)qs[i].size(), len};
        qb += (long long)qs[i].size(); pw += ((long long)qs[i].size() + 31) / 32; tw += (len + 31) / 32;
    }
    std::vector<uint8_t> qpool((size_t)qb + 64);
    for (int i = 0; i < units; ++i) memcpy(&qpool[qoff[i]], qs[i].data(), qs[i].size());
    uint8_t tlut[256] = {0}; uint16_t eqtbl[256] = {0};
    for (int s = 0; s < 4; ++s) { tlut[(uint8_t)"ACGT"[s]] = (uint8_t)s; eqtbl[(uint8_t)"ACGT"[s]] = (uint16_t)(1u << s); }

    uint8_t *d_q, *d_t, *d_tlut; uint16_t* d_eq; long long *d_qoff, *d_toff; LaneUnit* d_lu; Plane2* d_pl; Tgt2* d_tg; int *d_flags, *d_alpha, *d_score;
    unsigned long long* d_ws;
    CK(hipMalloc(&d_q, qpool.size())); CK(hipMalloc(&d_t, tpool.size() + 64)); CK(hipMalloc(&d_tlut, 256)); CK(hipMalloc(&d_eq, 512));
    CK(hipMalloc(&d_qoff, 8 * units)); CK(hipMalloc(&d_toff, 8 * units)); CK(hipMalloc(&d_lu, sizeof(LaneUnit) * units));
    CK(hipMalloc(&d_pl, sizeof(Plane2) * (pw + 1))); CK(hipMalloc(&d_tg, sizeof(Tgt2) * (tw + 1)));
    CK(hipMalloc(&d_flags, 4 * units)); CK(hipMalloc(&d_alpha, 4 * units)); CK(hipMalloc(&d_score, 4 * units)); CK(hipMalloc(&d_ws, 8));
    CK(hipMemcpy(d_q, qpool.data(), qpool.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_t, tpool.data(), tpool.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tlut, tlut, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(d_eq, eqtbl, 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_qoff, qoff.data(), 8 * units, hipMemcpyHostToDevice)); CK(hipMemcpy(d_toff, toff.data(), 8 * units, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lu, lu.data(), sizeof(LaneUnit) * units, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    PackArgs pa{}; pa.qpool = d_q; pa.tpool = d_t; pa.tlut = d_tlut; pa.eqtbl = d_eq; pa.sigmaT = 4; pa.perUnit = 0;
    pa.units = d_lu; pa.numUnits = units; pa.planes = d_pl; pa.tgts = d_tg; pa.flags = d_flags; pa.alphaOut = d_alpha;
    float packMs = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0)); CK(launch_pack(pa, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < packMs) packMs = ms;
    }
    ScanArgs sa{}; sa.units = d_lu; sa.flags = d_flags; sa.numUnits = units; sa.planes = d_pl; sa.tgts = d_tg; sa.rate = -1.0f; sa.kcap = 0x3fffffff; sa.kmax = K; sa.outScore = d_score; sa.wordSteps = d_ws;
    std::vector<int> score(units), scoreStatic(units);
    for (int mode = 0; mode < 2; ++mode) {
        sa.denySeed = mode ? 0xffffffffu : 0u;
        float best = 1e9f; unsigned long long ws = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(d_ws, 0, 8));
            CK(hipEventRecord(e0, 0)); CK(launch_scan(sa, W, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(&ws, d_ws, 8, hipMemcpyDeviceToHost));
        }
        CK(hipMemcpy((mode ? scoreStatic : score).data(), d_score, 4 * units, hipMemcpyDeviceToHost));
        const double cells = (double)units * len * len;
        printf("{\"kernel\": \"lanepair_scan_kernel<%d>\", \"band\": \"%s\", \"units\": %d, \"len\": %d, \"K\": %d, \"ms\": %.3f, \"word_columns\": %.4g, "
               "\"words_per_unit_column\": %.2f, \"ns_per_wave_word_column\": %.3f, \"gcups\": %.0f}\n", W, mode ? "static" : "trimmed", units, len, K, best,
               (double)ws, (double)ws / ((double)units * len), best * 1e6 / ((double)ws / 64.0) * 1024.0, cells / best / 1e6);
    }
    printf("{\"kernel\": \"lanepair_pack_kernel\", \"ms\": %.3f, \"GBps\": %.0f}\n", packMs, (double)(qb + (long long)units * len) / packMs / 1e6);
    // ---- a sample against the host run of the same code (one lane at a time, trims as that lane alone would take them)
    int checked = 0, equal = 0, above = 0, staticEqual = 0, alphaOk = 0;
    std::vector<int> alpha(units); CK(hipMemcpy(alpha.data(), d_alpha, 4 * units, hipMemcpyDeviceToHost));
    for (int i = 0; i < units; i += std::max(1, units / 256)) {
        std::vector<Plane2> pl((qs[i].size() + 31) / 32, Plane2{0, 0}); std::vector<Tgt2> tg((len + 31) / 32 + 1, Tgt2{0, 0});
        for (size_t r = 0; r < qs[i].size(); ++r) { const u32 c = tlut[qs[i][r]]; pl[r >> 5].q0 |= (c & 1u) << (r & 31); pl[r >> 5].q1 |= ((c >> 1) & 1u) << (r & 31); }
        for (int c = 0; c < len; ++c) { const u32 s = tlut[tpool[(size_t)i * len + c]]; tg[c >> 5].t0 |= (s & 1u) << (c & 31); tg[c >> 5].t1 |= ((s >> 1) & 1u) << (c & 31); }
        const int na = lp_band_words((int)qs[i].size(), len, K);
        int want = kNoBand;
        if (na > 0 && na <= W) want = lp_scan<48>(pl.data(), 0u, tg.data(), 0u, (int)qs[i].size(), len, K, na < 3 ? 3 : na, (len + 31) / 32, 0u, nullptr);
        ++checked;
        // exact iff <= K: the device may hold more words than the lane alone (wave maxima), so values above K may differ
        if (want <= K ? score[i] == want : score[i] >= kAboveFinal) ++equal;
        if (want > K) ++above;
        if (want <= K ? scoreStatic[i] == want : scoreStatic[i] >= kAboveFinal) ++staticEqual;
        alphaOk += alpha[i] == 4;
    }
    printf("{\"host_check\": {\"sampled\": %d, \"trimmed_agree\": %d, \"static_agree\": %d, \"above_k\": %d, \"alphabet_ok\": %d}}\n", checked, equal, staticEqual, above, alphaOk);
    return (equal == checked && staticEqual == checked) ? 0 : 1;
}
