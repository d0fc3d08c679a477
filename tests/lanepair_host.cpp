// lanepair_host.cpp -- TEST INFRASTRUCTURE: the device code of the lane-per-pair NW scan (edlib_amd/csrc/lanepair_core.hpp),
// compiled for the host and run one lane at a time, so that tests/test_lanepair_model.py can check the window logic (private
// row offsets, virtual rows, slides, dead tests and trims, the final decode) against the oracle without a GPU.
// Built by tests/test_lanepair_model.py with g++ into build/liblanepair_host.so; never linked into the product.
#include "../edlib_amd/csrc/lanepair_core.hpp"
#include <stddef.h>
#include <vector>

using namespace edlib_amd::lanepair;

template <int W>
static int run(const std::vector<Plane2>& planes, const std::vector<Tgt2>& tgt, int m, int T, int K, int na, int nblk, unsigned deny, int* ws)
{
    return lp_scan<W>(planes.data(), 0u, tgt.data(), 0u, m, T, K, na, nblk, deny, ws);
}

// q, t: symbol codes 0..3.  extraWords / extraBlocks: the wave's maxima exceed this lane's own needs by that much.
// Returns the computed value (exact iff <= K), -2 if the band does not fit W words, -3 if |T - m| > K.
extern "C" int lanepair_host_nw(const unsigned char* q, int m, const unsigned char* t, int T, int K, int W,
                                unsigned denySeed, int extraWords, int extraBlocks, int* wordSteps)
{
    int na = lp_band_words(m, T, K);
    if (na == 0) return -3;
    na += extraWords;
    if (na < 3) na = 3;
    if (na > W) return -2;
    std::vector<Plane2> planes((size_t)(m + 31) / 32, Plane2{0u, 0u});
    for (int i = 0; i < m; ++i) {
        planes[i >> 5].q0 |= (u32)(q[i] & 1u) << (i & 31);
        planes[i >> 5].q1 |= (u32)((q[i] >> 1) & 1u) << (i & 31);
    }
    const int nblk = (T + 31) / 32 + extraBlocks;
    std::vector<Tgt2> tgt((size_t)(T + 31) / 32 + 1, Tgt2{0u, 0u});
    for (int c = 0; c < T; ++c) {
        tgt[c >> 5].t0 |= (u32)(t[c] & 1u) << (c & 31);
        tgt[c >> 5].t1 |= (u32)((t[c] >> 1) & 1u) << (c & 31);
    }
    switch (W) {
        case 8: return run<8>(planes, tgt, m, T, K, na, nblk, denySeed, wordSteps);
        case 16: return run<16>(planes, tgt, m, T, K, na, nblk, denySeed, wordSteps);
        case 48: return run<48>(planes, tgt, m, T, K, na, nblk, denySeed, wordSteps);
        default: return -4;
    }
}
