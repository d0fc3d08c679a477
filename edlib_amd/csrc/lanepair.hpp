// lanepair.hpp -- launch interface of the lane-per-pair NW distance scan (lanepair_core.hpp: what a lane does;
// lanepair_kernels.hpp: the kernels).  Host-visible structs and launchers only: the engine includes this, the kernels are
// compiled in their own translation units (lanepair_kernels.hip, lanepair_kernels24.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LANEPAIR_HAVE_PLANES 1
namespace edlib_amd {
namespace lanepair {

struct Plane2 { uint32_t q0, q1; };   // 32 query rows: bit i of q0 / q1 = low / high bit of the symbol code of row 32 w + i
struct Tgt2 { uint32_t t0, t1; };     // 32 target columns: bit j of t0 / t1 = low / high bit of the symbol code of column 32 b + j

// outScore of a unit the level did not settle:
static constexpr int kNoBand = 0x3fffffff;        // not scanned (foreign symbols, |T - m| beyond its threshold's band): nothing known
static constexpr int kAboveOpen = 0x3ffffffe;     // its distance exceeds its threshold (which is below the caller's k): the next level
static constexpr int kAboveFinal = 0x3ffffffd;    // its distance exceeds the caller's k: final

// The threshold of a unit at this level: the probe's rate (edits per base, edlibAlign's k-doubling replaced by an estimate:
// any threshold >= the distance gives the same answer, edlib.cpp:197-217) times the shorter length, plus what the length
// difference exceeds three sigma of an indel drift by, plus 3.5 sigma of a count of that mean (config 4: distances of 1136 +- 28 get
// 1143 + 118 + 16: what a unit's own scatter and a probe that reads 2 % low leave is still three of its sigmas), capped by the caller's k and
// by what the window holds.  One formula for the kernel and for whoever wants to know what a unit was scanned with.
__host__ __device__ inline int unit_threshold(int m, int T, float rate, int kcap, int kmax)
{
    const float mean = rate * (float)(m < T ? m : T);
    const float sd = sqrtf(mean > 1.0f ? mean : 1.0f);
    const float diff = (float)(m > T ? m - T : T - m);
    const float tail = diff > 3.0f * sd ? diff - 3.0f * sd : 0.0f;
    long long k = (long long)(mean + tail + 3.5f * sd + 16.0f);
    if (k > kmax) k = kmax;
    if (k > kcap) k = kcap;
    return (int)k;
}

// A unit as the kernels read it: where its bytes are, where its packed forms go, how long it is.
struct LaneUnit {
    long long qoff, toff;     // first byte of the query / target in the batch's pools
    long long planeOff;       // first Plane2 of the query (ceil(m / 32) entries)
    long long tgtOff;         // first Tgt2 of the target (ceil(T / 32) entries)
    int m, T;
};

struct PackArgs {
    const uint8_t* qpool;
    const uint8_t* tpool;
    const uint8_t* tlut;      // [256] target byte -> symbol id
    const uint16_t* eqtbl;    // [256] query byte -> set of target symbols it equals
    int sigmaT;
    int perUnit;              // 1: a unit's codes come from ITS target's distinct bytes (batches over more than four target symbols, identity
                              // equality only): a genome with N or soft-masked stretches keeps the level for every unit that stays within four
    const LaneUnit* units;
    int numUnits;
    Plane2* planes;
    Tgt2* tgts;
    int* flags;               // [units] written: 1 = a query byte that four target symbols cannot code (the unit stays on the rings)
    int* alphaOut;            // [units] or null: alphabetLength (edlib.cpp:162), counted on the way
};

struct ScanArgs {
    const LaneUnit* units;
    const int* flags;          // [units] 1 = not for this kernel (may be null)
    int numUnits;
    const Plane2* planes;
    const Tgt2* tgts;
    float rate;                // edits per base (the probe's estimate): a unit's threshold is unit_threshold(m, T, rate, kcap, kmax)
    int kcap;                  // the caller's k (0x3fffffff: none)
    int kmax;                  // window_max_k(W) of the launch, or a fixed threshold for every unit when rate < 0 (tests, tools)
    int* outScore;             // [units] D[m][T] when it is <= the unit's threshold (exact), else kAboveOpen / kAboveFinal / kNoBand
    unsigned long long* wordSteps;   // += 32-row word-columns computed (one atomic per wave), may be null
    unsigned denySeed;         // tests / benchmark: 0 = trims as the lanes vote; 0xffffffff = never trim (the static band)
};

// The instantiated windows (words of 32 rows per lane): 16 (four waves per SIMD), 24 (three), 42 and 48 (two).  A scan kernel
// is a ladder of one loop per window height down from W, and the allocator parks query planes in scratch in the stages where
// 4 NA + ~85 registers exceed its budget -- with the stages above 42 compiled in, from 30 words on (tools/lanepair_ubench.hip:
// 31 ns per word-column at 41 words against 18); so 42 is its own kernel, and 48 serves the thresholds beyond it.
// window_for_k: the smallest window that holds threshold K for every |T - m| -- the band is K + 1 diagonals at most and the
// window covers it at every c % 32: (K + 1 + 62) / 32 words; 0 = beyond this kernel.
inline int window_max_k(int W) { return 32 * W - 32; }
inline int window_for_k(int K)
{
    static const int ws[4] = {16, 24, 42, 48};
    for (int w : ws) if (K <= window_max_k(w)) return w;
    return 0;
}

}  // namespace lanepair

hipError_t launch_lanepair_pack(const lanepair::PackArgs& a, hipStream_t s);
hipError_t launch_lanepair_scan(const lanepair::ScanArgs& a, int W, hipStream_t s);     // W: 16, 24, 42 or 48 (a.kmax <= window_max_k(W))

}  // namespace edlib_amd
