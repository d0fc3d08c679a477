// lanepair_kernels.hpp -- the kernels around lanepair_core.hpp: packing a pair batch into the form a lane reads
// (query bit planes, target bit planes) and the scan itself, one wave = 64 units.  Included by lanepair_kernels.hip (the
// product) and tools/lanepair_ubench.hip (the micro-benchmark of DESIGN.md 4e): one source for both.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lanepair_core.hpp"

namespace edlib_amd {
namespace lanepair {

static constexpr int kNoBand = 0x3fffffff;        // outScore of a unit the level cannot hold (|T - m| > K, band beyond W words, foreign symbols)

// A unit as the scan reads it: where its packed forms start and how long it is.
struct LaneUnit {
    long long planeOff;       // first Plane2 of the query (ceil(m / 32) entries)
    long long tgtOff;         // first Tgt2 of the target (ceil(T / 32) entries)
    int m, T;
};

// ------------------------------------------------------------------ packing
// One workgroup per unit.  Lane i of a wave loads byte 64 k + i of the sequence (one coalesced line per wave), looks its
// symbol code up and two ballots ARE the bit planes of those 64 rows / columns (the cooperative build of
// build_peq_reads_kernel).  Query codes: the target symbol the byte equals (eqtbl: 16-bit set of target symbols per byte,
// EqualityDefinition edlib.cpp:63-94); a byte that equals none takes a code no target symbol has when there are fewer
// than four, and a byte that equals none of four symbols, or more than one, marks the unit as not for this kernel
// (flags[unit] = 1: it stays on the rings).  alphaOut (optional): the unit's alphabetLength (distinct bytes of query and
// target, edlib.cpp:162) -- every byte of both is in a register here anyway.
struct PackArgs {
    const long long* qoff;    // [units] first query byte
    const long long* toff;
    const int* qlen;          // (may be null: lengths from LaneUnit)
    const uint8_t* qpool;
    const uint8_t* tpool;
    const uint8_t* tlut;      // [256] target byte -> symbol id
    const uint16_t* eqtbl;    // [256] query byte -> set of target symbols it equals
    int sigmaT;
    const LaneUnit* units;
    int numUnits;
    Plane2* planes;
    Tgt2* tgts;
    int* flags;               // [units] written: 0 / 1
    int* alphaOut;            // [units] or null
};

__global__ void __launch_bounds__(256)
lanepair_pack_kernel(const PackArgs a)
{
    __shared__ uint8_t qcode[256];
    __shared__ uint32_t seen[8];
    __shared__ int bad;
    const int u = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const uint32_t set = a.eqtbl[tid];
        uint8_t c;
        if (set == 0) c = a.sigmaT < 4 ? 3 : 255;
        else if ((set & (set - 1)) == 0 && set < 16u) c = (uint8_t)(__ffs(set) - 1);
        else c = 255;
        qcode[tid] = c;
        if (tid < 8) seen[tid] = 0;
        if (tid == 0) bad = 0;
    }
    __syncthreads();
    const LaneUnit un = a.units[u];
    const uint8_t* q = a.qpool + a.qoff[u];
    const uint8_t* t = a.tpool + a.toff[u];
    Plane2* pl = a.planes + un.planeOff;
    Tgt2* tg = a.tgts + un.tgtOff;
    uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool foreign = false;
    for (int base = wave * 64; base < un.m; base += 256) {
        const int i = base + lane;
        uint32_t c = 0;
        if (i < un.m) {
            const uint8_t b = q[i];
            if (a.alphaOut) mine[b >> 5] |= 1u << (b & 31);
            c = qcode[b];
            if (c == 255) { foreign = true; c = 0; }
        }
        const unsigned long long b0 = __ballot(c & 1u), b1 = __ballot(c & 2u);
        if (lane == 0) pl[base >> 5] = Plane2{(u32)b0, (u32)b1};
        if (lane == 1 && base + 32 < un.m) pl[(base >> 5) + 1] = Plane2{(u32)(b0 >> 32), (u32)(b1 >> 32)};
    }
    for (int base = wave * 64; base < un.T; base += 256) {
        const int i = base + lane;
        uint32_t c = 0;
        if (i < un.T) {
            const uint8_t b = t[i];
            if (a.alphaOut) mine[b >> 5] |= 1u << (b & 31);
            c = a.tlut[b];
        }
        const unsigned long long b0 = __ballot(c & 1u), b1 = __ballot(c & 2u);
        if (lane == 0) tg[base >> 5] = Tgt2{(u32)b0, (u32)b1};
        if (lane == 1 && base + 32 < un.T) tg[(base >> 5) + 1] = Tgt2{(u32)(b0 >> 32), (u32)(b1 >> 32)};
    }
    if (foreign) bad = 1;
    if (a.alphaOut) {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (mine[k]) atomicOr(&seen[k], mine[k]);
    }
    __syncthreads();
    if (tid == 0) {
        a.flags[u] = bad;
        if (a.alphaOut) {
            int n = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) n += __popc(seen[k]);
            a.alphaOut[u] = n;
        }
    }
}

// ------------------------------------------------------------------ the scan
struct ScanArgs {
    const LaneUnit* units;
    const int* flags;          // [units] 1 = not for this kernel (may be null)
    int numUnits;
    const Plane2* planes;
    const Tgt2* tgts;
    int K;                     // the level's threshold
    int* outScore;             // [units] computed D[m][T] (exact iff <= K), kNoBand when the level cannot hold the unit
    unsigned long long* wordSteps;   // += 32-row word-columns computed (one atomic per wave), may be null
    unsigned denySeed;         // tests / benchmark: 0 = trims as the lanes vote; 0xffffffff = never trim (the static band)
};

template <int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W > 28 ? 2 : 4)))
lanepair_scan_kernel(const ScanArgs a)
{
    const int lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    const bool have = idx < a.numUnits;
    LaneUnit un = a.units[have ? idx : a.numUnits - 1];
    int need = have ? lp_band_words(un.m, un.T, a.K) : 0;
    if (need > W || (a.flags && have && a.flags[idx])) need = 0;
    const bool live = need > 0;
    int m = un.m, T = un.T, K = a.K;
    const Plane2* planes = a.planes + un.planeOff;
    const Tgt2* tgt = a.tgts + un.tgtOff;
    int nplanes = (un.m + 31) / 32;
    if (!live) { m = 1; T = 0; K = 0; nplanes = 1; }            // never active: reads its unit's first words, writes nothing
    const int naWave = lp_wave_max(need < 3 ? (live ? 3 : 0) : need);
    const int nblkWave = lp_wave_max(live ? (T + 31) / 32 : 0);
    int score = kNoBand;
    if (naWave > 0) {
        int ws = 0;
        unsigned deny = a.denySeed;
        const int got = lp_scan<W>(planes, nplanes, tgt, m, T, K, naWave, nblkWave, deny, &ws);
        if (live) score = got;
        if (a.wordSteps && lane == 0) atomicAdd(a.wordSteps, (unsigned long long)ws * 32ull * 64ull);
    }
    if (have) a.outScore[idx] = score;
}

// the smallest instantiated window that holds threshold K for every |T - m| (the band is K + 1 diagonals at most)
inline int window_for_k(int K)
{
    const int need = (K + 1 + 62) / 32;
    if (need <= 24) return 24;
    if (need <= 48) return 48;
    return 0;
}
inline int window_max_k(int W) { return 32 * W - 32; }     // lp_band_words(m, T, K) <= W for every unit: (K + 1 + 62) / 32 <= W

inline hipError_t launch_pack(const PackArgs& a, hipStream_t s)
{
    if (a.numUnits <= 0) return hipSuccess;
    hipLaunchKernelGGL(lanepair_pack_kernel, dim3(a.numUnits), dim3(256), 0, s, a);
    return hipGetLastError();
}
inline hipError_t launch_scan(const ScanArgs& a, int W, hipStream_t s)
{
    if (a.numUnits <= 0) return hipSuccess;
    const dim3 grid((a.numUnits + 63) / 64), block(64);
#ifndef LANEPAIR_ONLY48            /* (tools: one instantiation builds in half the time) */
    if (W == 24) hipLaunchKernelGGL(lanepair_scan_kernel<24>, grid, block, 0, s, a);
    else
#endif
    if (W == 48) hipLaunchKernelGGL(lanepair_scan_kernel<48>, grid, block, 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace lanepair
}  // namespace edlib_amd
