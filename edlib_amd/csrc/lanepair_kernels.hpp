// lanepair_kernels.hpp -- the kernels around lanepair_core.hpp: packing a pair batch into the form a lane reads
// (query bit planes, target bit planes) and the scan itself, one wave = 64 units.  Included by lanepair_kernels.hip (the
// product) and tools/lanepair_ubench.hip (the micro-benchmark of DESIGN.md 4e): one source for both.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lanepair.hpp"
#include "lanepair_core.hpp"

namespace edlib_amd {
namespace lanepair {

// ------------------------------------------------------------------ packing
// One workgroup per unit.  Lane i of a wave loads byte 64 k + i of the sequence (one coalesced line per wave), looks its
// symbol code up and two ballots ARE the bit planes of those 64 rows / columns (the cooperative build of
// build_peq_reads_kernel).  Query codes: the target symbol the byte equals (eqtbl: 16-bit set of target symbols per byte,
// EqualityDefinition edlib.cpp:63-94); a byte that equals none takes a code no target symbol has when there are fewer
// than four, and a byte that equals none of four symbols, or more than one, marks the unit as not for this kernel
// (flags[unit] = 1: it stays on the rings).  alphaOut (optional): the unit's alphabetLength (distinct bytes of query and
// target, edlib.cpp:162) -- every byte of both is in a register here anyway.
#if !defined(LANEPAIR_NO_PACK)
__global__ void __launch_bounds__(256)
lanepair_pack_kernel(const PackArgs a)
{
    __shared__ uint8_t qcode[256], tcode[256], mark[256];
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
        tcode[tid] = a.tlut[tid];
        mark[tid] = 0;
        if (tid == 0) bad = 0;
    }
    __syncthreads();
    const LaneUnit un = a.units[u];
    const uint8_t* q = a.qpool + un.qoff;
    const uint8_t* t = a.tpool + un.toff;
    Plane2* pl = a.planes + un.planeOff;
    Tgt2* tg = a.tgts + un.tgtOff;
    bool foreign = false;
    // four chunks of 64 rows per trip and wave: four loads in flight
    for (int base = wave * 256; base < un.m; base += 1024) {
        uint8_t b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = base + 64 * k + lane; b[k] = i < un.m ? q[i] : q[0]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + 64 * k + lane;
            uint32_t c = 0;
            if (i < un.m) {
                mark[b[k]] = 1;
                c = qcode[b[k]];
                if (c == 255) { foreign = true; c = 0; }
            }
            const unsigned long long b0 = __ballot(c & 1u), b1 = __ballot(c & 2u);
            const int w = (base + 64 * k) >> 5;
            if (lane == 0 && base + 64 * k < un.m) pl[w] = Plane2{(u32)b0, (u32)b1};
            if (lane == 1 && base + 64 * k + 32 < un.m) pl[w + 1] = Plane2{(u32)(b0 >> 32), (u32)(b1 >> 32)};
        }
    }
    for (int base = wave * 256; base < un.T; base += 1024) {
        uint8_t b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = base + 64 * k + lane; b[k] = i < un.T ? t[i] : t[0]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + 64 * k + lane;
            uint32_t c = 0;
            if (i < un.T) { mark[b[k]] = 1; c = tcode[b[k]]; }
            const unsigned long long b0 = __ballot(c & 1u), b1 = __ballot(c & 2u);
            const int w = (base + 64 * k) >> 5;
            if (lane == 0 && base + 64 * k < un.T) tg[w] = Tgt2{(u32)b0, (u32)b1};
            if (lane == 1 && base + 64 * k + 32 < un.T) tg[w + 1] = Tgt2{(u32)(b0 >> 32), (u32)(b1 >> 32)};
        }
    }
    if (foreign) bad = 1;
    __syncthreads();
    const unsigned long long seen = __ballot(mark[tid] != 0);
    __shared__ int cnt[4];
    if (lane == 0) cnt[wave] = __popcll(seen);
    __syncthreads();
    if (tid == 0) {
        a.flags[u] = bad;
        if (a.alphaOut) a.alphaOut[u] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    }
}

#endif

// ------------------------------------------------------------------ the scan
template <int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W <= 16 ? 4 : (W <= 24 ? 3 : 2))))
lanepair_scan_kernel(const ScanArgs a)
{
    const int lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    const bool have = idx < a.numUnits;
    LaneUnit un = a.units[have ? idx : a.numUnits - 1];
    // the unit's own threshold (rate < 0: kmax for every unit)
    const int Kl = a.rate < 0.0f ? (a.kmax < a.kcap ? a.kmax : a.kcap) : unit_threshold(un.m, un.T, a.rate, a.kcap, a.kmax);
    int need = have ? lp_band_words(un.m, un.T, Kl) : 0;
    if (need > W || (a.flags && have && a.flags[idx])) need = 0;
    const bool live = need > 0;
    int m = un.m, T = un.T, K = Kl;
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
        if (live) score = got <= Kl ? got : (Kl >= a.kcap ? kAboveFinal : kAboveOpen);
        if (a.wordSteps && lane == 0) atomicAdd(a.wordSteps, (unsigned long long)ws * 32ull * 64ull);
    }
    if (have) a.outScore[idx] = score;
}

#if !defined(LANEPAIR_NO_PACK)
inline hipError_t launch_pack(const PackArgs& a, hipStream_t s)
{
    if (a.numUnits <= 0) return hipSuccess;
    hipLaunchKernelGGL(lanepair_pack_kernel, dim3(a.numUnits), dim3(256), 0, s, a);
    return hipGetLastError();
}
#endif
// LANEPAIR_WINDOWS: which instantiations this translation unit carries (bit 0: 16, 1: 24, 2: 42, 3: 48)
#if !defined(LANEPAIR_WINDOWS)
#define LANEPAIR_WINDOWS 15
#endif
inline hipError_t launch_scan(const ScanArgs& a, int W, hipStream_t s)
{
    if (a.numUnits <= 0) return hipSuccess;
    const dim3 grid((a.numUnits + 63) / 64), block(64);
#if LANEPAIR_WINDOWS & 1
    if (W == 16) { hipLaunchKernelGGL(lanepair_scan_kernel<16>, grid, block, 0, s, a); return hipGetLastError(); }
#endif
#if LANEPAIR_WINDOWS & 2
    if (W == 24) { hipLaunchKernelGGL(lanepair_scan_kernel<24>, grid, block, 0, s, a); return hipGetLastError(); }
#endif
#if LANEPAIR_WINDOWS & 4
    if (W == 42) { hipLaunchKernelGGL(lanepair_scan_kernel<42>, grid, block, 0, s, a); return hipGetLastError(); }
#endif
#if LANEPAIR_WINDOWS & 8
    if (W == 48) { hipLaunchKernelGGL(lanepair_scan_kernel<48>, grid, block, 0, s, a); return hipGetLastError(); }
#endif
    return hipErrorInvalidValue;
}

}  // namespace lanepair
}  // namespace edlib_amd
