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
// One workgroup per unit; waves 0 / 1 take the query, 2 / 3 the target, 1024 bytes per wave and trip: a lane loads 16 bytes
// (one dwordx4), looks the symbol code of each up (LDS: 256 entries) and builds 16 bits of each plane; two neighbours make a
// word.  (The first build loaded a byte per lane and let two ballots be the planes: 0.96 TB/s.)  Query codes: the target
// symbol the byte equals (eqtbl: 16-bit set of target symbols per byte, EqualityDefinition edlib.cpp:63-94); a byte that
// equals none takes a code no target symbol has when there are fewer than four, and a byte that equals none of four symbols,
// or more than one, marks the unit as not for this kernel (flags[unit] = 1: it stays on the rings).  alphaOut (optional): the
// unit's alphabetLength (distinct bytes of query and target, edlib.cpp:162) -- every byte of both is in a register here
// anyway: each marks its entry of a 256-entry table in LDS, as alphabet_count_kernel does.
#if !defined(LANEPAIR_NO_PACK)
template <bool QUERY>
__device__ __forceinline__ bool lanepair_pack_sequence(const uint8_t* __restrict__ seq, const int len, const uint8_t* code, uint8_t* mark,
                                                       const bool doMark, uint32_t* __restrict__ out, const int lane, const int sub)
{
    bool foreign = false;
    for (int base = sub * 1024; base < len; base += 2048) {
        const int r0 = base + 16 * lane;
        uint32_t w[4] = {0, 0, 0, 0};
        if (r0 < len) __builtin_memcpy(w, seq + r0, 16);                 // (the pools are padded: 16 bytes past a sequence's end exist)
        uint32_t b0 = 0, b1 = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 255u;
            if (r0 + k < len) {
                if (doMark) mark[b] = 1;
                uint32_t c = code[b];
                if (QUERY && c == 255u) { foreign = true; c = 0; }
                b0 |= (c & 1u) << k; b1 |= ((c >> 1) & 1u) << k;
            }
        }
        const uint32_t mine = b0 | (b1 << 16);
        const uint32_t other = (uint32_t)__shfl_xor((int)mine, 1);
        if (!(lane & 1) && r0 < len) {
            uint2 v;
            v.x = (mine & 0xffffu) | (other << 16);                      // plane 0: my 16 rows below my neighbour's
            v.y = (mine >> 16) | (other & 0xffff0000u);                  // plane 1
            *reinterpret_cast<uint2*>(out + 2 * (r0 >> 5)) = v;
        }
    }
    return foreign;
}

__global__ void __launch_bounds__(256)
lanepair_pack_kernel(const PackArgs a)
{
    __shared__ uint8_t qcode[256], tcode[256], mark[256];
    __shared__ int bad, cnt[4];
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
    bool foreign = false;
    if (a.perUnit) {
        // the unit's own alphabet: mark the bytes of its target, rank them (byte order), code = rank; a query byte outside
        // takes the free code when there is one.  More than four: the unit stays on the rings.
        __shared__ uint8_t tmark[256];
        tmark[tid] = 0;
        __syncthreads();
        const uint8_t* t = a.tpool + un.toff;
        for (int r0 = 16 * tid; r0 < un.T; r0 += 16 * 256) {
            uint32_t w[4];
            __builtin_memcpy(w, t + r0, 16);
#pragma unroll
            for (int k = 0; k < 16; ++k) if (r0 + k < un.T) tmark[(w[k >> 2] >> (8 * (k & 3))) & 255u] = 1;
        }
        __syncthreads();
        const bool mk = tmark[tid] != 0;
        const unsigned long long bal = __ballot(mk);
        if (lane == 0) cnt[wave] = __popcll(bal);
        __syncthreads();
        int rank = __popcll(bal & ((1ull << lane) - 1ull));
        for (int w2 = 0; w2 < wave; ++w2) rank += cnt[w2];
        const int total = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        __syncthreads();                                   // (cnt is used again below)
        tcode[tid] = (uint8_t)(mk ? (rank & 3) : 0);
        qcode[tid] = (uint8_t)(mk ? (rank & 3) : (total < 4 ? 3 : 255));
        if (total > 4) foreign = true;
        __syncthreads();
    }
    if (wave < 2) foreign = lanepair_pack_sequence<true>(a.qpool + un.qoff, un.m, qcode, mark, a.alphaOut != nullptr,
                                                         reinterpret_cast<uint32_t*>(a.planes + un.planeOff), lane, wave);
    else lanepair_pack_sequence<false>(a.tpool + un.toff, un.T, tcode, mark, a.alphaOut != nullptr,
                                       reinterpret_cast<uint32_t*>(a.tgts + un.tgtOff), lane, wave - 2);
    if (foreign) bad = 1;
    __syncthreads();
    const unsigned long long seen = __ballot(mark[tid] != 0);
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
    if (!live) { m = 1; T = 0; K = 0; }                          // never active: reads its unit's first words, writes nothing
    const int naWave = lp_wave_max(need < 3 ? (live ? 3 : 0) : need);
    const int nblkWave = lp_wave_max(live ? (T + 31) / 32 : 0);
    int score = kNoBand;
    if (naWave > 0) {
        int ws = 0;
        unsigned deny = a.denySeed;
        const int got = lp_scan<W>(a.planes, (u32)un.planeOff, a.tgts, (u32)un.tgtOff, m, T, K, naWave, nblkWave, deny, &ws);
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
