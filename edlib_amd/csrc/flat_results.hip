// flat_results.hip -- the results of a flat pair batch laid out on the DEVICE in the caller-facing flat form
// (edlib_amd.h: EdlibAmdResultsView), and edlibAlignmentToCigar over a whole batch.
//
// What the reference does per call after its scans (edlib.cpp:219-299: the -1 rule of the padded last block, k as a
// filter, start locations 0 for NW / SHW, the all-insert path of an empty window) and per op string
// (edlibAlignmentToCigar, :303-350) was host code over per-unit records here until round 4: a run of 262,144 short
// pairs took 1.7 ms and its collection 10 ms.  Now a flat batch is finalised where its scan results are:
//   (1) flat_counts_kernel   one thread per unit: edit distance after the user's k, number of locations, op-string
//                            length; exclusive sums of the two counts inside each workgroup + the workgroup's totals;
//   (2) flat_block_offsets_kernel   one workgroup scans the workgroups' totals (and leaves the grand totals);
//   (3) flat_write_kernel    one wave per unit: location offsets, end / start locations and op bytes at their final,
//                            dense place;
// and ONE block goes to pinned host memory.  CIGAR strings are made from the dense op bytes the same way (lengths, scan,
// write): a wave per op string, 64 ops per trip, runs found by a ballot.
#include "flat_results.hpp"

namespace edlib_amd {

typedef unsigned long long u64;
typedef uint32_t u32;

// ------------------------------------------------------------------ (1) counts

// inclusive -> exclusive sum of `v` over the 256 threads of a workgroup; returns the exclusive prefix, *total = the sum
__device__ __forceinline__ long long block_exclusive_256(long long v, long long* s_part /*[4]*/, long long* total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const long long o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    long long before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const long long p = s_part[q]; if (q < wv) before += p; all += p; }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

__global__ void __launch_bounds__(256)
flat_counts_kernel(const FlatResultArgs a)
{
    __shared__ long long s_part[4];
    const int u = blockIdx.x * 256 + threadIdx.x;
    long long nloc = 0, alen = 0;
    if (u < a.n) {
        const int m = a.descs ? a.descs[u].qlen : a.qlens[u];
        const int score = a.score[u];
        int ed = -1;
        if (a.mode == 0) {                                           // NW (edlib.cpp:744-747, 917; end location T - 1: :221-225)
            if (!(a.k >= 0 && score > a.k)) { ed = score; nloc = 1; }
        } else {
            // SHW / HW: the empty prefix (position -1, score m) takes part exactly when the reference's padded last block
            // sees it, W = 64 ceil(m / 64) - m > 0 (:661, 670, 681-693)
            const int W = ((m + 63) / 64) * 64 - m;
            const bool kAllowsM = a.k < 0 || a.k >= m;
            if (score < 0) { if (W > 0 && kAllowsM) { ed = m; nloc = 1; } }
            else { ed = score; const int c = a.count[u]; nloc = (c > 0 ? c : 0) + ((W > 0 && score == m) ? 1 : 0); }
        }
        a.editDistance[u] = ed;
        a.numLocations[u] = (int)nloc;
        a.status[u] = 0;
        if (a.alphabet) a.alphabetLength[u] = a.alphabet[u] + a.alphaBase;
        if (a.wantPath && ed >= 0 && nloc > 0) {
            // the path of the FIRST location (:276-289); an empty window -- the first location is the empty prefix -- is m inserts (:1168-1175)
            const int W = ((m + 63) / 64) * 64 - m;
            const bool lead = a.mode != 0 && W > 0 && ed == m;
            alen = lead ? m : a.opsLen[u];
        }
        a.alnLen[u] = (int)alen;
    }
    long long tl = 0, ta = 0;
    const long long pl = block_exclusive_256(nloc, s_part, &tl);
    const long long pa = block_exclusive_256(alen, s_part, &ta);
    if (u < a.n) { a.locOff[u] = pl; a.alnOff[u] = pa; }             // (relative to the workgroup until flat_write_kernel adds its base)
    if (threadIdx.x == 0) { a.blockLoc[blockIdx.x] = tl; a.blockAln[blockIdx.x] = ta; }
}

// ------------------------------------------------------------------ (2) workgroup bases

// exclusive scan of the workgroups' totals, in place, by ONE workgroup; totals[0..1] = the grand totals
__global__ void __launch_bounds__(256)
flat_block_offsets_kernel(long long* __restrict__ blockLoc, long long* __restrict__ blockAln /* may be null */, const int nblocks, long long* __restrict__ totals)
{
    __shared__ long long s_part[4];
    long long baseL = 0, baseA = 0;
    for (int i0 = 0; i0 < nblocks; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const long long vl = i < nblocks ? blockLoc[i] : 0, va = (i < nblocks && blockAln) ? blockAln[i] : 0;
        long long tl = 0, ta = 0;
        const long long pl = block_exclusive_256(vl, s_part, &tl);
        const long long pa = block_exclusive_256(va, s_part, &ta);
        if (i < nblocks) { blockLoc[i] = baseL + pl; if (blockAln) blockAln[i] = baseA + pa; }
        baseL += tl; baseA += ta;
    }
    if (threadIdx.x == 0) { totals[0] = baseL; totals[1] = baseA; }
}

// ------------------------------------------------------------------ (3) the dense arrays

__global__ void __launch_bounds__(256)
flat_write_kernel(const FlatResultArgs a)
{
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= a.n) return;
    const long long lo = a.blockLoc[u >> 8] + a.locOff[u], ao = a.blockAln[u >> 8] + a.alnOff[u];
    // (every lane has read its unit's relative offsets before any lane of the workgroup overwrites them: the four units of a
    // workgroup are distinct entries)
    const int nloc = a.numLocations[u], alen = a.alnLen[u], ed = a.editDistance[u];
    if (lane == 0) {
        a.locOff[u] = lo; a.alnOff[u] = ao;
        if (u == a.n - 1) { a.locOff[a.n] = lo + nloc; a.alnOff[a.n] = ao + alen; }
    }
    const int m = a.descs ? a.descs[u].qlen : a.qlens[u];
    if (nloc > 0) {
        if (a.mode == 0) {
            if (lane == 0) { a.ends[lo] = (a.descs ? a.descs[u].tlen : a.sharedT) - 1; if (a.starts) a.starts[lo] = 0; }
        } else {
            const int W = ((m + 63) / 64) * 64 - m;
            const int lead = (W > 0 && ed == m) ? 1 : 0;             // position -1 comes first
            if (lead && lane == 0) { a.ends[lo] = -1; if (a.starts) a.starts[lo] = 0; }
            const int c = nloc - lead;
            // the unit's list: 16 positions beside its results, or -- a unit of the exact second pass -- its range of the overflow pool
            const int ov = a.ovfAt ? a.ovfAt[u] : -1;
            const int* src = ov >= 0 ? a.ovfPos + a.ovfOff[ov] : a.pos + (long long)u * a.posCap;
            for (int i = lane; i < c; i += 64) {
                a.ends[lo + lead + i] = src[i];
                if (a.starts) a.starts[lo + lead + i] = (a.devStarts && i < a.posCap) ? a.devStarts[(long long)u * a.posCap + i] : 0;
            }
        }
    }
    if (alen > 0) {
        uint8_t* dst = a.aln + ao;
        const int W = ((m + 63) / 64) * 64 - m;
        const bool lead = a.mode != 0 && W > 0 && ed == m;
        if (lead) { for (int i = lane; i < alen; i += 64) dst[i] = 1; }        // EDLIB_EDOP_INSERT
        else {
            const uint8_t* src = a.ops + a.opsOff[u + 1] - alen;      // the ops sit at the END of the unit's slot
            for (int i = lane; i < alen; i += 64) dst[i] = src[i];
        }
    }
}

hipError_t launch_flat_results(const FlatResultArgs& a, long long* totals, hipStream_t stream)
{
    if (a.n <= 0) return hipSuccess;
    const int nblocks = (a.n + 255) / 256;
    hipLaunchKernelGGL(flat_counts_kernel, dim3(nblocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(flat_block_offsets_kernel, dim3(1), dim3(256), 0, stream, a.blockLoc, a.blockAln, nblocks, totals);
    hipLaunchKernelGGL(flat_write_kernel, dim3((a.n + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ CIGAR

// reference edlibAlignmentToCigar (edlib.cpp:303-350) for every op string of a batch: run-length encoding with the letters
// "=IDX" (extended) or "MIDM" (standard: MATCH and MISMATCH are one letter, so their runs merge).  One wave per op string,
// 64 ops per trip: a lane whose op starts a run closes the run BEFORE it (length = its position - the previous start,
// which is the nearest lower set bit of the ballot or, failing that, the start carried over from earlier trips); the
// last run is closed behind the loop.  WRITE = false: only the length (digits + letter per run, + 1 for the NUL).
__device__ __forceinline__ int dec_digits(u32 v) {
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7
         : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ void put_run(char* out, u32 len, const int digits, const char letter) {
    for (int i = digits - 1; i >= 0; --i) { out[i] = (char)('0' + len % 10u); len /= 10u; }
    out[digits] = letter;
}

template <bool WRITE>
__global__ void __launch_bounds__(256)
cigar_kernel(const uint8_t* __restrict__ aln, const long long* __restrict__ alnOff, const int n, const int standard,
             long long* __restrict__ cigLen, const long long* __restrict__ blockBase, const long long* __restrict__ cigRel,
             char* __restrict__ out, long long* __restrict__ cigOffOut)
{
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n) return;
    const long long a0 = alnOff[u];
    const int len = (int)(alnOff[u + 1] - a0);
    const uint8_t* ops = aln + a0;
    const char letters[4] = {standard ? 'M' : '=', 'I', 'D', standard ? 'M' : 'X'};
    long long outAt = 0;
    if (WRITE) { outAt = blockBase[u >> 8] + cigRel[u]; if (lane == 0) cigOffOut[u] = outAt; }
    int openStart = 0;                                               // start of the run that is open at the top of a trip
    int prevCls = -1;                                                // class of the op before this trip's first
    int total = 0;                                                   // characters so far (uniform)
    int openCls = 0;
    for (int i0 = 0; i0 < len; i0 += 64) {
        const int i = i0 + lane;
        const int op = i < len ? (int)ops[i] : 0;
        const int cls = standard ? (op == 3 ? 0 : op) : op;
        int before = __shfl_up(cls, 1, 64);
        if (lane == 0) before = prevCls;
        const bool start = i < len && (i == 0 || cls != before);
        const u64 mask = __builtin_amdgcn_ballot_w64(start);
        // a start at i > 0 closes the run [ps, i): ps = the nearest start below it in this trip, else the carried one
        const bool closes = start && i > 0;
        const u64 lower = mask & ((1ull << lane) - 1ull);
        const int ps = lower ? i0 + (63 - __builtin_clzll(lower)) : openStart;
        const u32 rl = closes ? (u32)(i - ps) : 0u;
        const int dg = closes ? dec_digits(rl) : 0;
        const int chars = closes ? dg + 1 : 0;
        int incl = chars;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (WRITE && closes) {
            // the letter of the closed run is the class just before this op
            const int c = before;
            put_run(out + outAt + total + incl - chars, rl, dg, letters[c]);
        }
        total += __shfl(incl, 63, 64);
        if (mask) { openStart = i0 + (63 - __builtin_clzll(mask)); }
        const int lastLane = (len - i0 < 64 ? len - i0 : 64) - 1;
        prevCls = __shfl(cls, lastLane, 64);
        openCls = prevCls;
    }
    if (len > 0) {                                                   // the run that is still open
        const u32 rl = (u32)(len - openStart);
        const int dg = dec_digits(rl);
        if (WRITE && lane == 0) put_run(out + outAt + total, rl, dg, letters[openCls]);
        total += dg + 1;
    }
    if (WRITE) { if (lane == 0) { out[outAt + total] = '\0'; if (u == n - 1) cigOffOut[n] = outAt + total + 1; } }
    else if (lane == 0) cigLen[u] = total + 1;
}

// cigLen[u] -> exclusive sums inside each workgroup of 256 units (cigRel) + the workgroups' totals
__global__ void __launch_bounds__(256)
cigar_block_sums_kernel(const long long* __restrict__ cigLen, const int n, long long* __restrict__ cigRel, long long* __restrict__ blockTot)
{
    __shared__ long long s_part[4];
    const int u = blockIdx.x * 256 + threadIdx.x;
    long long t = 0;
    const long long p = block_exclusive_256(u < n ? cigLen[u] : 0, s_part, &t);
    if (u < n) cigRel[u] = p;
    if (threadIdx.x == 0) blockTot[blockIdx.x] = t;
}

hipError_t launch_cigars(const uint8_t* aln, const long long* alnOff, int n, int standard, long long* cigLen, long long* cigRel,
                         long long* blockTot, long long* totals, char* out, long long* cigOff, int phase, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int nblocks = (n + 255) / 256;
    if (phase == 0) {                                                // lengths, their sums, the grand total (totals[0])
        hipLaunchKernelGGL((cigar_kernel<false>), dim3((n + 3) / 4), dim3(256), 0, stream, aln, alnOff, n, standard, cigLen,
                           (const long long*)nullptr, (const long long*)nullptr, (char*)nullptr, (long long*)nullptr);
        hipLaunchKernelGGL(cigar_block_sums_kernel, dim3(nblocks), dim3(256), 0, stream, cigLen, n, cigRel, blockTot);
        hipLaunchKernelGGL(flat_block_offsets_kernel, dim3(1), dim3(256), 0, stream, blockTot, (long long*)nullptr, nblocks, totals);
    } else {                                                         // the strings (the caller sized `out` from totals[0])
        hipLaunchKernelGGL((cigar_kernel<true>), dim3((n + 3) / 4), dim3(256), 0, stream, aln, alnOff, n, standard, (long long*)nullptr,
                           blockTot, cigRel, out, cigOff);
    }
    return hipGetLastError();
}

}  // namespace edlib_amd
