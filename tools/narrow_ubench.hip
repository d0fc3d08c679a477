// narrow_ubench.hip -- what bounds the ONE-word band of the reads kernel?  Times the 10-op Myers column on
// one 32-row word with different ways of fetching the Peq row of the column's symbol, at 7 waves per SIMD
// on all CUs:
//   0  no fetch at all (fixed row): the VALU floor
//   1  s_mov m0 + s_nop + ds_read_addtid_b32, one column ahead     (what scan_reads_banded_kernel does)
//   2  v_lshl_add_u32 address + ds_read_b32, one column ahead      (1 half-rate VALU instead of 4 SALU)
//   3  as 2, four columns ahead (a quad of requests in flight)
//   4  pair table: one ds_read_b64 per two columns, indexed by two symbols
//   5  as 0 plus 6 scalar ALU ops per column                        (is scalar issue the limit?)
//   hipcc --offload-arch=gfx950 -O3 tools/narrow_ubench.hip -o build/narrow_ubench && build/narrow_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32;

#define XOR_OR(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xde)
#define OR_NOR(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xf1)

__device__ __forceinline__ void col1(const u32 eq, u32& Pv, u32& Mv)
{
    const u32 s = (eq & Pv) + Pv;
    const u32 Xh = XOR_OR(s, eq, Pv);
    const u32 Ph = OR_NOR(Mv, Xh, Pv);
    const u32 Mh = Pv & Xh;
    u32 ph, mh;
    asm("v_add_u32 %0, %1, %1" : "=v"(ph) : "v"(Ph));
    asm("v_add_u32 %0, %1, %1" : "=v"(mh) : "v"(Mh));
    const u32 Xv = eq | Mv;
    Pv = OR_NOR(mh, Xv, ph);
    Mv = ph & Xv;
}

// PADKB: extra LDS per workgroup (KB), to run a variant at the occupancy the real kernel would have with its
// other LDS tables (160 KB per CU: 52 KB per workgroup = 3 workgroups = 3 waves per SIMD)
template <int KIND, int PADKB = 0>
__global__ void __launch_bounds__(256) k_narrow(u32* out, const u32* __restrict__ tpk, int nwords, u32 seed)
{
    __shared__ u32 s_eq[4][4][64];            // [wave][sym][lane]
    __shared__ unsigned long long s_pair[4][16][64];
    __shared__ u32 s_pad[PADKB > 0 ? PADKB * 256 : 1];
    if (PADKB > 0 && seed == 0xdeadbeefu) s_pad[threadIdx.x] = seed;   // keeps the padding allocated
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 E[4];
    for (int i = 0; i < 4; ++i) { E[i] = seed * (threadIdx.x + 3) * (2654435761u + 40503u * i); s_eq[wv][i][lane] = E[i]; }
    for (int p = 0; p < 16; ++p) s_pair[wv][p][lane] = ((unsigned long long)E[p >> 2] << 32) | E[p & 3];
    __syncthreads();
    const u32 ldsBase = __builtin_amdgcn_readfirstlane((u32)(size_t)(__attribute__((address_space(3))) u32*)&s_eq[wv][0][0]);
    const u32 vbase = ldsBase + 4 * lane;
    const u32 pbase = (u32)(size_t)(__attribute__((address_space(3))) unsigned long long*)&s_pair[wv][0][lane];
    u32 Pv = ~0u, Mv = 0u, acc = 0;
    for (int w = 0; w < nwords; ++w) {
        const u32 tw = tpk[w];
        if (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) col1(E[0], Pv, Mv);
        } else if (KIND == 5) {
            u32 sacc = tw;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("s_lshr_b32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_and_b32 %0, %0, 0xffff\n\ts_xor_b32 %0, %0, 5\n\ts_lshl_b32 %0, %0, 1\n\ts_add_u32 %0, %0, 7" : "+s"(sacc));
                col1(E[0], Pv, Mv);
            }
            acc ^= sacc;
        } else if (KIND == 1) {
            u32 n0;
            { const u32 m0v = ldsBase + ((tw & 3u) << 8);
              asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0 offset:0" : "=v"(n0) : "s"(m0v) : "memory"); }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0));
                const u32 e0 = n0;
                if (j < 15) { const u32 m0v = ldsBase + (((tw >> (2 * (j + 1))) & 3u) << 8);
                  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0 offset:0" : "=v"(n0) : "s"(m0v) : "memory"); }
                col1(e0, Pv, Mv);
            }
        } else if (KIND == 2) {
            u32 n0;
            { u32 ad; const u32 sy = tw & 3u; asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(ad) : "s"(sy), "v"(vbase));
              asm volatile("ds_read_b32 %0, %1" : "=v"(n0) : "v"(ad) : "memory"); }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0));
                const u32 e0 = n0;
                if (j < 15) { u32 ad; const u32 sy = (tw >> (2 * (j + 1))) & 3u;
                  asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(ad) : "s"(sy), "v"(vbase));
                  asm volatile("ds_read_b32 %0, %1" : "=v"(n0) : "v"(ad) : "memory"); }
                col1(e0, Pv, Mv);
            }
        } else if (KIND == 3) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                u32 r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { u32 ad; const u32 sy = (tw >> (8 * qd + 2 * j)) & 3u;
                    asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(ad) : "s"(sy), "v"(vbase));
                    asm volatile("ds_read_b32 %0, %1" : "=v"(r[j]) : "v"(ad) : "memory"); }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
#pragma unroll
                for (int j = 0; j < 4; ++j) col1(r[j], Pv, Mv);
            }
        } else if (KIND == 4) {
            unsigned long long n;
            { u32 ad; const u32 sy = tw & 15u; asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(ad) : "s"(sy), "v"(pbase));
              asm volatile("ds_read_b64 %0, %1" : "=v"(n) : "v"(ad) : "memory"); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n));
                const unsigned long long e = n;
                if (j < 7) { u32 ad; const u32 sy = (tw >> (4 * (j + 1))) & 15u;
                  asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(ad) : "s"(sy), "v"(pbase));
                  asm volatile("ds_read_b64 %0, %1" : "=v"(n) : "v"(ad) : "memory"); }
                col1((u32)e, Pv, Mv); col1((u32)(e >> 32), Pv, Mv);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Pv ^ Mv ^ acc ^ (PADKB > 0 ? s_pad[(threadIdx.x * 7) & 255] & (seed == 0xdeadbeefu) : 0u);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F> static float time_ms(F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main()
{
    const int wps = 7, blocks = 256 * wps, nwords = 1 << 15;       // 512k columns
    u32* out; CK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(u32)));
    std::vector<u32> h(nwords); u32 x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    u32* tpk; CK(hipMalloc(&tpk, nwords * 4)); CK(hipMemcpy(tpk, h.data(), nwords * 4, hipMemcpyHostToDevice));
    const char* names[6] = {"fixed row (VALU floor)", "m0 + ds_read_addtid, 1 ahead", "v_lshl_add + ds_read_b32, 1 ahead",
                            "v_lshl_add + ds_read_b32, quad at once", "pair table ds_read_b64 / 2 cols", "fixed row + 6 SALU per column"};
    const double cols = (double)nwords * 16;
#define RUN(K) { float ms = time_ms([&] { hipLaunchKernelGGL(k_narrow<K>, dim3(blocks), dim3(256), 0, 0, out, tpk, nwords, 7u); }); \
                 printf("%-42s %8.3f ms  %6.2f ns per wave-column per SIMD  (%5.1f cycles at 2.1 GHz)\n", names[K], ms, ms * 1e6 / (cols * wps), ms * 1e6 / (cols * wps) * 2.1); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    // occupancy: the M0 fetch with 20 KB per workgroup (7-8 waves per SIMD, as in the kernel) against the pair
    // table with 52 KB per workgroup (3 waves per SIMD: what its 8 KB per wave would leave next to the rows)
#define RUNP(K, PAD, WPS, LABEL) { const int bl = 256 * WPS; float ms = time_ms([&] { hipLaunchKernelGGL((k_narrow<K, PAD>), dim3(bl), dim3(256), 0, 0, out, tpk, nwords, 7u); }); \
                 printf("%-42s %8.3f ms  %6.2f ns per wave-column per SIMD  (%5.1f cycles at 2.1 GHz)\n", LABEL, ms, ms * 1e6 / (cols * WPS), ms * 1e6 / (cols * WPS) * 2.1); }
    RUNP(1, 0, 8, "m0 fetch, 8 waves per SIMD")
    RUNP(4, 15, 3, "pair table, 52 KB LDS: 3 waves per SIMD")
    RUNP(4, 15, 6, "pair table, 52 KB LDS, 6 waves queued")
    RUNP(4, 0, 4, "pair table, 36 KB LDS: 4 waves per SIMD")
    return 0;
}
