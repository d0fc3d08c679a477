// narrow2_ubench.hip -- round-2 candidates for the ONE-word band of scan_reads_banded_kernel (follows
// narrow_ubench.hip, whose variant 1 is the round-1 kernel's inner loop: 29 cycles per column against 22.7 for the
// bare 10-op column).  Every non-VALU instruction of the column costs about a cycle there: 3 SALU for M0, the LDS
// request, the wait, the loop.  The candidates remove them:
//   * one wave per workgroup: the LDS slice starts at 0, so M0 is the row offset itself;
//   * the target expanded to 16 bits per column = row offset (symbol << 8), read 16 columns per s_load_dwordx8:
//     one SALU per column writes M0 (s_pack_ll_b32_b16 / s_lshr_b32), no symbol extraction;
//   * the four rows of the NEXT quad are requested while the current quad computes; one s_waitcnt per quad.
// Variants (all at 8 waves per SIMD, 64-thread workgroups):
//   0  fixed row, no fetch                                   (VALU floor in this harness)
//   1  quad in one hand-scheduled asm block, M0 wait state filled by the column's first VALU op
//   2  as 1 plus the band checkpoint (2 v_bcnt + v_cmp + branch per quad)
//   3  requests of the next quad in one asm block (s_nop wait states), VALU left to the compiler
//   4  as 3 plus the checkpoint
//   5  as 2, checkpoint skipped while the last one left a margin (two thresholds per check)
//   hipcc --offload-arch=gfx950 -O3 tools/narrow2_ubench.hip -o build/narrow2_ubench && build/narrow2_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32;
typedef u32 u32x8 __attribute__((ext_vector_type(8)));

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

// one column on row register C with the request of the next quad's row N folded in (MSET writes M0; the
// v_and after it is the wait state M0 needs before an add-TID LDS instruction)
#define NB_COL(C, MSET, N)                                              \
    MSET "\n\t"                                                         \
    "v_and_b32 %[t], " C ", %[pv]\n\t"                                  \
    "ds_read_addtid_b32 " N " offset:0\n\t"                             \
    "v_add_u32 %[t], %[t], %[pv]\n\t"                                   \
    "v_bitop3_b32 %[x], %[t], " C ", %[pv] bitop3:0xde\n\t"             \
    "v_bitop3_b32 %[p], %[mv], %[x], %[pv] bitop3:0xf1\n\t"             \
    "v_and_b32 %[m], %[pv], %[x]\n\t"                                   \
    "v_add_u32 %[p], %[p], %[p]\n\t"                                    \
    "v_add_u32 %[m], %[m], %[m]\n\t"                                    \
    "v_or_b32 %[x], " C ", %[mv]\n\t"                                   \
    "v_bitop3_b32 %[pv], %[m], %[x], %[p] bitop3:0xf1\n\t"              \
    "v_and_b32 %[mv], %[p], %[x]\n\t"

__device__ __forceinline__ void quad_asm(u32 (&c)[4], u32& Pv, u32& Mv, const u32 nlo, const u32 nhi)
{
    u32 n0, n1, n2, n3, t, x, p, m;
    asm volatile(
        NB_COL("%[c0]", "s_pack_ll_b32_b16 m0, %[lo], 0", "%[n0]")
        NB_COL("%[c1]", "s_lshr_b32 m0, %[lo], 16", "%[n1]")
        NB_COL("%[c2]", "s_pack_ll_b32_b16 m0, %[hi], 0", "%[n2]")
        NB_COL("%[c3]", "s_lshr_b32 m0, %[hi], 16", "%[n3]")
        "s_waitcnt lgkmcnt(0)"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [t] "=&v"(t), [x] "=&v"(x), [p] "=&v"(p), [m] "=&v"(m),
          [pv] "+v"(Pv), [mv] "+v"(Mv)
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [lo] "s"(nlo), [hi] "s"(nhi)
        : "memory", "scc");          // s_lshr_b32 writes SCC (hipcc keeps carries / loop conditions there)
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void quad_request(u32 (&n)[4], const u32 nlo, const u32 nhi)
{
    asm volatile("s_pack_ll_b32_b16 m0, %[lo], 0\n\ts_nop 0\n\tds_read_addtid_b32 %[n0] offset:0\n\t"
                 "s_lshr_b32 m0, %[lo], 16\n\ts_nop 0\n\tds_read_addtid_b32 %[n1] offset:0\n\t"
                 "s_pack_ll_b32_b16 m0, %[hi], 0\n\ts_nop 0\n\tds_read_addtid_b32 %[n2] offset:0\n\t"
                 "s_lshr_b32 m0, %[hi], 16\n\ts_nop 0\n\tds_read_addtid_b32 %[n3] offset:0"
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]) : [lo] "s"(nlo), [hi] "s"(nhi) : "memory", "scc");
}

template <int KIND>
__global__ void __launch_bounds__(64) k_narrow2(u32* out, const u32x8* __restrict__ tx, int nblocks, u32 seed, int kq)
{
    __shared__ __attribute__((aligned(1024))) u32 s_eq[4][64];
    const int lane = threadIdx.x;
    u32 E[4];
    for (int i = 0; i < 4; ++i) { E[i] = seed * (blockIdx.x * 64 + lane + 3) * (2654435761u + 40503u * i); s_eq[i][lane] = E[i]; }
    __syncthreads();
    u32 Pv = ~0u, Mv = 0u, grow = 0;
    u32 c[4] = {E[0], E[1], E[2], E[3]};
    int skip = 0;
    u32x8 cur = tx[0];
    for (int b = 0; b < nblocks; ++b) {
        const u32x8 nxt = tx[b + 1];                        // the buffer holds one block more
        const u32 lo[5] = {cur[0], cur[2], cur[4], cur[6], nxt[0]}, hi[5] = {cur[1], cur[3], cur[5], cur[7], nxt[1]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (KIND == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) col1(E[0], Pv, Mv);
            } else if (KIND == 1 || KIND == 2 || KIND == 5) {
                quad_asm(c, Pv, Mv, lo[q + 1], hi[q + 1]);
            } else {
                u32 n[4];
                quad_request(n, lo[q + 1], hi[q + 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) col1(c[j], Pv, Mv);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]));
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = n[j];
            }
            if (KIND == 2 || KIND == 4) {
                const int up = __popc(Pv), dn = __popc(Mv) + kq;
                if (__builtin_amdgcn_ballot_w64(up <= dn) != 0ull) { Pv = ~0u; Mv = 0; ++grow; }
            }
            if (KIND == 5) {
                if (skip > 0) --skip;
                else {
                    const int up = __popc(Pv), dn = __popc(Mv) + kq;
                    if (__builtin_amdgcn_ballot_w64(up <= dn) != 0ull) { Pv = ~0u; Mv = 0; ++grow; }
                    else if (__builtin_amdgcn_ballot_w64(up <= dn + 4) == 0ull) skip = 1;
                }
            }
        }
        cur = nxt;
    }
    out[blockIdx.x * 64 + lane] = Pv ^ Mv ^ grow;
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
    const int wps = 8, waves = 256 * 4 * wps, nblocks = 1 << 15;       // 512k columns per wave
    u32* out; CK(hipMalloc(&out, (size_t)waves * 64 * sizeof(u32)));
    std::vector<u32> h((size_t)(nblocks + 1) * 8); u32 x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 3u) << 8) | (((x >> 20) & 3u) << 24); }   // two u16 row offsets
    u32x8* tx; CK(hipMalloc(&tx, h.size() * 4)); CK(hipMemcpy(tx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const char* names[6] = {"fixed row (VALU floor)", "asm quad, next quad requested inside", "asm quad + checkpoint",
                            "request block + compiler VALU", "request block + compiler VALU + checkpoint",
                            "asm quad + checkpoint with skip"};
    const double cols = (double)nblocks * 16;
#define RUN(K) { float ms = time_ms([&] { hipLaunchKernelGGL(k_narrow2<K>, dim3(waves), dim3(64), 0, 0, out, tx, nblocks, 7u, 3 + 6 - 32); }); \
                 printf("%-46s %8.3f ms  %6.2f ns per wave-column per SIMD  (%5.1f cycles at 2.1 GHz)\n", names[K], ms, ms * 1e6 / (cols * wps), ms * 1e6 / (cols * wps) * 2.1); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
