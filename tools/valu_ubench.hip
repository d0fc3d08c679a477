// valu_ubench.hip -- measures the issue rate of the integer VALU ops the scan kernels are made of
// (v_and_b32, v_bitop3_b32, v_alignbit_b32, v_add_co/v_addc_co chains) and of the whole Myers
// column body, on all 256 CUs at 1..8 waves per SIMD.  Gives the MEASURED VALU ceiling that
// bench.py's valu_roofline is priced against.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o build/valu_ubench && build/valu_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) k_op(u32* out, int iters, u32 seed)
{
    u32 r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1) + i * 77;
    u32 a = seed ^ threadIdx.x, b = ~seed + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OP(i)                                                                                           \
            if (KIND == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));                   \
            if (KIND == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xde" : "+v"(r[i]) : "v"(a), "v"(b)); \
            if (KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r[i]) : "v"(a));          \
            if (KIND == 3) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));                    \
            if (KIND == 4) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(r[i]) : "v"(a));                \
            if (KIND == 5) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));      \
            if (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a));           \
            if (KIND == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));                   \
            if (KIND == 8) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]));                         \
            if (KIND == 9) asm volatile("v_lshrrev_b32 %0, 31, %0" : "+v"(r[i]));                        \
            if (KIND == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));                  \
            if (KIND == 11) asm volatile("v_min_i32 %0, %0, %1" : "+v"(r[i]) : "v"(a));                  \
            if (KIND == 12) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));             \
            if (KIND == 13) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));      \
            if (KIND == 14) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));   \
            if (KIND == 15) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));      \
            if (KIND == 16) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a));           \
            if (KIND == 17) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(a) : "vcc");  \
            if (KIND == 18) asm volatile("v_cmp_le_i32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");      \
            if (KIND == 19) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(r[i]));                       \
            if (KIND == 20) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a)); \
            if (KIND == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "s"(seed));               \
            if (KIND == 22) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xde" : "+v"(r[i]) : "v"(a), "s"(seed));
            REP8(OP)
#undef OP
        }
    }
    u32 s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 64-bit forms: does one u64 op beat two u32 ops?
template <int KIND>
__global__ void __launch_bounds__(256) k_op64(u32* out, int iters, u32 seed)
{
    unsigned long long r[8];
    for (int i = 0; i < 8; ++i) r[i] = (unsigned long long)seed * (threadIdx.x + 1) + i * 77;
    unsigned long long a = seed ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OP(i)                                                                                           \
            if (KIND == 0) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[i]) : "v"(a));           \
            if (KIND == 1) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(r[i]));                         \
            if (KIND == 2) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a));           \
            if (KIND == 3) asm volatile("v_lshrrev_b64 %0, 31, %0" : "+v"(r[i]));
            REP8(OP)
#undef OP
        }
    }
    unsigned long long s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}

// carry chain: v_add_co + 4 x v_addc_co (the multi-word add of the column step)
__global__ void __launch_bounds__(256) k_addc(u32* out, int iters, u32 seed)
{
    u32 r[10];
    for (int i = 0; i < 10; ++i) r[i] = seed * (threadIdx.x + 1) + i * 77;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_add_co_u32 %0, vcc, %0, %5\n\tv_addc_co_u32 %1, vcc, %1, %6, vcc\n\t"
                         "v_addc_co_u32 %2, vcc, %2, %7, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
                         "v_addc_co_u32 %4, vcc, %4, %9, vcc"
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4])
                         : "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(r[9]) : "vcc");
        }
    }
    u32 s = 0;
    for (int i = 0; i < 10; ++i) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the Myers column body on NWD words with a fixed Eq row (no dispatch): ops/column as in the scan kernel
template <int NWD>
__device__ __forceinline__ void column_step(const u32 (&Eq)[NWD], u32 (&Pv)[NWD], u32 (&Mv)[NWD], int& score, u32 sh)
{
    u32 Ph[NWD], Mh[NWD]; u32 carry = 0;
#pragma unroll
    for (int i = 0; i < NWD; ++i) {
        const u32 t = Eq[i] & Pv[i]; u32 cout;
        const u32 s = __builtin_addc(t, Pv[i], carry, &cout); carry = cout;
        const u32 Xh = (s ^ Pv[i]) | Eq[i];
        Ph[i] = Mv[i] | ~(Xh | Pv[i]); Mh[i] = Pv[i] & Xh;
    }
    score += (int)__builtin_amdgcn_ubfe(Ph[NWD - 1], sh, 1) + __builtin_amdgcn_sbfe(Mh[NWD - 1], sh, 1);
#pragma unroll
    for (int i = NWD - 1; i >= 0; --i) {
        u32 ph, mh;
        if (i > 0) { ph = __builtin_amdgcn_alignbit(Ph[i], Ph[i - 1], 31); mh = __builtin_amdgcn_alignbit(Mh[i], Mh[i - 1], 31); }
        else { ph = Ph[0] << 1; mh = Mh[0] << 1; }
        const u32 Xv = Eq[i] | Mv[i];
        Pv[i] = mh | ~(Xv | ph); Mv[i] = ph & Xv;
    }
}

template <int SWITCH>
__global__ void __launch_bounds__(256) k_body(u32* out, const u32* __restrict__ tpk, int nwords, u32 seed)
{
    constexpr int NWD = 5;
    u32 E0[NWD], E1[NWD], E2[NWD], E3[NWD], Pv[NWD], Mv[NWD];
    for (int i = 0; i < NWD; ++i) {
        E0[i] = seed * (threadIdx.x + 3) + i; E1[i] = E0[i] * 2654435761u; E2[i] = E1[i] * 40503u + 7; E3[i] = ~(E0[i] | E1[i] | E2[i]);
        Pv[i] = ~0u; Mv[i] = 0;
    }
    int score = 150, best = 150, cnt = 0; const u32 sh = 21;
    for (int w = 0; w < nwords; ++w) {
        const u32 tw = tpk[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 sym = (tw >> (2 * j)) & 3u;
            if (SWITCH) {
                switch (sym) {
                    case 0: column_step<NWD>(E0, Pv, Mv, score, sh); asm volatile("; s0"); break;
                    case 1: column_step<NWD>(E1, Pv, Mv, score, sh); asm volatile("; s1"); break;
                    case 2: column_step<NWD>(E2, Pv, Mv, score, sh); asm volatile("; s2"); break;
                    default: column_step<NWD>(E3, Pv, Mv, score, sh); asm volatile("; s3"); break;
                }
            } else {
                column_step<NWD>(E0, Pv, Mv, score, sh);
            }
            if (SWITCH != 2) { if (score <= best) { if (score < best) { best = score; cnt = 0; } ++cnt; } }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Pv[0] ^ Mv[4] ^ score ^ best ^ cnt;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F>
static float time_ms(F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main()
{
    u32* out; CK(hipMalloc(&out, 256 * 4 * 8 * 64 * 4 * sizeof(u32)));
    const int nwords = 1 << 16;            // 1M columns
    std::vector<u32> h(nwords); u32 x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    u32* tpk; CK(hipMalloc(&tpk, nwords * 4)); CK(hipMemcpy(tpk, h.data(), nwords * 4, hipMemcpyHostToDevice));
    const char* names[23] = {"v_and_b32", "v_bitop3_b32", "v_alignbit_b32", "v_or_b32", "v_bfe_u32", "v_add3_u32", "v_lshl_add_u32", "v_xor_b32",
                             "v_lshlrev_b32", "v_lshrrev_b32", "v_add_u32", "v_min_i32", "v_bcnt_u32_b32", "v_bfi_b32", "v_and_or_b32", "v_or3_b32",
                             "v_lshl_or_b32", "v_add_co_u32", "v_cmp_le_i32", "v_ashrrev_i32", "v_mov_b32_dpp", "v_and_b32(sgpr)", "v_bitop3(sgpr)"};
    printf("# ops/clk/CU assume 2.4 GHz; peak model = 4 SIMD x 32 lanes = 128 lane-ops/clk/CU\n");
    for (int wps : {8}) {
        const int blocks = 256 * wps;          // 256-thread blocks: 4 waves = 1 per SIMD
        const int iters = 20000;
        const double laneops = (double)blocks * 256 * iters * 64;
#define RUN(K) { float ms = time_ms([&] { hipLaunchKernelGGL(k_op<K>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u); }); \
                 printf("waves/SIMD %d  %-16s %8.3f ms  %7.2f T lane-ops/s  %6.1f lane-ops/clk/CU@2.4GHz\n", wps, names[K], ms, laneops / ms / 1e9, laneops / (ms * 1e-3) / 2.4e9 / 256); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
        if (wps == 8) { RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) }
        if (wps == 8) {
            const char* n64[4] = {"v_lshl_add_u64(+0)", "v_lshlrev_b64", "v_lshl_add_u64(<<1)", "v_lshrrev_b64"};
#define RUN64(K) { float ms = time_ms([&] { hipLaunchKernelGGL(k_op64<K>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u); }); \
                 printf("waves/SIMD %d  %-20s %8.3f ms  %7.2f T 64-bit lane-ops/s\n", wps, n64[K], ms, laneops / ms / 1e9); }
            RUN64(0) RUN64(1) RUN64(2) RUN64(3)
        }
        { const double lo = (double)blocks * 256 * iters * 8 * 5;
          float ms = time_ms([&] { hipLaunchKernelGGL(k_addc, dim3(blocks), dim3(256), 0, 0, out, iters, 7u); });
          printf("waves/SIMD %d  %-16s %8.3f ms  %7.2f T lane-ops/s  %6.1f lane-ops/clk/CU@2.4GHz\n", wps, "add_co+4addc", ms, lo / ms / 1e9, lo / (ms * 1e-3) / 2.4e9 / 256); }
        { const double cols = (double)nwords * 16; const double waves = (double)blocks * 4;
          float m0 = time_ms([&] { hipLaunchKernelGGL(k_body<0>, dim3(blocks), dim3(256), 0, 0, out, tpk, nwords, 7u); });
          float m1 = time_ms([&] { hipLaunchKernelGGL(k_body<1>, dim3(blocks), dim3(256), 0, 0, out, tpk, nwords, 7u); });
          float m2 = time_ms([&] { hipLaunchKernelGGL(k_body<2>, dim3(blocks), dim3(256), 0, 0, out, tpk, nwords, 7u); });
          printf("waves/SIMD %d  column body: no-dispatch %.3f ms (%.1f ns/col/SIMD)  switch+track %.3f ms (%.1f)  switch no-track %.3f ms (%.1f)   [wave-columns per SIMD-second: %.3g / %.3g]\n",
                 wps, m0, m0 * 1e6 / (cols * wps), m1, m1 * 1e6 / (cols * wps), m2, m2 * 1e6 / (cols * wps), cols * waves / 1024 / (m0 * 1e-3), cols * waves / 1024 / (m1 * 1e-3)); }
    }
    return 0;
}
