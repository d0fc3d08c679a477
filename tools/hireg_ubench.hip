// hireg_ubench.hip -- does a wave that owns 256 VGPRs issue VALU at the same rate as one that owns 24?  The scan word of the
// lane-per-pair kernel (all VOP3 encodings) on low registers in a small kernel and on v200.. in a kernel that allocates 256.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define WORD(a, b, c, d, e, f, g, h, i, j, k, l, X, Y) \
    "v_xor_b32_e64 " a ", " a ", " X "\n v_bitop3_b32 " b ", " b ", " X ", " Y " bitop3:0x96\n v_alignbit_b32 " c ", " c ", " X ", 31\n v_bitop3_b32 " d ", " d ", " X ", " Y " bitop3:0x96\n" \
    "v_xor_b32_e64 " e ", " e ", " X "\n v_alignbit_b32 " f ", " f ", " X ", 31\n v_bitop3_b32 " g ", " g ", " X ", " Y " bitop3:0x96\n v_bitop3_b32 " h ", " h ", " X ", " Y " bitop3:0x96\n" \
    "v_bitop3_b32 " i ", " i ", " X ", " Y " bitop3:0x96\n v_addc_co_u32_e64 " j ", s[20:21], " j ", " X ", s[20:21]\n v_bitop3_b32 " k ", " k ", " X ", " Y " bitop3:0x96\n v_xor_b32_e64 " l ", " l ", " X "\n"
template <int HI>
__global__ void __launch_bounds__(64) k(int iters, unsigned* out, unsigned long long* clk, unsigned seed)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    if constexpr (HI == 3 || HI == 4) {
        // the same word on low registers, every register initialised: HI == 3 zeros / small constants, HI == 4 random bits per lane
        unsigned z = HI == 4 ? (threadIdx.x * 2654435761u + seed) : 0u;
        asm volatile("s_mov_b64 s[20:21], 0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n"
                     "v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n v_mov_b32 v12, %0\n v_mov_b32 v13, %1\n v_mov_b32 v14, %0\n v_mov_b32 v15, %1\n"
                     "v_mov_b32 v16, %0\n v_mov_b32 v17, %1\n v_mov_b32 v18, %0\n v_mov_b32 v19, %1\n v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n"
                     :: "v"(HI == 4 ? z * 0x9e3779b1u + 0x7f4a7c15u : 1u), "v"(HI == 4 ? (z ^ 0x5bd1e995u) * 0x85ebca6bu : 2u)
                     : "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "s20", "s21");
        for (int it = 0; it < iters; ++it)
            asm volatile(".rept 32\n" WORD("v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v2", "v3") ".endr\n" ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "s20", "s21");
        unsigned r; asm volatile("v_mov_b32 %0, v10" : "=v"(r)); if (r == 0x12345678u) out[threadIdx.x] = r;
    } else
    if constexpr (HI == 0) {
        asm volatile("s_mov_b64 s[20:21], 0\n v_mov_b32 v2, 1\n v_mov_b32 v3, 2\n" ::: "v2", "v3", "s20", "s21");
        for (int it = 0; it < iters; ++it)
            asm volatile(".rept 32\n" WORD("v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v2", "v3") ".endr\n" ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "s20", "s21");
        unsigned r; asm volatile("v_mov_b32 %0, v10" : "=v"(r)); if (r == 0x12345678u) out[threadIdx.x] = r;
    } else if constexpr (HI == 1) {
        asm volatile("s_mov_b64 s[20:21], 0\n v_mov_b32 v2, 1\n v_mov_b32 v3, 2\n" ::: "v2", "v3", "s20", "s21", "v255");
        for (int it = 0; it < iters; ++it)
            asm volatile(".rept 32\n" WORD("v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v2", "v3") ".endr\n" ::: "v2", "v3", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "s20", "s21", "v255");
        unsigned r; asm volatile("v_mov_b32 %0, v210" : "=v"(r)); if (r == 0x12345678u) out[threadIdx.x] = r;
    } else {
        // operands spread over the file like the compiled scan: sources from three different ranges
        asm volatile("s_mov_b64 s[20:21], 0\n v_mov_b32 v130, 1\n v_mov_b32 v67, 2\n" ::: "v130", "v67", "s20", "s21", "v255");
        for (int it = 0; it < iters; ++it)
            asm volatile(".rept 32\n" WORD("v14", "v111", "v225", "v18", "v146", "v34", "v251", "v117", "v150", "v201", "v99", "v6", "v130", "v67") ".endr\n" ::: "v130", "v67", "v14", "v111", "v225", "v18", "v146", "v34", "v251", "v117", "v150", "v201", "v99", "v6", "s20", "s21", "v255");
        unsigned r; asm volatile("v_mov_b32 %0, v14" : "=v"(r)); if (r == 0x12345678u) out[threadIdx.x] = r;
    }
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
template <int HI> static void run(const char* what, int wps, unsigned* d)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 8000 / wps;
    unsigned long long* clk = (unsigned long long*)(d + 64);
    hipLaunchKernelGGL((k<HI>), dim3(1024 * wps), dim3(64), 0, 0, 1, d, clk, 1u);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<HI>), dim3(1024 * wps), dim3(64), 0, 0, iters, d, clk, 12345u);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * 12 * 32;
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);                 // s_memtime ticks per microsecond of the 100 MHz counter
    printf("{\"regs\": \"%s\", \"waves_per_simd\": %d, \"simd_ns_per_instr\": %.4f, \"simd_cycles_per_instr_at_2.1GHz\": %.2f, \"s_memtime_MHz\": %.0f, \"ms\": %.2f}\n", what, wps, ms * 1e6 / n / wps, ms * 1e6 / n / wps * 2.1, mhz, ms);
}
int main()
{
    unsigned* d; CK(hipMalloc(&d, 1024));
    run<3>("v10..v21 all initialised to 1 / 2", 2, d); run<3>("v10..v21 all initialised to 1 / 2", 8, d);
    run<4>("v10..v21 random bits", 2, d); run<4>("v10..v21 random bits", 8, d);
    run<0>("v10..v21, small kernel", 1, d); run<0>("v10..v21, small kernel", 2, d); run<0>("v10..v21, small kernel", 8, d);
    run<1>("v210..v221, 256 VGPRs allocated", 1, d); run<1>("v210..v221, 256 VGPRs allocated", 2, d);
    run<2>("scattered over v6..v251, 256 allocated", 1, d); run<2>("scattered over v6..v251, 256 allocated", 2, d);
    return 0;
}
