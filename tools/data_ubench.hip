// data_ubench.hip -- is the issue rate of the scan word (VOP3 encodings) a function of the DATA in its registers?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define REGS "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "s20", "s21"
#define WORD(a, b, c, d, e, f, g, h, i, j, k, l, X, Y) \
    "v_xor_b32_e64 " a ", " a ", " X "\n v_bitop3_b32 " b ", " b ", " X ", " Y " bitop3:0x96\n v_alignbit_b32 " c ", " c ", " X ", 31\n v_bitop3_b32 " d ", " d ", " X ", " Y " bitop3:0x96\n" \
    "v_xor_b32_e64 " e ", " e ", " X "\n v_alignbit_b32 " f ", " f ", " X ", 31\n v_bitop3_b32 " g ", " g ", " X ", " Y " bitop3:0x96\n v_bitop3_b32 " h ", " h ", " X ", " Y " bitop3:0x96\n" \
    "v_bitop3_b32 " i ", " i ", " X ", " Y " bitop3:0x96\n v_addc_co_u32_e64 " j ", s[20:21], " j ", " X ", s[20:21]\n v_bitop3_b32 " k ", " k ", " X ", " Y " bitop3:0x96\n v_xor_b32_e64 " l ", " l ", " X "\n"
// BODY 0: the scan word.  1: 12 x v_bitop3.  2: the word without the addc (bitop3 instead).  3: without alignbit.
template <int BODY>
__global__ void __launch_bounds__(64) k(int iters, unsigned* out, unsigned long long* clk, unsigned a, unsigned b, int perLane)
{
    unsigned va = a, vb = b;
    if (perLane) { va = (threadIdx.x * 2654435761u + a) * 0x9e3779b1u; vb = (va ^ b) * 0x85ebca6bu + 0x7f4a7c15u; }
    asm volatile("s_mov_b64 s[20:21], 0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n"
                 "v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n v_mov_b32 v12, %0\n v_mov_b32 v13, %1\n v_mov_b32 v14, %0\n v_mov_b32 v15, %1\n"
                 "v_mov_b32 v16, %0\n v_mov_b32 v17, %1\n v_mov_b32 v18, %0\n v_mov_b32 v19, %1\n v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n" :: "v"(va), "v"(vb) : REGS);
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (BODY == 0) asm volatile(".rept 32\n" WORD("v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v2", "v3") ".endr\n" ::: REGS);
        if constexpr (BODY == 1) asm volatile(".rept 384\n v_bitop3_b32 v10, v10, v2, v3 bitop3:0x96\n .endr\n" ::: REGS);
        if constexpr (BODY == 2) asm volatile(".rept 384\n v_alignbit_b32 v10, v10, v2, 31\n .endr\n" ::: REGS);
        if constexpr (BODY == 3) asm volatile(".rept 384\n v_addc_co_u32_e64 v10, s[20:21], v10, v2, s[20:21]\n .endr\n" ::: REGS);
        if constexpr (BODY == 5) asm volatile(".p2align 3\n .rept 32\n" WORD("v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v2", "v3") ".endr\n" ::: REGS);
        if constexpr (BODY == 6) asm volatile(".p2align 3\n s_nop 0\n .rept 32\n" WORD("v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v2", "v3") ".endr\n" ::: REGS);
        if constexpr (BODY == 7) asm volatile(".p2align 3\n s_nop 0\n .rept 384\n v_bitop3_b32 v10, v10, v2, v3 bitop3:0x96\n .endr\n" ::: REGS);
        if constexpr (BODY == 8) asm volatile(".p2align 3\n s_nop 0\n .rept 384\n v_alignbit_b32 v10, v10, v2, 31\n .endr\n" ::: REGS);
        if constexpr (BODY == 9) asm volatile(".p2align 3\n .rept 384\n v_alignbit_b32 v10, v10, v2, 31\n .endr\n" ::: REGS);
        if constexpr (BODY == 10) asm volatile(".p2align 3\n .rept 32\n" WORD("v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v2", "v3") "s_nop 0\n" ".endr\n" ::: REGS);
        if constexpr (BODY == 11) asm volatile(".p2align 3\n .rept 32\n" WORD("v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v2", "v3") "s_nop 0\n s_nop 0\n" ".endr\n" ::: REGS);
        if constexpr (BODY == 4) asm volatile(".rept 384\n v_xor_b32_e64 v10, v10, v2\n .endr\n" ::: REGS);
    }
    const unsigned long long c1 = clock64();
    unsigned r; asm volatile("v_xor_b32 %0, v10, v12\n v_xor_b32 %0, %0, v19" : "=v"(r) :: REGS); if (r == 0x12345678u) out[threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = c1 - c0;
}
template <int BODY> static void run(const char* what, const char* data, unsigned a, unsigned b, int perLane, int wps, unsigned* d)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000 / wps;
    unsigned long long* clk = (unsigned long long*)(d + 64);
    hipLaunchKernelGGL((k<BODY>), dim3(1024 * wps), dim3(64), 0, 0, 1, d, clk, a, b, perLane);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<BODY>), dim3(1024 * wps), dim3(64), 0, 0, iters, d, clk, a, b, perLane);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * 384;
    printf("{\"body\": \"%s\", \"data\": \"%s\", \"waves_per_simd\": %d, \"simd_ns_per_instr\": %.4f, \"wave_clocks_per_instr\": %.2f}\n", what, data, wps, ms * 1e6 / n / wps, (double)h / n);
}
int main()
{
    unsigned* d; CK(hipMalloc(&d, 1024));
    for (int wps = 2; wps <= 8; wps *= 4) {
        run<5>("scan word, 8-byte aligned", "1 / 2", 1, 2, 0, wps, d);
        run<6>("scan word, 4 mod 8", "1 / 2", 1, 2, 0, wps, d);
        run<7>("v_bitop3 chain, 4 mod 8", "1 / 2", 1, 2, 0, wps, d);
        run<9>("v_alignbit chain, aligned", "1 / 2", 1, 2, 0, wps, d);
        run<8>("v_alignbit chain, 4 mod 8", "1 / 2", 1, 2, 0, wps, d);
        run<10>("scan word + one s_nop per word (phase alternates)", "1 / 2", 1, 2, 0, wps, d);
        run<11>("scan word + two s_nop per word", "1 / 2", 1, 2, 0, wps, d);
    }
    return 0;
}
