// ifetch_ubench.hip -- what bounds a wave that has its SIMD (almost) to itself: issue cadence, dependent latency, or the
// instruction stream?  (Round 6: the lane-per-pair NW scan holds 4 registers per band word and runs at 2 waves per SIMD;
// its first builds sat at 63-90 SIMD cycles per 12-instruction word.)  Straight-line bodies of N VALU instructions on
// 8 independent register chains, 4-byte (VOP2) and 8-byte (VOP3) encodings, body sizes from 0.5 KB to 32 KB, at 1, 2, 4 and
// 8 waves per SIMD.  Prints ns per instruction per WAVE and SIMD cycles per instruction at 2.1 GHz.
//   hipcc --offload-arch=gfx950 -O2 tools/ifetch_ubench.hip -o build/ifetch_ubench && build/ifetch_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define CH8(OP, SUF) OP " v10, v10, v2" SUF "\n" OP " v11, v11, v2" SUF "\n" OP " v12, v12, v2" SUF "\n" OP " v13, v13, v2" SUF "\n" \
                     OP " v14, v14, v2" SUF "\n" OP " v15, v15, v2" SUF "\n" OP " v16, v16, v2" SUF "\n" OP " v17, v17, v2" SUF "\n"
#define CH3(OP, SUF) OP " v10, v10, v2" SUF "\n" OP " v11, v11, v2" SUF "\n" OP " v12, v12, v2" SUF "\n"
#define CLOB "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17"

// MODE 0: v_xor_b32 (4 bytes), 8 chains.  1: v_bitop3_b32 (8 bytes), 8 chains.  2: v_xor, 3 chains.  3: v_bitop3, 3 chains.
// 4: the scan's mix per word: 9 full-rate (6 bitop3 + 3 VOP2) + 3 half-rate (2 alignbit + v_addc), 3-deep interleave
template <int MODE, int REPT>
__global__ void __launch_bounds__(64) body_kernel(int iters, unsigned* out)
{
    asm volatile("v_mov_b32 v2, 1\n v_mov_b32 v3, 2\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"
                 "v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n" ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) asm volatile(".rept %0\n" CH8("v_xor_b32", "") ".endr\n" :: "n"(REPT) : CLOB);
        if constexpr (MODE == 1) asm volatile(".rept %0\n" CH8("v_bitop3_b32", ", v3 bitop3:0x96") ".endr\n" :: "n"(REPT) : CLOB);
        if constexpr (MODE == 2) asm volatile(".rept %0\n" CH3("v_xor_b32", "") CH3("v_xor_b32", "") "v_xor_b32 v10, v10, v2\n v_xor_b32 v11, v11, v2\n" ".endr\n" :: "n"(REPT) : CLOB);
        if constexpr (MODE == 3) asm volatile(".rept %0\n" CH3("v_bitop3_b32", ", v3 bitop3:0x96") CH3("v_bitop3_b32", ", v3 bitop3:0x96") "v_bitop3_b32 v10, v10, v2, v3 bitop3:0x96\n v_bitop3_b32 v11, v11, v2, v3 bitop3:0x96\n" ".endr\n" :: "n"(REPT) : CLOB);
        if constexpr (MODE == 4) asm volatile(".rept %0\n"
            "v_xor_b32 v10, v10, v2\n v_bitop3_b32 v11, v11, v2, v3 bitop3:0x96\n v_alignbit_b32 v12, v12, v2, 31\n"
            "v_bitop3_b32 v10, v10, v2, v3 bitop3:0x96\n v_and_b32 v11, v11, v2\n v_alignbit_b32 v12, v12, v2, 31\n"
            "v_bitop3_b32 v10, v10, v2, v3 bitop3:0x96\n v_bitop3_b32 v11, v11, v2, v3 bitop3:0x96\n v_bitop3_b32 v12, v12, v2, v3 bitop3:0x96\n"
            "v_addc_co_u32 v10, vcc, v10, v2, vcc\n v_bitop3_b32 v12, v12, v2, v3 bitop3:0x96\n v_and_b32 v13, v12, v2\n"
            ".endr\n" :: "n"(REPT) : CLOB, "vcc");
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v10, v11\n v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v13" : "=v"(r) :: CLOB);
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int MODE, int REPT>
static void run(const char* what, int perRept, int bytesPerRept, unsigned* d)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 8; wps *= 2) {
        const long long target = 40000000LL;                      // instructions per wave
        const int iters = (int)(target / ((long long)perRept * REPT));
        hipLaunchKernelGGL((body_kernel<MODE, REPT>), dim3(1024 * wps), dim3(64), 0, 0, 1, d);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((body_kernel<MODE, REPT>), dim3(1024 * wps), dim3(64), 0, 0, iters, d);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)iters * perRept * REPT;
        printf("{\"body\": \"%s\", \"body_bytes\": %d, \"waves_per_simd\": %d, \"ns_per_instr_per_wave\": %.3f, \"simd_cycles_per_instr\": %.2f}\n",
               what, bytesPerRept * REPT, wps, ms * 1e6 / n, ms * 1e6 / n * 2.1 / wps);
    }
}

int main()
{
    unsigned* d; CK(hipMalloc(&d, 256));
    run<0, 8>("v_xor x8 chains", 8, 32, d);
    run<0, 128>("v_xor x8 chains", 8, 32, d);
    run<0, 1024>("v_xor x8 chains", 8, 32, d);
    run<1, 8>("v_bitop3 x8 chains", 8, 64, d);
    run<1, 128>("v_bitop3 x8 chains", 8, 64, d);
    run<1, 512>("v_bitop3 x8 chains", 8, 64, d);
    run<2, 64>("v_xor x3 chains", 8, 32, d);
    run<3, 64>("v_bitop3 x3 chains", 8, 64, d);
    run<4, 48>("scan mix, 3-deep", 12, 84, d);
    run<4, 400>("scan mix, 3-deep", 12, 84, d);
    return 0;
}
