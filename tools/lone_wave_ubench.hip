// lone_wave_ubench.hip -- what ONE wave64 alone on its SIMD gets per instruction: the wide kernel (wide_kernels.hip) is a
// chain of dependent steps on lone waves, so its step time is set by single-wave issue and result latencies, not by the
// 8-waves-per-SIMD rates of valu_ubench.hip.  One workgroup of 64 threads; cycles from s_memtime around N instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/lone_wave_ubench.hip -o build/lone_wave_ubench && build/lone_wave_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
typedef unsigned long long u64;

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

template <int KIND>
__global__ void __launch_bounds__(64) k(u64* out, u32 seed, int iters)
{
    __shared__ u32 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 4) & 4092;
    __syncthreads();
    u32 a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x1234, d = ~seed;
    u32 e = a + 1, f = b + 2, g = c + 3, h = d + 4;
    u32 addr = (threadIdx.x * 4) & 4092;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }                       // dependent chain
        if (KIND == 1) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                                            : "+v"(a), "+v"(c), "+v"(e), "+v"(g) : "v"(b));) }                    // 4 independent chains
        if (KIND == 2) { REP64(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xde" : "+v"(a) : "v"(b), "v"(c));) }
        if (KIND == 3) { REP64(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a) : "v"(b));) }
        if (KIND == 4) { REP64(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }   // dependent DPP chain
        if (KIND == 5) { REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %1, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %1, %1, %2"
                                            : "+v"(a), "+v"(c) : "v"(b));) }                                      // dpp + 3 dependent adds (64 instr)
        if (KIND == 6) { REP64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr));) }          // dependent LDS chain
        if (KIND == 7) { REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(seed) :: "scc");) }   // dependent SALU (s_add writes SCC: without the clobber the loop's own compare was lost and the kernel never ended)
        if (KIND == 8) { REP16(asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 1" : "+v"(a), "+s"(seed) : "v"(b) : "scc");) }  // VALU / SALU interleaved
        if (KIND == 9) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %0, %4\n v_add_u32 %3, %1, %4"
                                            : "+v"(a), "+v"(c), "+v"(e), "+v"(g) : "v"(b));) }                    // 2 chains
        if (KIND == 10) { REP16(asm volatile("v_readlane_b32 %1, %0, 3\n v_mov_b32 %0, %1\n v_readlane_b32 %1, %0, 3\n v_mov_b32 %0, %1" : "+v"(a), "+s"(seed));) }   // readlane -> mov round trips
        if (KIND == 11) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                                             : "+v"(a), "+v"(c), "+v"(e), "+v"(g), "+v"(d), "+v"(f), "+v"(h), "+v"(addr) : "v"(b));) }   // 8 independent chains (128 instr)
        if (KIND == 12) { REP16(asm volatile("ds_read_b32 %1, %0\n v_add_u32 %2, %2, %3\n v_add_u32 %2, %2, %3\n s_waitcnt lgkmcnt(0)" : "+v"(addr), "+v"(a), "+v"(c) : "v"(b));) }  // LDS + 2 adds, waited
        if (KIND == 13) { REP64(asm volatile("v_lshrrev_b32 %0, 31, %0" : "+v"(a));) }
        if (KIND == 14) { REP64(asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 15) { REP64(asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xca" : "+v"(a) : "s"(seed), "v"(c));) }  // SGPR operand
    }
    const u64 t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + c + e + g + d + f + h + addr + seed; }
}

template <int KIND> static void run(const char* what, int instrPerIter)
{
    u64* d; hipMalloc(&d, 16);
    const int iters = 2000;
    k<KIND><<<1, 64>>>(d, 5, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KIND><<<1, 64>>>(d, 5, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * instrPerIter;
    printf("{\"case\": \"%s\", \"ns_per_instr\": %.2f, \"counter_ticks_per_instr\": %.3f}\n", what, ms * 1e6 / n, (double)h[0] / n);
    hipFree(d);
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    run<0>("v_add_u32 dependent chain", 64);
    run<1>("v_add_u32 4 independent chains", 64);
    run<9>("v_add_u32 2 chains", 64);
    run<11>("v_add_u32 8 independent chains", 128);
    run<2>("v_bitop3 dependent chain", 64);
    run<3>("v_alignbit dependent chain", 64);
    run<13>("v_lshrrev dependent chain", 64);
    run<14>("v_and dependent chain", 64);
    run<15>("v_bitop3 with SGPR operand, dependent", 64);
    run<4>("v_mov_dpp wave_shr dependent chain", 64);
    run<5>("dpp + 3 dependent adds", 64);
    run<6>("ds_read_b32 dependent chain (per read)", 64);
    run<12>("ds_read + 2 adds + wait (per group of 3)", 16);
    run<7>("s_add dependent chain", 64);
    run<8>("v_add / s_add interleaved", 64);
    run<10>("readlane -> v_mov round trip (per instr)", 64);
    return 0;
}
