// pk16_ubench.hip -- would kernel A's first band word fit two reads per lane?  Issue rates of the packed 16-bit integer ops
// the Myers recurrence would need on two 16-row halves of a dword (v_pk_add_u16, v_pk_lshlrev_b16, v_pk_lshrrev_b16)
// against v_and_b32 / v_add_u32, and the one-word column body in both forms: 32 rows of one read per lane (today's pass 1)
// and 16 rows of two reads per lane.  Eight independent chains per lane, 8 waves per SIMD, all 256 CUs (tools/valu_ubench.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/pk16_ubench.hip -o build/pk16_ubench && build/pk16_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) k_op(u32* out, int iters, u32 seed)
{
    u32 r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1) + i * 77;
    u32 a = seed ^ threadIdx.x, one = 0x00010001u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OP(i)                                                                                           \
            if (KIND == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));                   \
            if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));                   \
            if (KIND == 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a));                \
            if (KIND == 3) asm volatile("v_pk_lshlrev_b16 %0, %1, %0" : "+v"(r[i]) : "v"(one));          \
            if (KIND == 4) asm volatile("v_pk_lshrrev_b16 %0, %1, %0" : "+v"(r[i]) : "v"(one));          \
            if (KIND == 5) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP8(OP)
#undef OP
        }
    }
    u32 s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one band word per column, HW top row (carry in 0), Eq chosen by the target symbol from four registers
template <bool PACKED>
__global__ void __launch_bounds__(256) k_body(u32* out, const u32* __restrict__ tpk, int nwords, u32 seed)
{
    u32 E[4], Pv = ~0u, Mv = 0u;
    for (int i = 0; i < 4; ++i) E[i] = (seed * (threadIdx.x + 3) + i) * 2654435761u;
    const u32 one = 0x00010001u;
    u32 acc = 0;
    for (int w = 0; w < nwords; ++w) {
        const u32 tw = tpk[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 sym = (tw >> (2 * j)) & 3u;                     // (wave-uniform: scalar selects)
            const u32 Eq = sym == 0 ? E[0] : (sym == 1 ? E[1] : (sym == 2 ? E[2] : E[3]));
            const u32 t = Eq & Pv;
            u32 s;
            if (PACKED) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(s) : "v"(t), "v"(Pv));
            else s = t + Pv;
            const u32 Xh = (s ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
            acc += Ph >> 31;                                           // (stands in for the checkpoint's use of the deltas)
            if (PACKED) {
                asm volatile("v_pk_lshlrev_b16 %0, %1, %0" : "+v"(Ph) : "v"(one));
                asm volatile("v_pk_lshlrev_b16 %0, %1, %0" : "+v"(Mh) : "v"(one));
            } else { Ph += Ph; Mh += Mh; }
            const u32 Xv = Eq | Mv;
            Pv = Mh | ~(Xv | Ph); Mv = Ph & Xv;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Pv ^ Mv ^ acc;
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
    const int wps = 8, blocks = 256 * wps, iters = 20000, nwords = 1 << 15;
    u32* out; CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(u32)));
    u32* tpk; CK(hipMalloc(&tpk, nwords * 4)); CK(hipMemset(tpk, 0x1b, nwords * 4));
    const char* names[6] = {"v_and_b32", "v_add_u32", "v_pk_add_u16", "v_pk_lshlrev_b16", "v_pk_lshrrev_b16", "v_pk_sub_u16"};
    const double laneops = (double)blocks * 256 * iters * 64;
    printf("{\"waves_per_simd\": %d, \"ops\": {", wps);
#define RUN(K) { float ms = time_ms([&] { hipLaunchKernelGGL(k_op<K>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u); }); \
                 printf("%s\"%s\": {\"ms\": %.3f, \"T_lane_ops_per_s\": %.2f}", K ? ", " : "", names[K], ms, laneops / ms / 1e9); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    const double cols = (double)nwords * 16;
    float m32 = time_ms([&] { hipLaunchKernelGGL(k_body<false>, dim3(blocks), dim3(256), 0, 0, out, tpk, nwords, 7u); });
    float m16 = time_ms([&] { hipLaunchKernelGGL(k_body<true>, dim3(blocks), dim3(256), 0, 0, out, tpk, nwords, 7u); });
    printf("}, \"one_word_column_body\": {\"columns\": %.0f, \"u32_one_read_per_lane_ms\": %.3f, \"pk16_two_reads_per_lane_ms\": %.3f, "
           "\"ns_per_wave_column_u32\": %.2f, \"ns_per_wave_column_pk16\": %.2f, \"reads_per_lane_pk16\": 2}}\n",
           cols, m32, m16, m32 * 1e6 / (cols * wps), m16 * 1e6 / (cols * wps));
    return 0;
}
