// mix_ubench.hip -- SIMD cycles per instruction of the lane-per-pair scan's instruction classes, alone and mixed with
// full-rate v_bitop3_b32 (round 6: the scan's 12-op word ran at 3.55 cycles per instruction at every occupancy where
// 9 full-rate + 3 half-rate ops predict 2.5).  8 waves per SIMD, independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CLOB "v2", "v3", "v4", "v5", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "vcc", "s20", "s21", "s22", "s23"
#define B3(r) "v_bitop3_b32 " r ", " r ", v2, v3 bitop3:0x96\n"
#define AL(r) "v_alignbit_b32 " r ", " r ", v2, 31\n"
#define AC(r) "v_addc_co_u32 " r ", vcc, " r ", v2, vcc\n"
#define ACS(r) "v_addc_co_u32_e64 " r ", s[20:21], " r ", v2, s[20:21]\n"
#define AD(r) "v_add_u32 " r ", " r ", v2\n"
#define ADC(r) "v_add_co_u32 " r ", vcc, " r ", v2\n"
#define XR(r) "v_xor_b32 " r ", " r ", v2\n"
#define XR64(r) "v_xor_b32_e64 " r ", " r ", v2\n"
#define LO(r) "v_lshl_or_b32 " r ", " r ", 1, v2\n"
#define LA64(r, rr) "v_lshl_add_u64 " rr ", " rr ", 1, v[4:5]\n"
#define SH64(rr) "v_lshlrev_b64 " rr ", 1, " rr "\n"
#define BFE(r) "v_bfe_i32 " r ", " r ", 0, 1\n"
#define LSR(r) "v_lshrrev_b32 " r ", 1, " r "\n"
#define LSL(r) "v_lshlrev_b32 " r ", 1, " r "\n"
#define ANDOR(r) "v_and_or_b32 " r ", " r ", v2, v3\n"
#define OR3(r) "v_or3_b32 " r ", " r ", v2, v3\n"
#define CND(r) "v_cndmask_b32 " r ", " r ", v2, vcc\n"

template <int MODE>
__global__ void __launch_bounds__(64) k(int iters, unsigned* out)
{
    asm volatile("v_mov_b32 v2, 1\n v_mov_b32 v3, 2\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n s_mov_b64 s[20:21], 0\n s_mov_b64 vcc, 0\n"
                 "v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n"
                 "v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" ::: CLOB);
    for (int it = 0; it < iters; ++it) {
#define BODY(S) asm volatile(".rept 32\n" S ".endr\n" ::: CLOB)
        if constexpr (MODE == 0) BODY(B3("v10") B3("v11") B3("v12") B3("v13") B3("v14") B3("v15") B3("v16") B3("v17"));
        if constexpr (MODE == 1) BODY(AL("v10") AL("v11") AL("v12") AL("v13") AL("v14") AL("v15") AL("v16") AL("v17"));
        if constexpr (MODE == 2) BODY(AC("v10") AC("v11") AC("v12") AC("v13") AC("v14") AC("v15") AC("v16") AC("v17"));
        if constexpr (MODE == 3) BODY(AL("v10") B3("v11") B3("v12") B3("v13") AL("v14") B3("v15") B3("v16") B3("v17"));
        if constexpr (MODE == 4) BODY(AC("v10") B3("v11") B3("v12") B3("v13") AC("v14") B3("v15") B3("v16") B3("v17"));
        if constexpr (MODE == 5) BODY(AL("v10") AL("v11") B3("v12") B3("v13") B3("v14") B3("v15") B3("v16") B3("v17"));
        if constexpr (MODE == 6) BODY(ACS("v10") B3("v11") B3("v12") B3("v13") ACS("v14") B3("v15") B3("v16") B3("v17"));
        if constexpr (MODE == 7) BODY(AD("v10") AD("v11") AD("v12") AD("v13") AD("v14") AD("v15") AD("v16") AD("v17"));
        if constexpr (MODE == 8) BODY(ADC("v10") B3("v11") B3("v12") B3("v13") ADC("v14") B3("v15") B3("v16") B3("v17"));
        if constexpr (MODE == 9) BODY(XR("v10") XR("v11") B3("v12") B3("v13") XR("v14") XR("v15") B3("v16") B3("v17"));
        if constexpr (MODE == 10) BODY(LO("v10") LO("v11") LO("v12") LO("v13") LO("v14") LO("v15") LO("v16") LO("v17"));
        if constexpr (MODE == 11) BODY(LA64("v10", "v[10:11]") LA64("v12", "v[12:13]") LA64("v14", "v[14:15]") LA64("v16", "v[16:17]") LA64("v18", "v[18:19]") LA64("v20", "v[20:21]") LA64("v22", "v[22:23]") LA64("v10", "v[10:11]"));
        if constexpr (MODE == 12) BODY(SH64("v[10:11]") SH64("v[12:13]") SH64("v[14:15]") SH64("v[16:17]") SH64("v[18:19]") SH64("v[20:21]") SH64("v[22:23]") SH64("v[10:11]"));
        if constexpr (MODE == 13) BODY(BFE("v10") BFE("v11") BFE("v12") BFE("v13") BFE("v14") BFE("v15") BFE("v16") BFE("v17"));
        if constexpr (MODE == 14) BODY(LSR("v10") LSR("v11") LSR("v12") LSR("v13") LSR("v14") LSR("v15") LSR("v16") LSR("v17"));
        if constexpr (MODE == 15) BODY(LSL("v10") LSL("v11") LSL("v12") LSL("v13") LSL("v14") LSL("v15") LSL("v16") LSL("v17"));
        if constexpr (MODE == 16) BODY(ANDOR("v10") ANDOR("v11") ANDOR("v12") ANDOR("v13") ANDOR("v14") ANDOR("v15") ANDOR("v16") ANDOR("v17"));
        if constexpr (MODE == 17) BODY(CND("v10") CND("v11") CND("v12") CND("v13") CND("v14") CND("v15") CND("v16") CND("v17"));
        if constexpr (MODE == 18) BODY(XR("v10") B3("v11") AL("v12") B3("v13") XR("v14") AL("v15") B3("v16") B3("v17") B3("v18") AC("v19") B3("v20") XR("v21"));
        if constexpr (MODE == 19) BODY(XR("v10") B3("v11") B3("v12") B3("v13") XR("v14") B3("v15") B3("v16") B3("v17") B3("v18") AC("v19") B3("v20") XR("v21"));
        if constexpr (MODE == 21) BODY(B3("v10") B3("v11") AL("v12") B3("v13") B3("v14") AL("v15") B3("v16") B3("v17") B3("v18") B3("v19") B3("v20") B3("v21"));
        if constexpr (MODE == 22) BODY(XR("v10") XR("v11") XR("v13") AL("v12") XR("v14") XR("v16") XR("v17") AL("v15"));
        if constexpr (MODE == 23) BODY(XR("v10") B3("v11") B3("v13") AL("v12") XR("v14") B3("v16") B3("v17") AL("v15"));
        if constexpr (MODE == 24) BODY(XR("v10") B3("v11") AL("v12") B3("v13") B3("v14") AL("v15") B3("v16") B3("v17") B3("v18") B3("v19") B3("v20") B3("v21"));
        if constexpr (MODE == 25) BODY(B3("v10") B3("v11") AL("v12") B3("v13") B3("v14") AL("v15") B3("v16") B3("v17") B3("v18") AC("v19") B3("v20") B3("v21"));
        if constexpr (MODE == 26) BODY(XR("v10") B3("v11") B3("v13") B3("v14") XR("v16") B3("v17") B3("v18") B3("v19") B3("v20") XR("v21") AL("v12") AL("v15"));
        if constexpr (MODE == 27) BODY(AL("v12") AL("v15") AC("v19") XR("v10") B3("v11") B3("v13") B3("v14") XR("v16") B3("v17") B3("v18") B3("v20") XR("v21"));
        if constexpr (MODE == 28) BODY(XR64("v10") XR64("v11") XR64("v13") AL("v12") XR64("v14") XR64("v16") XR64("v17") AL("v15"));
        if constexpr (MODE == 29) BODY(XR("v10") XR("v11") XR("v13") LO("v12") XR("v14") XR("v16") XR("v17") LO("v15"));
        if constexpr (MODE == 30) BODY(XR("v10") XR("v11") XR("v13") SH64("v[18:19]") XR("v14") XR("v16") XR("v17") SH64("v[20:21]"));
        if constexpr (MODE == 31) BODY(XR("v10") XR("v11") XR("v13") AL("v12") "s_nop 0\n" XR("v14") XR("v16") XR("v17") AL("v15") "s_nop 0\n");
        if constexpr (MODE == 32) BODY(XR("v10") XR("v11") XR("v13") XR("v12") XR("v14") XR("v16") AL("v17") AL("v15"));
        if constexpr (MODE == 33) BODY(XR64("v10") B3("v11") AL("v12") B3("v13") XR64("v14") AL("v15") B3("v16") B3("v17") B3("v18") ACS("v19") B3("v20") XR64("v21"));
        if constexpr (MODE == 34) BODY(AD("v10") AD("v11") AD("v13") AL("v12") AD("v14") AD("v16") AD("v17") AL("v15"));
        if constexpr (MODE == 20) BODY(XR("v10") B3("v11") AL("v12") B3("v13") XR("v14") AL("v15") B3("v16") B3("v17") B3("v18") B3("v19") B3("v20") XR("v21"));
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v10, v11\n v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v13\n v_xor_b32 %0, %0, v19\n v_xor_b32 %0, %0, v21" : "=v"(r) :: CLOB);
    if (r == 0x12345678u) out[threadIdx.x] = r;
}
template <int MODE> static void run(const char* what, int per, unsigned* d)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(8192), dim3(64), 0, 0, 1, d);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE>), dim3(8192), dim3(64), 0, 0, iters, d);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * per * 32;
    printf("{\"mix\": \"%s\", \"simd_ns_per_instr\": %.4f, \"simd_cycles_per_instr_at_2.1GHz\": %.2f}\n", what, ms * 1e6 / n / 8, ms * 1e6 / n / 8 * 2.1);
}
int main()
{
    unsigned* d; CK(hipMalloc(&d, 256));
    run<0>("8 x v_bitop3", 8, d); run<1>("8 x v_alignbit", 8, d); run<2>("8 x v_addc_co (vcc)", 8, d);
    run<3>("alignbit + 3 bitop3", 8, d); run<4>("addc(vcc) + 3 bitop3", 8, d); run<5>("2 alignbit adjacent + 6 bitop3", 8, d);
    run<6>("addc(sgpr pair) + 3 bitop3", 8, d); run<7>("8 x v_add_u32", 8, d); run<8>("v_add_co + 3 bitop3", 8, d);
    run<9>("2 v_xor + 2 bitop3", 8, d); run<10>("8 x v_lshl_or_b32", 8, d); run<11>("8 x v_lshl_add_u64", 8, d); run<12>("8 x v_lshlrev_b64", 8, d);
    run<13>("8 x v_bfe_i32", 8, d); run<14>("8 x v_lshrrev_b32", 8, d); run<15>("8 x v_lshlrev_b32", 8, d); run<16>("8 x v_and_or_b32", 8, d); run<17>("8 x v_cndmask(vcc)", 8, d);
    run<18>("scan word: 3 xor/and, 6 bitop3, 2 alignbit, 1 addc", 12, d); run<19>("scan word without alignbit", 12, d); run<20>("scan word without addc", 12, d);
    run<21>("as 'without addc' but v_xor -> bitop3", 12, d); run<22>("3 v_xor + alignbit", 8, d); run<23>("xor, 2 bitop3, alignbit", 8, d);
    run<24>("one v_xor, 9 bitop3, 2 alignbit", 12, d); run<25>("9 bitop3, 2 alignbit, addc", 12, d); run<26>("scan word (no addc), alignbits last and adjacent", 12, d);
    run<28>("3 v_xor_e64 + alignbit", 8, d); run<29>("3 v_xor + v_lshl_or", 8, d); run<30>("3 v_xor + v_lshlrev_b64", 8, d); run<31>("3 v_xor + alignbit + s_nop", 8, d);
    run<32>("6 v_xor + 2 alignbit adjacent", 8, d); run<33>("scan word, all VOP3 encodings (xor e64, addc sgpr pair)", 12, d); run<34>("3 v_add_u32 + alignbit", 8, d);
    run<27>("scan word, alignbit alignbit addc first", 12, d);
    return 0;
}
