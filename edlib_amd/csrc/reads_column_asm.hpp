// reads_column_asm.hpp -- the column of scan_reads_kernel (reads_kernels.hip: the full-height scan of kernel A, pass 2 of the
// north-star batch) as ONE asm statement per word count.  calculateBlock (edlib.cpp:412-447) on NWD words of 32 rows as one
// long Myers word, top-down, 10 VALU ops per word; the bottom-row score moves with bit `sh` of the last word's horizontal delta.
//
// Why asm: a stream that mixes full-rate and half-rate VALU instructions issues at the sum of their rates only while its
// 8-byte instructions start at addresses that are 4 mod 8 (tools/data_ubench.hip); the compiler's column mixes 4-byte (VOP2)
// and 8-byte (VOP3) encodings, so the phase flips every few instructions: 158 SIMD cycles per 5-word column where its 35 full-
// rate + 18 half-rate instructions add up to 133 (rounds 1-5).  Here every instruction is a VOP3 encoding behind an alignment
// fence.  (The macro chains are unrolled by hand-run Python, see tools/gen_lanepair_asm.py for the pattern.)
#pragma once

#define RC_HEAD ".p2align 3\n\ts_nop 0\n\t"
#define RC_W_A(i)        "v_and_b32_e64 %[t], %[e" #i "], %[p" #i "]\n\t"
#define RC_W_ADD0(i)     "v_add_co_u32_e64 %[s], %[cy], %[t], %[p" #i "]\n\t"
#define RC_W_ADDC(i)     "v_addc_co_u32_e64 %[s], %[cy], %[t], %[p" #i "], %[cy]\n\t"
#define RC_W_B(i, c)     "v_bitop3_b32 %[xh], %[s], %[e" #i "], %[p" #i "] bitop3:0xde\n\t" \
                         "v_bitop3_b32 %[ph" #c "], %[m" #i "], %[xh], %[p" #i "] bitop3:0xf1\n\t" \
                         "v_and_b32_e64 %[mh" #c "], %[p" #i "], %[xh]\n\t"
#define RC_W_SH(c, p)    "v_alignbit_b32 %[phs], %[ph" #c "], %[ph" #p "], 31\n\t" "v_alignbit_b32 %[mhs], %[mh" #c "], %[mh" #p "], 31\n\t"
#define RC_W_SH_HW(c)    "v_alignbit_b32 %[phs], %[ph" #c "], 0, 31\n\t" "v_alignbit_b32 %[mhs], %[mh" #c "], 0, 31\n\t"      /* row -1 of HW: hin = 0 */
#define RC_W_SH_NW(c)    "v_alignbit_b32 %[phs], %[ph" #c "], -1, 31\n\t" "v_alignbit_b32 %[mhs], %[mh" #c "], 0, 31\n\t"     /* SHW / NW: hin = +1 (edlib.cpp:584,779) */
#define RC_W_C(i)        "v_or_b32_e64 %[xv], %[e" #i "], %[m" #i "]\n\t" \
                         "v_bitop3_b32 %[pn" #i "], %[mhs], %[xv], %[phs] bitop3:0xf1\n\t" \
                         "v_and_b32_e64 %[mn" #i "], %[phs], %[xv]\n\t"
#define RC_SCORE(c)      "v_bfe_u32 %[t], %[ph" #c "], %[sh], 1\n\t" "v_bfe_i32 %[s], %[mh" #c "], %[sh], 1\n\t" "v_add3_u32 %[scoreN], %[score], %[t], %[s]\n\t"
#define RC_WORD0_HW      RC_W_A(0) RC_W_ADD0(0) RC_W_B(0, 0) RC_W_SH_HW(0) RC_W_C(0)
#define RC_WORD0_NW      RC_W_A(0) RC_W_ADD0(0) RC_W_B(0, 0) RC_W_SH_NW(0) RC_W_C(0)
#define RC_WORD(i, c, p) RC_W_A(i) RC_W_ADDC(i) RC_W_B(i, c) RC_W_SH(c, p) RC_W_C(i)
// (new state in registers of its own, not tied to the old: with tied operands the four bodies of the symbol dispatch each got a
// copy of the whole state in front -- 11 v_mov per 5-word column)
#define RC_OUT_W(i) [pn##i] "=&v"(Pn[i]), [mn##i] "=&v"(Mn[i])
#define RC_IN_W(i) [e##i] "v"(Eq[i]), [p##i] "v"(Pv[i]), [m##i] "v"(Mv[i])
#define RC_REST_1
#define RC_REST_2 RC_REST_1 RC_WORD(1, 1, 0)
#define RC_REST_3 RC_REST_2 RC_WORD(2, 0, 1)
#define RC_REST_4 RC_REST_3 RC_WORD(3, 1, 0)
#define RC_REST_5 RC_REST_4 RC_WORD(4, 0, 1)
#define RC_REST_6 RC_REST_5 RC_WORD(5, 1, 0)
#define RC_REST_7 RC_REST_6 RC_WORD(6, 0, 1)
#define RC_REST_8 RC_REST_7 RC_WORD(7, 1, 0)
#define RC_OUTS_1 RC_OUT_W(0)
#define RC_INS_1 RC_IN_W(0)
#define RC_OUTS_2 RC_OUTS_1, RC_OUT_W(1)
#define RC_INS_2 RC_INS_1, RC_IN_W(1)
#define RC_OUTS_3 RC_OUTS_2, RC_OUT_W(2)
#define RC_INS_3 RC_INS_2, RC_IN_W(2)
#define RC_OUTS_4 RC_OUTS_3, RC_OUT_W(3)
#define RC_INS_4 RC_INS_3, RC_IN_W(3)
#define RC_OUTS_5 RC_OUTS_4, RC_OUT_W(4)
#define RC_INS_5 RC_INS_4, RC_IN_W(4)
#define RC_OUTS_6 RC_OUTS_5, RC_OUT_W(5)
#define RC_INS_6 RC_INS_5, RC_IN_W(5)
#define RC_OUTS_7 RC_OUTS_6, RC_OUT_W(6)
#define RC_INS_7 RC_INS_6, RC_IN_W(6)
#define RC_OUTS_8 RC_OUTS_7, RC_OUT_W(7)
#define RC_INS_8 RC_INS_7, RC_IN_W(7)
#define RC_LAST_1 0
#define RC_LAST_2 1
#define RC_LAST_3 0
#define RC_LAST_4 1
#define RC_LAST_5 0
#define RC_LAST_6 1
#define RC_LAST_7 0
#define RC_LAST_8 1
#define RC_TEMPS [t] "=&v"(t_), [s] "=&v"(s_), [xh] "=&v"(xh_), [ph0] "=&v"(ph0_), [ph1] "=&v"(ph1_), [mh0] "=&v"(mh0_), [mh1] "=&v"(mh1_), [phs] "=&v"(phs_), [mhs] "=&v"(mhs_), [xv] "=&v"(xv_), [cy] "=&s"(cy_)
#define RC_SCORE_X(c) RC_SCORE(c)
#define RC_COLUMN_ASM(N, WORD0) asm(RC_HEAD WORD0 RC_REST_##N RC_SCORE_X(RC_LAST_##N) : RC_OUTS_##N, [scoreN] "=&v"(scoreN), RC_TEMPS : RC_INS_##N, [score] "v"(score), [sh] "v"(sh))
// the same column without a followed row (the band of scan_reads_banded_kernel below its full height)
#define RC_COLUMN_ASM_NS(N, WORD0) asm(RC_HEAD WORD0 RC_REST_##N : RC_OUTS_##N, RC_TEMPS : RC_INS_##N)
#define RC_COLUMN_DISPATCH_NS(NA, WORD0) \
    if constexpr (NA == 1) RC_COLUMN_ASM_NS(1, WORD0); if constexpr (NA == 2) RC_COLUMN_ASM_NS(2, WORD0); if constexpr (NA == 3) RC_COLUMN_ASM_NS(3, WORD0); \
    if constexpr (NA == 4) RC_COLUMN_ASM_NS(4, WORD0); if constexpr (NA == 5) RC_COLUMN_ASM_NS(5, WORD0); if constexpr (NA == 6) RC_COLUMN_ASM_NS(6, WORD0); \
    if constexpr (NA == 7) RC_COLUMN_ASM_NS(7, WORD0); if constexpr (NA == 8) RC_COLUMN_ASM_NS(8, WORD0);
#define RC_COLUMN_DISPATCH(NWD, WORD0) \
    if constexpr (NWD == 1) RC_COLUMN_ASM(1, WORD0); if constexpr (NWD == 2) RC_COLUMN_ASM(2, WORD0); if constexpr (NWD == 3) RC_COLUMN_ASM(3, WORD0); \
    if constexpr (NWD == 4) RC_COLUMN_ASM(4, WORD0); if constexpr (NWD == 5) RC_COLUMN_ASM(5, WORD0); if constexpr (NWD == 6) RC_COLUMN_ASM(6, WORD0); \
    if constexpr (NWD == 7) RC_COLUMN_ASM(7, WORD0); if constexpr (NWD == 8) RC_COLUMN_ASM(8, WORD0);
