// block64.hpp -- the 64-row block update shared by the block-per-lane kernels (pair_kernels.hip) and the fused
// single-pair kernel (one_pair.hip).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace edlib_amd {

// v_bitop3_b32 truth tables: bit i of the immediate is f(a,b,c) with i = a*4 + b*2 + c
#define BITOP3_XOR_OR(a, b, c)   __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xde)   /* (a ^ c) | b   */
#define BITOP3_OR_NOR(a, b, c)   __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xf1)   /* a | ~(b | c)  */

// reference calculateBlock (edlib.cpp:412-447) on one 64-row block held as two 32-bit halves.
// hpos / hneg are the two bits of hin (+1 / -1).  The booleans are written per half so that each
// 3-input function is one v_bitop3_b32; the add and the two shifts use the 64-bit pair
// (v_lshl_add_u64 / v_lshlrev_b64: 4 cycles per 64 bits, tools/valu_ubench.hip).  ph/mh return the
// un-shifted horizontal delta vectors (bit r = row r of the block).
struct Block64 { uint32_t p0, p1, m0, m1; };
__device__ __forceinline__ void advance_block64(Block64& B, const uint32_t e0, const uint32_t e1,
                                                const uint32_t hpos, const uint32_t hneg,
                                                uint32_t& ph0, uint32_t& ph1, uint32_t& mh0, uint32_t& mh1,
                                                uint32_t& xh0, uint32_t& xh1)
{
    const uint32_t xv0 = e0 | B.m0, xv1 = e1 | B.m1;                 // Xv = Eq | Mv        (:421)
    const uint32_t q0 = e0 | hneg;                                   // Eq |= hinIsNeg      (:423)
    const uint32_t t0 = q0 & B.p0, t1 = e1 & B.p1;
    unsigned long long s;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(s) : "v"(((unsigned long long)t1 << 32) | t0), "v"(((unsigned long long)B.p1 << 32) | B.p0));
    xh0 = BITOP3_XOR_OR((uint32_t)s, q0, B.p0); xh1 = BITOP3_XOR_OR((uint32_t)(s >> 32), e1, B.p1);                  // (:424)
    ph0 = BITOP3_OR_NOR(B.m0, xh0, B.p0); ph1 = BITOP3_OR_NOR(B.m1, xh1, B.p1);                        // (:426)
    mh0 = B.p0 & xh0; mh1 = B.p1 & xh1;                                                                // (:427)
    unsigned long long phs, mhs;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(phs) : "v"(((unsigned long long)ph1 << 32) | ph0));
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(mhs) : "v"(((unsigned long long)mh1 << 32) | mh0));
    const uint32_t a0 = (uint32_t)phs | hpos, b0 = (uint32_t)mhs | hneg;       // (:435-441)
    B.p0 = BITOP3_OR_NOR(b0, xv0, a0);
    B.p1 = BITOP3_OR_NOR((uint32_t)(mhs >> 32), xv1, (uint32_t)(phs >> 32));
    B.m0 = a0 & xv0;
    B.m1 = (uint32_t)(phs >> 32) & xv1;
}
__device__ __forceinline__ void advance_block64(Block64& B, const uint32_t e0, const uint32_t e1,
                                                const uint32_t hpos, const uint32_t hneg,
                                                uint32_t& ph0, uint32_t& ph1, uint32_t& mh0, uint32_t& mh1)
{
    uint32_t xh0, xh1;
    advance_block64(B, e0, e1, hpos, hneg, ph0, ph1, mh0, mh1, xh0, xh1);
}

// the two planes of a column-store entry (pair_kernels.hpp StoreEntry) from the block after the column, Ph and Xh
#define BITOP3_ANDN_OR(a, b, c)  __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x0e)   /* ~a & (b | c) */
__device__ __forceinline__ void store_planes(const Block64& B, const uint32_t ph0, const uint32_t ph1,
                                             const uint32_t xh0, const uint32_t xh1,
                                             unsigned long long& x, unsigned long long& y)
{
    const uint32_t y0 = BITOP3_ANDN_OR(B.p0, ph0, xh0), y1 = BITOP3_ANDN_OR(B.p1, ph1, xh1);   // (the builtin returns a signed int)
    x = ((unsigned long long)(B.p1 | ph1) << 32) | (B.p0 | ph0);
    y = ((unsigned long long)y1 << 32) | y0;
}

}  // namespace edlib_amd
