// bf16x3.h -- exact split of fp32 values into three bf16 terms for v_mfma_f32_16x16x32_bf16.
//
// x = x0 + x1 + x2 with x0 = the upper 16 bits of x (truncation), x1 = the upper 16 bits of (x - x0), x2 = x - x0 - x1:
// every subtraction is exact and x2 has at most 8 significant bits, so the three terms carry all 24 significand bits.
// A product is then a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 (the dropped terms are below 2^-24 relative), accumulated in
// fp32 by the MFMA.  Scalar subtractions on purpose: packed fp32 VALU next to MFMAs measured slower, and
// v_dot2c_f32_bf16 is neither exact nor faster here (scripts/ubench/split_rate.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace fn2 {

typedef short __attribute__((ext_vector_type(8))) bf16x8;   // MFMA operand: 8 bf16 in 4 VGPRs
typedef unsigned __attribute__((ext_vector_type(4))) u4;
typedef unsigned __attribute__((ext_vector_type(2))) u2;

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi)
{
    // bf16(lo) in bits 0..15, bf16(hi) in bits 16..31 (both by truncation: the upper halves of the floats)
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// 8 values -> three packed operands
__device__ __forceinline__ void split3(const float (&r)[8], u4 &t0, u4 &t1, u4 &t2)
{
    float h0[8], h1[8], h2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h0[k] = trunc_bf16(r[k]);
        const float r1 = r[k] - h0[k];   // exact
        h1[k] = trunc_bf16(r1);
        h2[k] = r1 - h1[k];              // exact, at most 8 significant bits
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        t0[m] = pack_hi16(h0[2 * m], h0[2 * m + 1]);
        t1[m] = pack_hi16(h1[2 * m], h1[2 * m + 1]);
        t2[m] = pack_hi16(h2[2 * m], h2[2 * m + 1]);
    }
}

// 4 values -> three packed half operands
__device__ __forceinline__ void split3_half(const float (&r)[4], u2 &t0, u2 &t1, u2 &t2)
{
    float h0[4], h1[4], h2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h0[k] = trunc_bf16(r[k]);
        const float r1 = r[k] - h0[k];
        h1[k] = trunc_bf16(r1);
        h2[k] = r1 - h1[k];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        t0[m] = pack_hi16(h0[2 * m], h0[2 * m + 1]);
        t1[m] = pack_hi16(h1[2 * m], h1[2 * m + 1]);
        t2[m] = pack_hi16(h2[2 * m], h2[2 * m + 1]);
    }
}

} // namespace fn2
