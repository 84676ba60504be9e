// f16x2_split.h -- the block-scaled two-term f16 split shared by the f16x2 correlation kernels (forward and backward).
//
// An fp32 operand x of a task (one workgroup pass: an A/B tile pair of the forward, an X/G operand pair of the backward) is
// represented as  x * 2^k = h + l,  h = RNE_f16(x * 2^k),  l = RNE_f16(x * 2^k - h)  with ONE exponent k per operand and
// task; the product of two operands then carries 2^(ka + kb), which the epilogue removes exactly (v_ldexp_f32) together with
// the 1/C of the reference (correlation_cuda_kernel.cu:139-143, :229, :322).
//
// Why: f16 keeps 11 bits down to 2^-14 only, so h + l carries >= 22 bits of a value above 2^-3, has an ABSOLUTE floor of
// 2^-25 below that and overflows at 65520.  Unscaled (round 2) that window sat at fixed magnitudes: operands of 1e-6 -- what
// gradOutput is in training -- kept 2 digits, 1e-8 became 0.  The reference multiplies and adds in fp32 at any magnitude
// (correlation_cuda_kernel.cu:112,124,214-229).
//
// k places the TYPICAL magnitude of the operand -- the mean binary exponent of the non-zero values of a sample of 256
// elements spread over the task's operand (all channels of the tile / the gO image of the central displacements), read
// straight from global memory by the staging waves while the previous task finishes -- at 2^T_GEO = 1/2, where unit-variance
// data sit by themselves (the mean exponent of N(0,1) values is -1.4: k = 0, the split of round 2):
//   - values down to 2^-(3 + T_GEO) = 1/4 of the typical magnitude keep >= 22 bits, smaller ones an absolute error of
//     2^-(25 + T_GEO) = 2^-24 of it: fp32-class sums at any input magnitude (measured: 0.65 x the fp32 MFMA kernel's error
//     against fp64 from 2^-27 to 2^13, tests/test_gpu_parity.py);
//   - values up to 2^(16 - T_GEO) = 2^17 = 131072 x the typical magnitude fit; anything larger makes h infinite, the outputs it
//     touches come out non-finite and are recomputed by a plain fp32 fma chain (exact_corr / exact_grad: slow, always right).
// The mean exponent (not the maximum) is used because a single huge element must not push everything else of the tile
// into the f16 subnormals.  Scaling by a power of two is exact: for operands of unit magnitude the split, the products
// and the sums are those of the unscaled kernel up to the position of the floor.
#pragma once

namespace fn2 {
namespace f16s {

constexpr int T_GEO = -1;   // the sample's mean exponent lands at 2^-1

// Two-term split of an (already scaled) pair: h = RNE_f16 of both values (v_cvt_pk_f16_f32), l = RNE_f16 of the exact fp32
// residuals (v_fma_mix_f32 forms x - h in one instruction): four VALU instructions per pair.  The scale itself is a
// v_pk_mul_f32 on two adjacent registers (callers multiply whole 16-byte vectors: half an instruction per value; measured
// with scripts/ubench/f16_split_rate.hip: 4 clocks per v_pk_mul_f32 -- folding the scale into v_fma_mix{lo,hi}_f16 instead
// needs no extra instruction but those run at half rate: 8 clocks each, +50 % on the whole split).
__device__ __forceinline__ unsigned pk_f16(float a, float b)   // v_cvt_pk_f16_f32, round to nearest even
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){a, b}, h2v));
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &l)
{
    h = pk_f16(x0, x1);
    float r0, r1;   // x - (float)half: one instruction each, exact
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    l = pk_f16(r0, r1);
}

// v * 2^k for two / four adjacent registers: v_pk_mul_f32 with the scale pair {2^k, 2^k} in SGPRs (written as assembly with
// a 64-bit integer operand: left to the compiler -- or given a float2 operand -- the uniform scale ends up in a VGPR pair,
// which the matrix waves of the backward kernel do not have)
typedef float f2s __attribute__((ext_vector_type(2)));
typedef float f4s __attribute__((ext_vector_type(4)));
typedef unsigned long long scale2_t;
// (A scale of 1 -- unit-magnitude operands, k = 0 -- is multiplied like any other: a uniform branch around the two or four
// v_pk_mul_f32 of an item measured as slow as the multiplies it skips, it splits the staging loop's basic blocks; a second
// instantiation of the step loop for k = 0 spilled 66 registers.)
__device__ __forceinline__ f2s pk_scale(f2s v, scale2_t s2)
{
#ifdef FN2_ABL_NOMUL   // timing ablation (scripts/gpu_ablate.sh): results wrong unless k == 0
    return v;
#else
    f2s r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(v), "s"(s2));
    return r;
#endif
}
__device__ __forceinline__ f4s pk_scale4(f4s v, scale2_t s2)
{
    const f2s lo = pk_scale(__builtin_shufflevector(v, v, 0, 1), s2), hi = pk_scale(__builtin_shufflevector(v, v, 2, 3), s2);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// A value the compiler must keep in an SGPR (v_readfirstlane as assembly: the builtin is dropped when the operand is known to
// be uniform, and everything derived from it -- e.g. the scale pair of pk_scale -- then lives in vector registers).  The s_nops
// are the wait states the hazard recognizer inserts around a v_readfirstlane it can see: without the first one the instruction
// read the register BEFORE the preceding VALU instruction had written it (scripts/ubench/scale_probe.hip).
__device__ __forceinline__ int to_sgpr(int v)
{
    int s;
    asm volatile("s_nop 4\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(s) : "v"(v));
    return s;
}

// Exponent statistics of a sample of NS values per lane (one wave): biased exponents of the non-zero values, count in the
// upper half word.  One packed DPP reduction; the result is wave-uniform (SGPR).
__device__ __forceinline__ unsigned exp_stat(unsigned bits)   // one value -> (e != 0) << 16 | e
{
    const unsigned e = (bits >> 23) & 0xffu;
    return e + ((e != 0u ? 1u : 0u) << 16);
}
__device__ __forceinline__ unsigned wave_sum(unsigned b)      // DPP butterflies inside the rows of 16, the four rows on the scalar unit
{
    b += (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    b += (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    b += (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xf, 0xf, false);   // row_half_mirror
    b += (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xf, 0xf, false);   // row_mirror
    return __builtin_amdgcn_readlane(b, 0) + __builtin_amdgcn_readlane(b, 16) + __builtin_amdgcn_readlane(b, 32) +
           __builtin_amdgcn_readlane(b, 48);
}

// exponent k of the scale 2^k from a wave's packed statistics (sum of the lanes' exp_stat values; <= 256 samples: no carry
// into the count).  No non-zero value in the sample: k = 0, the unscaled split.
__device__ __forceinline__ int scale_exp(unsigned packed)
{
    const unsigned sum = packed & 0xffffu, cnt = packed >> 16;
    if (cnt == 0u) return 0;
    const int e = (int)((float)sum * __builtin_amdgcn_rcpf((float)cnt) + 0.5f);   // mean biased exponent, 1 .. 255
    const int k = T_GEO + 127 - to_sgpr(e);                                        // -129 .. 125 before clamping
    // 2^k must be a NORMAL float for scale_from_exp / scale2_from_exp (biased exponent 127 + k in 1 .. 254): a sample around
    // 2^126 (e = 253) would otherwise encode the scale as 0.0 -- every finite operand of the tile multiplied to zero, outputs
    // finite and wrong --, e = 254 as -inf, e = 255 (a sample of inf / nan) as -2^127.  Clamped, such a tile is scaled by
    // 2^-126: its largest values overflow the f16 and are recomputed by the fp32 chain, as any out-of-range operand is.
    return k > 127 ? 127 : k < -126 ? -126 : k;
}
__device__ __forceinline__ float scale_from_exp(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }
__device__ __forceinline__ scale2_t scale2_from_exp(int k)   // {2^k, 2^k}
{
    const unsigned b = (unsigned)(127 + k) << 23;
    return ((scale2_t)b << 32) | b;
}

} // namespace f16s
} // namespace fn2
