// Microbenchmark: VALU cost of the two-term f16 split of a pair of fp32 values (correlation_f16x2*.hip), with and without
// the block scale of round 3.
//   0: unscaled  cvt_pk, fma_mix, fma_mix, cvt_pk                          (round 2)
//   1: scaled    fma_mixlo, fma_mixhi, fma_mixlo, fma_mixhi  (scale = SGPR)  (f16x2_split.h, first version)
//   2: scaled    v_pk_mul_f32 (both values) + variant 0
//   3: scaled    v_mul, v_mul + variant 0
//   4: scaled    cvt_pk of (x*s) via fma_mix_f32 x2 -> cvt_pk ; residual: fma_mix x2 (x*s - h) -> cvt_pk   (6 ops)
// Build: hipcc --offload-arch=gfx950 -O3 -o f16_split_rate f16_split_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, h2)); }
__device__ __forceinline__ float resid_lo(unsigned hp, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x)); return r; }
__device__ __forceinline__ float resid_hi(unsigned hp, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x)); return r; }
template <int V> __device__ __forceinline__ void split(float x0, float x1, float s, unsigned &h, unsigned &l)
{
    if (V == 1) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "s"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "s"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "s"(s), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "s"(s), "v"(h));
        return;
    }
    if (V == 2) { f2 v = {x0, x1}; f2 sc = {s, s}; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "s"(sc)); x0 = v[0]; x1 = v[1]; }
    if (V == 3) { x0 *= s; x1 *= s; }
    h = pk_f16(x0, x1);
    l = pk_f16(resid_lo(h, x0), resid_hi(h, x1));
}
template <int V> __global__ void rate(const float *in, unsigned *out, int iters, float s)
{
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + 256 * i];
    s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h, l;
            split<V>(v[2 * i], v[2 * i + 1], s, h, l);
            acc ^= h + l;                                                             // 2 bookkeeping int ops per pair
            v[2 * i] = __uint_as_float(__float_as_uint(v[2 * i]) ^ (l & 0x7fu));    // + 2: loop-carried, keeps the split live
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    float *d; unsigned *o;
    hipMalloc(&d, 2048 * 4); hipMalloc(&o, 1024 * 256 * 4);
    hipMemset(d, 0x3c, 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int wps = 1; wps <= 4; wps *= 2)      // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        for (int V = 0; V < 4; ++V) {
            const int blocks = 256 * wps;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (V) {
                case 0: rate<0><<<blocks, 256>>>(d, o, iters, 4.0f); break;
                case 1: rate<1><<<blocks, 256>>>(d, o, iters, 4.0f); break;
                case 2: rate<2><<<blocks, 256>>>(d, o, iters, 4.0f); break;
                default: rate<3><<<blocks, 256>>>(d, o, iters, 4.0f); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("waves/SIMD %d variant %d: %.3f ms -> %.2f clk per pair per SIMD (incl. 4 bookkeeping ops)\n", wps, V, ms, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 4));
        }
    return 0;
}
