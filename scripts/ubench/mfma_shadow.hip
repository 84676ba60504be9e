// Microbenchmark: do VALU instructions of the SAME wave hide in the shadow of its MFMAs on gfx950?  One wave per SIMD runs a
// chain-free stream of v_mfma_f32_16x16x32_f16 (4 accumulators) with k independent VALU ops (v_cvt_pk_f16_f32 / v_fma_mix /
// v_pk_mul_f32, the staging split) after each MFMA.  If the time is flat in k up to ~3, 12 of an MFMA's 16 cycles are free for
// the wave's own VALU work; if it grows by 4 cycles per op, MFMA and VALU issue are serialised (scripts/ubench/mfma_valu_overlap
// found that for DIFFERENT waves of one SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int K, int WAVES> __global__ __launch_bounds__(64 * WAVES) void k(float *sink, int iters)
{
    const int lane = threadIdx.x & 63;
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0, 0, 0, 0};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane + i); b[i] = (_Float16)(lane * 3 + i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane * 8 + i) * 1.0001f;
    unsigned junk = 0;
    float c0 = 1.0001f + lane * 1e-9f, c1 = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < K; ++v) {   // 8 independent VALU chains (v_fma_f32), none touches the MFMA's registers
                float &t = x[(i * K + v) & 7];
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(c0), "v"(c1));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f || junk == 0x12345u) sink[threadIdx.x] = s;
}
template <int K, int WAVES> static void run(float *sink, const char *what)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: WAVES/4 waves x iters x 4 MFMAs
    const double per = best * 1e-3 * 2.4e9 / ((WAVES / 4.0) * iters * 4.0);
    printf("%s: %d VALU per MFMA, %d waves/SIMD: %.1f cycles per (MFMA + %d VALU) per SIMD\n", what, K, WAVES / 4, per, K);
}
int main()
{
    float *sink; hipMalloc(&sink, 1 << 16);
    run<0, 4>(sink, "same wave"); run<1, 4>(sink, "same wave"); run<2, 4>(sink, "same wave"); run<3, 4>(sink, "same wave");
    run<4, 4>(sink, "same wave"); run<6, 4>(sink, "same wave");
    run<0, 8>(sink, "two waves"); run<2, 8>(sink, "two waves"); run<4, 8>(sink, "two waves"); run<6, 8>(sink, "two waves");
    run<0, 16>(sink, "four waves"); run<2, 16>(sink, "four waves"); run<4, 16>(sink, "four waves"); run<6, 16>(sink, "four waves");
    return 0;
}
