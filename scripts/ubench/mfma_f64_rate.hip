// What does v_mfma_f64_16x16x4_f64 sustain?  Register-only stream: every wave runs ITERS x 8 MFMAs on 8 independent accumulators;
// 1, 2 and 4 waves per SIMD on every CU.   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_f64_rate scripts/ubench/mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void stream(double *sink, int iters)
{
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
}
int main()
{
    double *sink; hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps, blocks = 256;       // one workgroup per CU, wps waves per SIMD
        stream<<<blocks, threads>>>(sink, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0); stream<<<blocks, threads>>>(sink, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * (threads / 64) * iters * 8 * 2048.0;
        printf("%d wave(s) per SIMD: %.1f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wps, flop / (ms * 1e-3) / 1e12,
               ms * 1e-3 * 2.4e9 / ((double)wps * iters * 8));
    }
    return 0;
}
