// Microbenchmark (profiling aid): do bf16 MFMA work and plain VALU work (v_and / v_sub / v_perm, the operand
// split of the bf16x3 kernel) from DIFFERENT waves of the same SIMD overlap?  8 waves per workgroup: waves 0-3
// run MFMAs, waves 4-7 run VALU.  Also: how many cycles does a wave64 VALU op take, and does one wave reach it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float __attribute__((ext_vector_type(4))) f4;
typedef short __attribute__((ext_vector_type(8))) bf8;

__global__ __launch_bounds__(512, 2) void k(float *sink, int mode, int nmfma, int nvalu, int valu_waves)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (!(mode & 1)) return;
        f4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = (f4){0, 0, 0, 0};
        bf8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (short)(lane + i); hb[i] = (short)(lane * 3 + i); }
        for (int it = 0; it < nmfma; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) sink[threadIdx.x] = s;
    } else {
        if (!(mode & 2) || wave >= 4 + valu_waves) return;
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = (float)(lane * 8 + i) * 1.0001f;
        unsigned acc = 0;
        for (int it = 0; it < nvalu; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // the split: and, sub, and, sub  (+ perm per pair)
                float h0 = __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
                float r1 = x[i] - h0;
                float h1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                float h2 = r1 - h1;
                x[i] = h2 + x[i];   // keeps the chain alive (1 extra add)
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc ^= __builtin_amdgcn_perm(__float_as_uint(x[2 * i + 1]), __float_as_uint(x[2 * i]), 0x07060302u);
        }
        if (acc == 0x12345678u) sink[threadIdx.x] = x[0];
    }
}

int main(int argc, char **argv)
{
    const int nmfma = argc > 1 ? atoi(argv[1]) : 800;
    const int nvalu = argc > 2 ? atoi(argv[2]) : 2000;
    float *sink; hipMalloc(&sink, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int vw = 4; vw >= 4; vw -= 4)
        for (int mode = 1; mode <= 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, sink, mode, nmfma, nvalu, vw);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            // per VALU wave: nvalu * (8*5 + 4*2) = 48 VALU ops per iteration
            printf("mode %d (%s): %.1f us | mfma ideal %.1f us (16 cyc each) | VALU: %.2f cycles/op at 2.4 GHz if alone\n", mode,
                   mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both", best * 1e3, nmfma * 12 * 16 / 2.4e3,
                   best * 1e-3 * 2.4e9 / (nvalu * 48.0));
        }
    return 0;
}
