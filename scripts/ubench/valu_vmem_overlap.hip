// Microbenchmark (profiling aid): does plain VALU work (the bf16 operand split) or LDS read traffic of waves 0-3 overlap
// with LDS-DMA streaming (global_load_lds_dwordx4) issued by waves 4-7 of the same workgroup (w and w+4 share a SIMD)?
// Modes: 1 = compute only, 2 = DMA only, 3 = both.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_vmem_overlap valu_vmem_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float __attribute__((ext_vector_type(4))) f4;

template <int WORK>   // 0: VALU (and/sub/perm split chain), 1: ds_read_b128 stream, 2: VALU + ds_read_b128
__global__ __launch_bounds__(512, 2) void k(const float *src, float *sink, int mode, int ncomp, int nload, size_t span_floats, int lanes)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (!(mode & 1)) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = (float)(lane * 8 + i) * 1.0001f;
        unsigned acc = 0;
        f4 accf = (f4){0, 0, 0, 0};
        for (int it = 0; it < ncomp; ++it) {
            if (WORK != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) accf += *reinterpret_cast<const f4 *>(lds + ((lane * 4 + j * 256 + it * 4) & 8191));
            }
            if (WORK != 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x0 = __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
                    const float r1 = v[i] - x0;
                    const float x1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                    const float r2 = r1 - x1;
                    acc ^= __builtin_amdgcn_perm(__float_as_uint(x0), __float_as_uint(x1), 0x07060302u) + __float_as_uint(r2);
                    v[i] = __uint_as_float(__float_as_uint(v[i]) ^ (acc & 0x7f));
                }
            }
        }
        if (acc == 0x12345678u || accf[0] == 12345.678f) sink[threadIdx.x] = (float)acc + accf[1];
    } else {
        if (!(mode & 2)) return;
        const size_t wg_off = ((size_t)blockIdx.x * 4 + (wave - 4)) * 64 * 4;
        for (int it = 0; it < nload; ++it) {
            size_t off = (wg_off + (size_t)it * 262144 * 4 + (size_t)lane * 4) % span_floats;
            off &= ~(size_t)3;
            if (lane < lanes)
                __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(lds + 8192 + (wave - 4) * 2048 + (it & 7) * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lds[8192 + lane] == 12345.678f) sink[threadIdx.x] = lds[8192 + lane];
    }
}

int main(int argc, char **argv)
{
    const int ncomp = argc > 1 ? atoi(argv[1]) : 1500;
    const int nload = argc > 2 ? atoi(argv[2]) : 1200;  // 1 KB per wave-iteration
    const size_t span = 16u << 20;
    float *src, *sink;
    (void)hipMalloc(&src, span * 4);
    (void)hipMalloc(&sink, 4096);
    (void)hipMemset(src, 0, span * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int work = 0; work < 3; ++work)
        for (int lanes = 64; lanes >= 26; lanes -= 38)
        for (int mode = 1; mode <= 3; ++mode) {
            if (lanes != 64 && mode == 1) continue;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(e0);
                if (work == 0) hipLaunchKernelGGL((k<0>), dim3(256), dim3(512), 0, 0, src, sink, mode, ncomp, nload, span, lanes);
                if (work == 1) hipLaunchKernelGGL((k<1>), dim3(256), dim3(512), 0, 0, src, sink, mode, ncomp * 4, nload, span, lanes);
                if (work == 2) hipLaunchKernelGGL((k<2>), dim3(256), dim3(512), 0, 0, src, sink, mode, ncomp, nload, span, lanes);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("work %s dma-lanes %d mode %d (%s): %.1f us\n", work == 0 ? "valu     " : work == 1 ? "lds-read " : "valu+lds ", lanes, mode,
                   mode == 1 ? "compute only" : mode == 2 ? "dma only" : "both", best * 1e3);
        }
    return 0;
}
