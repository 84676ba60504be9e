// Probe (profiling aid): L2 -> CU load throughput per CU for different lane -> address patterns of buffer_load_b128.
// One 16-wave workgroup per CU; the 32 workgroups of an XCD (blockIdx % 8) stream the same 2 MB region, so everything after
// the first touch is an L2 hit.  Patterns:
//   0  16 B per lane, lanes contiguous (1 KB per instruction)
//   1  the staging pattern of correlation_f16x2: a lane owns 32 B (two loads, +0 and +16), 4 lanes = 128 B of a row,
//      rows of 256 B 512 B apart, channels 12 KB apart
//   2  16 lanes x 16 B = one 256-B row per 16 lanes, 4 rows (512 B apart) per instruction
//   3  as 1 but the two halves swapped in time: all +0 loads of 4 items first, then the +16 loads
//   4  8 B per lane (b64), lanes contiguous
// Build: hipcc --offload-arch=gfx950 -O3 load_pattern.hip -o load_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <int PAT>
__global__ __launch_bounds__(1024, 4) void k(const float *src, float *sink, int iters, int nwaves)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const unsigned region = 2u << 20;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src) + (size_t)(blockIdx.x & 7) * (region / 4), 0, region, 0x00020000);
    u4 acc = (u4)(0u);
    // a "step" = 64 KB per workgroup (as one channel step of the forward kernel): per wave 8 KB = 8 instructions of 1 KB
    unsigned vo;
    if (PAT == 0) vo = lane * 16;
    else if (PAT == 1 || PAT == 3) vo = (lane >> 5) * 12288 + ((lane >> 2) & 3) * 512 + ((lane & 3) + 4 * ((lane >> 4) & 1)) * 32;
    else if (PAT == 2) vo = (lane >> 4) * 512 + (lane & 15) * 16;
    else vo = lane * 8;
    const unsigned wbase = (unsigned)((blockIdx.x >> 3) * 65536 + wave * 4096) % region;
    for (int it = 0; it < iters; ++it) {
        const unsigned so = (wbase + (unsigned)it * 262144u) % (region - 65536);
        if (PAT == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)(so + q * 1024), 0);
        } else if (PAT == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)(so + q * 2048), 0);
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo + 16), (int)(so + q * 2048), 0);
            }
        } else if (PAT == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc ^= __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)(so + q * 2048), 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc ^= __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo + 16), (int)(so + q * 2048), 0);
        } else if (PAT == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)(so + (q & 1) * 256 + (q >> 1) * 2048), 0);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vo, (int)(so + q * 512), 0);
                acc[0] ^= v[0]; acc[1] ^= v[1];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345u) sink[threadIdx.x] = 1.0f;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    float *src, *sink;
    hipMalloc(&src, 16u << 20); hipMalloc(&sink, 8192);
    hipMemset(src, 0, 16u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    for (int nw = 4; nw <= 16; nw *= 2)
        for (int pat = 0; pat < 5; ++pat) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                switch (pat) {
                case 0: hipLaunchKernelGGL((k<0>), dim3(256), dim3(1024), 0, 0, src, sink, iters, nw); break;
                case 1: hipLaunchKernelGGL((k<1>), dim3(256), dim3(1024), 0, 0, src, sink, iters, nw); break;
                case 2: hipLaunchKernelGGL((k<2>), dim3(256), dim3(1024), 0, 0, src, sink, iters, nw); break;
                case 3: hipLaunchKernelGGL((k<3>), dim3(256), dim3(1024), 0, 0, src, sink, iters, nw); break;
                default: hipLaunchKernelGGL((k<4>), dim3(256), dim3(1024), 0, 0, src, sink, iters, nw); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes_cu = (double)iters * nw * 8192.0;
            printf("waves %2d pattern %d: %8.1f us  %.1f GB/s per CU  (%.1f B/clk at %.2f GHz)  chip %.2f TB/s\n", nw, pat, best * 1e3,
                   bytes_cu / (best * 1e-3) / 1e9, bytes_cu / (best * 1e-3) / (clk * 1e3), clk / 1e6, bytes_cu * 256 / (best * 1e-3) / 1e12);
        }
    return 0;
}
