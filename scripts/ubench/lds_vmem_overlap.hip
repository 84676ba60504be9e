// Probe (profiling aid): do LDS-fed f16 MFMA waves and global-load waves of the same CU overlap?
// 16 waves per workgroup, one workgroup per CU.  Waves 0-7 ("matrix"): loop of ds_read_b64_tr_b16 x 8 + 6 MFMA 16x16x32 f16.
// Waves 8-15 ("loaders"): stream an L2-resident buffer, 8 x 16 B per lane per iteration, either into VGPRs (consumed by
// xor) or by LDS-DMA.  Modes: 1 matrix only, 2 loaders only, 3 both.
// Build: hipcc --offload-arch=gfx950 -O3 lds_vmem_overlap.hip -o lds_vmem_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define LDS3(T) __attribute__((address_space(3))) T

template <int KIND>   // 0: buffer_load to VGPR, 1: global_load_lds
__global__ __launch_bounds__(1024, 4) void k(const float *src, float *sink, int mode, int nmat, int nload, size_t span_floats)
{
    __shared__ __attribute__((aligned(16))) char lds[98304];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 98304 / 4; i += 1024) reinterpret_cast<unsigned *>(lds)[i] = 0x3c003c00u;
    __syncthreads();
    if (wave < 8) {
        if (!(mode & 1)) return;
        f4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = (f4){0, 0, 0, 0};
        const char *base = lds + (wave & 3) * 8192 + ((lane >> 4) * 4 + ((lane & 15) >> 2)) * 288 + (lane & 3) * 8;
        for (int it = 0; it < nmat; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS3(s4) *)(base + j * 32));
                const s4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS3(s4) *)(base + j * 32 + 4608));
                const h8 a = __builtin_bit_cast(h8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc[j], 0, 0, 0);
                acc[(j + 1) % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc[(j + 1) % 6], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) sink[threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        const size_t wg_off = ((size_t)blockIdx.x * 8 + (wave - 8)) * 64 * 4;
        u4 accv = (u4)(0u);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (unsigned)(span_floats * 4), 0x00020000);
        for (int it = 0; it < nload; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                size_t off = (wg_off + (size_t)(it * 8 + q) * 65536 * 4 + (size_t)lane * 4) % span_floats;
                off &= ~(size_t)3;
                if (KIND == 0) accv ^= __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4), 0, 0);
                else __builtin_amdgcn_global_load_lds(src + off, (LDS3(void) *)(lds + 65536 + (wave - 8) * 2048 + (q & 1) * 1024), 16, 0, 0);
            }
        }
        if (KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (accv[0] + accv[1] + accv[2] + accv[3] == 12345u) sink[threadIdx.x] = 1.0f;
    }
}

int main(int argc, char **argv)
{
    const int nmat = argc > 1 ? atoi(argv[1]) : 400;     // x 8 MFMA + 8 tr reads per wave
    const int nload = argc > 2 ? atoi(argv[2]) : 100;    // x 8 KB per wave
    const size_t span = 8u << 20;                        // 8M floats = 32 MB (L2/MALL resident)
    float *src, *sink;
    hipMalloc(&src, span * 4); hipMalloc(&sink, 8192);
    hipMemset(src, 0, span * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int mode = 1; mode <= 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL((k<0>), dim3(256), dim3(1024), 0, 0, src, sink, mode, nmat, nload, span);
                else hipLaunchKernelGGL((k<1>), dim3(256), dim3(1024), 0, 0, src, sink, mode, nmat, nload, span);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("loads %s mode %d (%s): %.1f us   [matrix: %d x 8 MFMA/wave; loaders: %.1f MB/CU]\n", kind ? "lds-dma" : "vgpr   ", mode,
                   mode == 1 ? "matrix only" : mode == 2 ? "loads only " : "both       ", best * 1e3, nmat, nload * 8 * 8 / 1024.0);
        }
    return 0;
}
