// Microbenchmark (profiling aid, not part of the product): do MFMA work and global->VGPR / global->LDS traffic
// overlap on one CU when they come from DIFFERENT waves?  Each workgroup has 8 waves: waves 0-3 run a pure
// v_mfma_f32_16x16x4_f32 loop, waves 4-7 stream an L2-resident buffer.  Modes: 1 = MFMA only, 2 = loads only,
// 3 = both.  Build: hipcc --offload-arch=gfx950 -O3 mfma_vmem_overlap.hip -o mfma_vmem_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float __attribute__((ext_vector_type(4))) f4;

typedef short __attribute__((ext_vector_type(8))) bf8;
template <int KIND, int MT>   // KIND 0: global_load_dwordx4 to VGPR, 1: global_load_lds_dwordx4;  MT 0: f32 16x16x4, 1: bf16 16x16x32
__global__ __launch_bounds__(512, 2) void k(const float *src, float *sink, int mode, int nmfma, int nload, size_t span_floats, int prio)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (!(mode & 1)) return;
        f4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = (f4){0, 0, 0, 0};
        float a = (float)lane, b = (float)(lane ^ 5);
        bf8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (short)(lane + i); hb[i] = (short)(lane * 3 + i); }
        for (int it = 0; it < nmfma; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                if (MT == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) sink[threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        if (prio) __builtin_amdgcn_s_setprio(3);
        const size_t wg_off = ((size_t)blockIdx.x * 4 + (wave - 4)) * 64 * 4;
        f4 accv = (f4){0, 0, 0, 0};
        for (int it = 0; it < nload; ++it) {
            size_t off = (wg_off + (size_t)it * 262144 * 4 + (size_t)lane * 4) % span_floats;
            off &= ~(size_t)3;
            if (KIND == 0) {
                f4 v = *reinterpret_cast<const f4 *>(src + off);
                accv += v;
            } else {
                __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(lds + (wave - 4) * 2048 + (it & 7) * 256), 16, 0, 0);
            }
        }
        if (KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (accv[0] + accv[1] + accv[2] + accv[3] == 12345.678f) sink[threadIdx.x] = accv[0];
        if (KIND == 1 && lds[lane] == 12345.678f) sink[threadIdx.x] = lds[lane];
    }
}

int main(int argc, char **argv)
{
    const int nmfma = argc > 1 ? atoi(argv[1]) : 400;   // x12 MFMAs per wave
    const int nload = argc > 2 ? atoi(argv[2]) : 1200;  // 1 KB per wave-iteration
    const size_t span = 16u << 20;                      // 64 MB of floats? no: 16M floats = 64 MB (MALL-resident)
    float *src, *sink;
    hipMalloc(&src, span * 4);
    hipMalloc(&sink, 4096);
    hipMemset(src, 0, span * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mt = 0; mt < 2; ++mt)
    for (int prio = 0; prio < 2; ++prio)
    for (int kind = 0; kind < 2; ++kind)
        for (int mode = 1; mode <= 3; ++mode) {
            if (prio && mode != 3) continue;
            float best = 1e9;
            const int nm = mt ? nmfma * 2 : nmfma;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (kind == 0 && mt == 0) hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(512), 0, 0, src, sink, mode, nm, nload, span, prio);
                if (kind == 1 && mt == 0) hipLaunchKernelGGL((k<1, 0>), dim3(256), dim3(512), 0, 0, src, sink, mode, nm, nload, span, prio);
                if (kind == 0 && mt == 1) hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(512), 0, 0, src, sink, mode, nm, nload, span, prio);
                if (kind == 1 && mt == 1) hipLaunchKernelGGL((k<1, 1>), dim3(256), dim3(512), 0, 0, src, sink, mode, nm, nload, span, prio);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("mfma %s prio %d kind %s mode %d (%s): %.1f us\n", mt ? "bf16 16x16x32" : "f32  16x16x4 ", prio, kind ? "lds-dma" : "vgpr   ",
                   mode, mode == 1 ? "mfma only" : mode == 2 ? "loads only" : "both", best * 1e3);
        }
    return 0;
}
