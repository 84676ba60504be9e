// Probe (profiling aid): how long does the dispatcher take to start all waves of a persistent grid (one workgroup per CU)?
// For each configuration (threads per workgroup, LDS bytes, register budget) every wave stamps s_memtime at entry; we print
// per workgroup (last wave start - first wave start) and the kernel duration of an otherwise empty kernel.
// Build: hipcc --offload-arch=gfx950 -O3 launch_ramp.hip -o launch_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int T, int WPS, int LDS>
__global__ __launch_bounds__(T, WPS) void k(unsigned long long *out, int spin)
{
    __shared__ char lds[LDS];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) lds[0] = 1;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t0;
    // keep the register budget honest: nothing to do, the launch bounds set the allocation
    if (spin && lds[0] == 7) out[0] = 0;
}

template <int T, int WPS, int LDS>
void run(const char *name, unsigned long long *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> h(256 * 16);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(d, 0, 256 * 16 * 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<T, WPS, LDS>), dim3(256), dim3(T), 0, 0, d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    hipMemcpy(h.data(), d, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < T / 64; ++w) { lo = std::min(lo, h[b * 16 + w]); hi = std::max(hi, h[b * 16 + w]); }
        sum += (double)(hi - lo); mx = std::max(mx, (double)(hi - lo));
    }
    printf("%-44s waves %2d: in-workgroup start spread mean %7.0f max %7.0f ticks, empty kernel %.1f us\n", name, T / 64, sum / 256, mx, best * 1e3);
}

int main()
{
    unsigned long long *d; hipMalloc(&d, 256 * 16 * 8);
    run<1024, 4, 65536>("1024 thr, 128 regs, 64 KB LDS", d);
    run<1024, 4, 160000>("1024 thr, 128 regs, 160 KB LDS", d);
    run<1024, 4, 1024>("1024 thr, 128 regs, 1 KB LDS", d);
    run<768, 3, 160000>("768 thr, 168 regs, 160 KB LDS", d);
    run<512, 2, 160000>("512 thr, 256 regs, 160 KB LDS", d);
    run<512, 4, 160000>("512 thr, 128 regs, 160 KB LDS", d);
    run<512, 8, 1024>("512 thr, 64 regs, 1 KB LDS", d);
    run<256, 1, 160000>("256 thr, 512 regs, 160 KB LDS", d);
    run<256, 8, 1024>("256 thr, 64 regs, 1 KB LDS", d);
    return 0;
}
