// Microbenchmark (profiling aid): fp32 adds into LDS as compare-and-swap loops -- one chain at a time (the shipped
// lds_add_f32 of resample2d.hip) vs N independent chains in flight per lane (reads issued together, swaps issued together,
// only the failed ones repeated).  Random addresses in a 31 KB window (the backward kernel's accumulation window).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int CELLS = 7760;
template <int NCH, int NT>
__global__ __launch_bounds__(NT) void k(float *sink, int iters)
{
    __shared__ float w[CELLS];
    unsigned *wu = reinterpret_cast<unsigned *>(w);
    for (int i = threadIdx.x; i < CELLS; i += NT) w[i] = 0;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int it = 0; it < iters; ++it) {
        int a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; a[j] = (int)((h >> 8) % (unsigned)CELLS); }
        if (NCH == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned old = wu[a[j]], assumed;
                do { assumed = old; old = atomicCAS(&wu[a[j]], assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; g += NCH) {
                unsigned old[NCH];
                bool done[NCH];
#pragma unroll
                for (int j = 0; j < NCH; ++j) { old[j] = wu[a[g + j]]; done[j] = false; }
                bool all;
                do {
                    unsigned got[NCH];
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
                        if (!done[j]) got[j] = atomicCAS(&wu[a[g + j]], old[j], __float_as_uint(__uint_as_float(old[j]) + 1.0f));
                    all = true;
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
                        if (!done[j]) { done[j] = got[j] == old[j]; old[j] = got[j]; all = all && done[j]; }
                } while (!all);
            }
        }
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < CELLS; i += NT) s += w[i];
    sink[blockIdx.x * NT + threadIdx.x] = s;
}
template <int NCH, int NT>
void run(float *sink, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NCH, NT>), dim3(512), dim3(NT), 0, 0, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // check: every add landed
    static float host[512 * 1024];
    hipMemcpy(host, sink, sizeof(float) * 512 * NT, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int i = 0; i < NT; ++i) tot += host[i];
    const double ops = 512.0 * NT * iters * 4;
    printf("%-28s %4d threads x 2 per CU: %.1f us -> %.2f lane-adds / clk / CU   (block 0 sum %.0f, expected %.0f)\n", name, NT, best * 1e3,
           ops / 256 / (best * 1e-3 * 2.4e9), tot, (double)NT * iters * 4);
}
int main()
{
    float *sink; hipMalloc(&sink, sizeof(float) * 512 * 1024);
    run<1, 512>(sink, "one chain at a time");
    run<2, 512>(sink, "two chains in flight");
    run<4, 512>(sink, "four chains in flight");
    run<1, 1024>(sink, "one chain at a time");
    run<2, 1024>(sink, "two chains in flight");
    run<4, 1024>(sink, "four chains in flight");
    return 0;
}
