// Microbenchmark (profiling aid): LDS atomic throughput on gfx950 -- ds_add_f32 vs ds_add_u32 (no return)
// vs ds_add_rtn_u32, random addresses in a 24 KB window vs conflict-free (lane-linear) addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, int PATTERN>
__global__ __launch_bounds__(512) void k(float *sink, int iters)
{
    __shared__ float w[6144];
    unsigned *wu = reinterpret_cast<unsigned *>(w);
    for (int i = threadIdx.x; i < 6144; i += 512) w[i] = 0;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const int a = PATTERN == 0 ? (int)((h >> 8) % 6144u) : (int)((threadIdx.x + it * 512) % 6144);
        if (KIND == 0) atomicAdd(&w[a], 1.0f);
        else if (KIND == 1) atomicAdd(&wu[a], 1u);
        else if (KIND == 2) acc += atomicAdd(&wu[a], 1u);
        else if (KIND == 3) {   // float add as a compare-and-swap loop
            unsigned old = wu[a], assumed;
            do { assumed = old; old = atomicCAS(&wu[a], assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
        } else if (KIND == 4) { // 64-bit integer add (fixed point), half as many cells
            atomicAdd(reinterpret_cast<unsigned long long *>(w) + (a >> 1), 12345ull);
        } else {                // double add (ds_add_f64), half as many cells
            atomicAdd(reinterpret_cast<double *>(w) + (a >> 1), 1.0);
        }
    }
    __syncthreads();
    if (w[threadIdx.x] == 12345.0f || acc == 0xdeadbeefu) sink[threadIdx.x] = w[threadIdx.x];
}
int main()
{
    float *sink; hipMalloc(&sink, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char *kn[6] = {"ds_add_f32 (no rtn)", "ds_add_u32 (no rtn)", "ds_add_rtn_u32", "f32 add via CAS loop", "ds_add_u64 (no rtn)", "ds_add_f64 (no rtn)"};
    for (int kind = 1; kind < 6; ++kind)
        for (int pat = 0; pat < 2; ++pat) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (kind == 0 && pat == 0) hipLaunchKernelGGL((k<0, 0>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 0 && pat == 1) hipLaunchKernelGGL((k<0, 1>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 1 && pat == 0) hipLaunchKernelGGL((k<1, 0>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 1 && pat == 1) hipLaunchKernelGGL((k<1, 1>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 2 && pat == 0) hipLaunchKernelGGL((k<2, 0>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 2 && pat == 1) hipLaunchKernelGGL((k<2, 1>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 3 && pat == 0) hipLaunchKernelGGL((k<3, 0>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 3 && pat == 1) hipLaunchKernelGGL((k<3, 1>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 4 && pat == 0) hipLaunchKernelGGL((k<4, 0>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 4 && pat == 1) hipLaunchKernelGGL((k<4, 1>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 5 && pat == 0) hipLaunchKernelGGL((k<5, 0>), dim3(512), dim3(512), 0, 0, sink, iters);
                if (kind == 5 && pat == 1) hipLaunchKernelGGL((k<5, 1>), dim3(512), dim3(512), 0, 0, sink, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double ops = 512.0 * 512 * iters;   // 2 workgroups per CU
            printf("%-22s %-10s: %.1f us  -> %.2f lane-atomics / clk / CU\n", kn[kind], pat ? "linear" : "random", best * 1e3,
                   ops / 256 / (best * 1e-3 * 2.4e9));
        }
    return 0;
}
