// Probe (profiling aid): how long does it take to write the cost volume (8 x 441 x 48 x 64 fp32 = 43.4 MB) from 256
// workgroups of 16 waves, by store pattern?
//   0  linear: workgroup b writes a contiguous 1/256 of the buffer, 16 B per lane
//   1  the forward kernel's rows: a task (n, py, rg, u) writes 16 planes x 21 = 336 rows of 256 B:
//      row (tj = 4u + bi - ai, ti, y = 2 (4 rg + ai) + py); 4 rows per wave instruction, 16 B per lane
//   2  as 1 with non-temporal stores
//   3  as 0 with non-temporal stores
// Build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(1024, 4) void k(float *out, int rounds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 v = {1.0f, 2.0f, 3.0f, (float)lane};
    if (PAT == 0 || PAT == 3) {
        const size_t total = (size_t)8 * 441 * 48 * 64 / 4;          // f4 elements
        const size_t per = total / 256;
        f4 *dst = reinterpret_cast<f4 *>(out) + (size_t)blockIdx.x * per;
        for (size_t i = threadIdx.x; i < per; i += 1024) {
            if (PAT == 3) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
        }
        return;
    }
    // 576 tasks: n = task / 72, (py, rg, u) = rest; workgroup b takes tasks b, b + 256, b + 512 of a list ordered n-major
    // with the stream/XCD mapping of the kernel: n = b & 7
    const int n = blockIdx.x & 7, j = blockIdx.x >> 3;
    for (int r = 0; r < rounds; ++r) {
        const int t = j + 32 * r;
        if (t >= 72) break;
        const int u = t % 6, rg = (t / 6) % 6, py = t / 36;
        for (int i = 0; i < 6; ++i) {
            const int row = wave * 4 + (lane >> 4) + 64 * i;
            if (row >= 336) continue;
            const int pl = row / 21, ti = row - pl * 21, ai = pl >> 2, bi = pl & 3;
            const int tj = 4 * u + bi - ai;
            if (tj < 0 || tj >= 21) continue;
            const int y = 2 * (4 * rg + ai) + py;
            f4 *dst = reinterpret_cast<f4 *>(out + (((size_t)n * 441 + tj * 21 + ti) * 48 + y) * 64 + 4 * (lane & 15));
            if (PAT == 2) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
}

int main()
{
    float *out;
    const size_t bytes = (size_t)8 * 441 * 48 * 64 * 4;
    hipMalloc(&out, bytes);
    hipMemset(out, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 10; ++rep) {
            hipEventRecord(e0);
            switch (pat) {
            case 0: hipLaunchKernelGGL((k<0>), dim3(256), dim3(1024), 0, 0, out, 3); break;
            case 1: hipLaunchKernelGGL((k<1>), dim3(256), dim3(1024), 0, 0, out, 3); break;
            case 2: hipLaunchKernelGGL((k<2>), dim3(256), dim3(1024), 0, 0, out, 3); break;
            default: hipLaunchKernelGGL((k<3>), dim3(256), dim3(1024), 0, 0, out, 3); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; if (rep >= 2) sum += ms;
        }
        printf("store pattern %d: best %.1f us, mean %.1f us  (%.2f TB/s at best)\n", pat, best * 1e3, sum / 8 * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
