// Microbenchmark: cost and exactness of the fp32 -> 3 x bf16 operand split used by corr_fwd_mfma_bf16x3.
//   variant 0: mask / subtract (v_and, v_sub, v_and, v_sub per value + 3 v_perm per pair)   = 11 VALU per pair
//   variant 1: v_dot2c_f32_bf16 residuals (r = x - bf16hi(x) in ONE op from the packed pair)  =  7 VALU per pair
// Build: hipcc --offload-arch=gfx950 -O3 -o split_rate split_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_hi16(float x, float y) {
    return __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
}
__device__ __forceinline__ void split_mask(float x, float y, unsigned &p0, unsigned &p1, unsigned &p2) {
    const float x0 = __uint_as_float(__float_as_uint(x) & 0xffff0000u), y0 = __uint_as_float(__float_as_uint(y) & 0xffff0000u);
    const float rx = x - x0, ry = y - y0;
    const float x1 = __uint_as_float(__float_as_uint(rx) & 0xffff0000u), y1 = __uint_as_float(__float_as_uint(ry) & 0xffff0000u);
    const float sx = rx - x1, sy = ry - y1;
    p0 = pack_hi16(x, y); p1 = pack_hi16(rx, ry); p2 = pack_hi16(sx, sy);
}
__device__ __forceinline__ void split_dot2(float x, float y, unsigned &p0, unsigned &p1, unsigned &p2) {
    const bf2 selx = __builtin_bit_cast(bf2, 0x0000bf80u), sely = __builtin_bit_cast(bf2, 0xbf800000u);
    p0 = pack_hi16(x, y);
    const float rx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p0), selx, x, false);
    const float ry = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p0), sely, y, false);
    p1 = pack_hi16(rx, ry);
    const float sx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p1), selx, rx, false);
    const float sy = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p1), sely, ry, false);
    p2 = pack_hi16(sx, sy);
}
template <int V> __global__ void rate(const float *in, unsigned *out, int iters) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + 256 * i];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned p0, p1, p2;
            if (V == 0) split_mask(v[2 * i], v[2 * i + 1], p0, p1, p2); else split_dot2(v[2 * i], v[2 * i + 1], p0, p1, p2);
            acc ^= p0 + p1 + p2;                                  // 3 extra int ops per pair in both variants
            v[2 * i] = __uint_as_float(__float_as_uint(v[2 * i]) ^ (p2 & 0x7fu));      // loop-carried, keeps the split live
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int V> __global__ void exact(const float *in, unsigned *p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned p0, p1, p2;
    if (V == 0) split_mask(in[2 * i], in[2 * i + 1], p0, p1, p2); else split_dot2(in[2 * i], in[2 * i + 1], p0, p1, p2);
    p[3 * i] = p0; p[3 * i + 1] = p1; p[3 * i + 2] = p2;
}
static float bf(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int m = i & 7;
        float f = (rand() / (float)RAND_MAX - 0.5f) * 4.0f;
        if (m == 1) f *= 1e-3f; if (m == 2) f *= 1e4f; if (m == 3) f *= 1e-20f; if (m == 4) f *= 1e-37f; if (m == 5) f *= 3e-41f / 1e-0f * 1e-0f;
        if (i < 8) f = (i == 0) ? 0.0f : (i == 1) ? -0.0f : (i == 2) ? 1.0f : (i == 3) ? -1.0f : f;
        h[i] = f;
    }
    float *d; unsigned *dp;
    hipMalloc(&d, n * 4); hipMalloc(&dp, (size_t)n / 2 * 3 * 4 + 1024 * 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> hp(n / 2 * 3);
    for (int V = 0; V < 2; ++V) {
        if (V == 0) exact<0><<<n / 2 / 256, 256>>>(d, dp, n); else exact<1><<<n / 2 / 256, 256>>>(d, dp, n);
        hipMemcpy(hp.data(), dp, hp.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0, badden = 0;
        for (int i = 0; i < n / 2; ++i)
            for (int s = 0; s < 2; ++s) {
                const float x = h[2 * i + s];
                const int sh = s ? 16 : 0;
                const double sum = (double)bf((hp[3 * i] >> sh) & 0xffff) + (double)bf((hp[3 * i + 1] >> sh) & 0xffff) + (double)bf((hp[3 * i + 2] >> sh) & 0xffff);
                if (sum != (double)x) { if (fabsf(x) < 1e-30f) ++badden; else ++bad; }
            }
        printf("variant %d: inexact splits: %ld normal-range, %ld tiny (|x|<1e-30) of %d\n", V, bad, badden, n);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 1024;   // 4 blocks/CU x 4 waves = 4 waves per SIMD
    for (int V = 0; V < 2; ++V) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (V == 0) rate<0><<<blocks, 256>>>(d, dp, iters); else rate<1><<<blocks, 256>>>(d, dp, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: 4 waves x iters x 4 pairs
            const double clk = ms * 1e-3 * 2.4e9;
            if (rep == 2) printf("variant %d: %.3f ms -> %.2f clk per pair-split per SIMD (incl. 4 bookkeeping int ops)\n", V, ms, clk / (4.0 * iters * 4));
        }
    }
    return 0;
}
