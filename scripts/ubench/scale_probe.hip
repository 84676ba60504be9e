// Probe (hardware semantics): v_pk_mul_f32 with an SGPR-pair scale, the asm v_readfirstlane of f16x2_split.h's scale_exp,
// and the block-scaled split: prints a few values and the number of mismatches against the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../flownet2-pytorch_amd/csrc/f16x2_split.h"
using namespace fn2::f16s;
__global__ void k(const float *x, float *o, unsigned *words, int *kout)
{
    __shared__ unsigned scl[2];
    if (threadIdx.x < 2) scl[threadIdx.x] = 0;
    __syncthreads();
    ExpStat st = {0u, 0u};
    exp_sample(st, __builtin_bit_cast(unsigned, x[2 * threadIdx.x]));
    exp_sample(st, __builtin_bit_cast(unsigned, x[2 * threadIdx.x + 1]));
    post_stat(scl, st, threadIdx.x & 63);
    __syncthreads();
    const int kk = scale_exp(scl);
    const scale2_t s2 = scale2_from_exp(kk);
    f2s v = {x[2 * threadIdx.x], x[2 * threadIdx.x + 1]};
    v = pk_scale(v, s2);
    unsigned h, l;
    split2(v[0], v[1], h, l);
    o[4 * threadIdx.x] = v[0]; o[4 * threadIdx.x + 1] = v[1];
    o[4 * threadIdx.x + 2] = __builtin_bit_cast(float, h); o[4 * threadIdx.x + 3] = __builtin_bit_cast(float, l);
    if (threadIdx.x == 0) { words[0] = scl[0]; words[1] = scl[1]; kout[0] = kk; }
}
int main()
{
    const int n = 256;
    std::vector<float> h(2 * n);
    for (int i = 0; i < 2 * n; ++i) h[i] = 1e-6f * (float)((i * 37 % 101) - 50) / 17.0f;
    float *d, *o; unsigned *w; int *kk;
    hipMalloc(&d, 2 * n * 4); hipMalloc(&o, 4 * n * 4); hipMalloc(&w, 8); hipMalloc(&kk, 4);
    hipMemcpy(d, h.data(), 2 * n * 4, hipMemcpyHostToDevice);
    k<<<1, n>>>(d, o, w, kk);
    std::vector<float> r(4 * n); unsigned hw[2]; int hk;
    hipMemcpy(r.data(), o, 4 * n * 4, hipMemcpyDeviceToHost); hipMemcpy(hw, w, 8, hipMemcpyDeviceToHost); hipMemcpy(&hk, kk, 4, hipMemcpyDeviceToHost);
    unsigned sum = 0, cnt = 0;
    for (int i = 0; i < 2 * n; ++i) { unsigned b; memcpy(&b, &h[i], 4); unsigned e = (b >> 23) & 255; sum += e; cnt += e != 0; }
    printf("device sum %u cnt %u k %d | host sum %u cnt %u mean exp %.2f\n", hw[0], hw[1], hk, sum, cnt, (double)sum / cnt);
    int bad = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < 2; ++j) if (r[4 * i + j] != ldexpf(h[2 * i + j], hk)) ++bad;
    printf("pk_mul mismatches: %d of %d; sample: x %.6e -> %.6e (expect %.6e), pair hi %.6e -> %.6e\n", bad, 2 * n, h[2], r[4], ldexpf(h[2], hk), h[3], r[5]);
    return 0;
}
