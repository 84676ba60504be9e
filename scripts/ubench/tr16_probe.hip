// Probe (profiling aid, not product code): semantics of ds_read_b64_tr_b16 with per-lane addresses, f16 subnormals
// in v_mfma_f32_16x16x32_f16, buffer-load range checking with a scalar offset, and cache-resident streaming rates.
// Build: hipcc --offload-arch=gfx950 -O3 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define LDS3(T) __attribute__((address_space(3))) T

__global__ void tr_probe(short *out, int mode)
{
    __shared__ __attribute__((aligned(16))) short lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    // mode 0: lane l supplies chunk l (elements 4l..4l+3); mode 1: lane l supplies chunk perm(l) = (l*5+3) % 64
    // mode 2: chunk index 64 + ((l & 15) >> 2) * 72 + (l & 3) * 2 + (l >> 4) * 288   (kernel-like strides, in chunks)
    int chunk = mode == 0 ? l : mode == 1 ? (l * 5 + 3) % 64 : ((l & 15) >> 2) * 72 / 2 + (l & 3) + (l >> 4) * 144;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS3(s4) *)(lds + 4 * chunk));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

__global__ void denorm_probe(float *out)
{
    const int l = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    // A[row = l&15][k = 8*(l>>4) + i], B[k][col = l&15]: put one subnormal f16 (2^-20) in A[0][0], 1024 in B[0][0..15]
    if (l == 0) a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);   // 16 * 2^-24 = 2^-20
    if ((l >> 4) == 0) b[0] = (_Float16)1024.0f;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void oob_probe(const float *src, float *out, unsigned nbytes)
{
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, nbytes, 0x00020000);
    unsigned voff = (l & 1) ? 0x80000000u : (unsigned)(l * 16);
    u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 1024, 0);       // soffset 1024 bytes
    u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + 16), 1024, 0);
    out[l * 2] = __builtin_bit_cast(f4, v)[0];
    out[l * 2 + 1] = __builtin_bit_cast(f4, w)[0];
}

__global__ __launch_bounds__(256) void stream_k(const f4 *src, f4 *dst, size_t n4, int mode)
{
    // mode 1 read only, 2 write only, 3 copy
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (mode & 1) { f4 v = src[i]; if (mode == 1) acc += v; else dst[i] = v; }
        else dst[i] = (f4){1.0f, 2.0f, 3.0f, (float)i};
    }
    if (mode == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) dst[0] = acc;
}

int main()
{
    short *d; hipMalloc(&d, 256 * 2);
    std::vector<short> h(256);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("tr16 mode %d (lane: 4 element indices)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  %2d: %4d %4d %4d %4d", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
            if ((l & 3) == 3) printf("\n");
        }
        // expected by the kernel's assumption: result[lane i of group g][j] = element (i&3) of the chunk supplied by lane 16g + 4j + (i>>2)
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int sl = (l & ~15) + 4 * j + ((l & 15) >> 2);
                const int chunk = mode == 0 ? sl : mode == 1 ? (sl * 5 + 3) % 64 : ((sl & 15) >> 2) * 72 / 2 + (sl & 3) + (sl >> 4) * 144;
                if (h[4 * l + j] != 4 * chunk + (l & 3)) ++bad;
            }
        printf("  assumed mapping mismatches: %d\n", bad);
    }
    float *f; hipMalloc(&f, 4096);
    std::vector<float> hf(256);
    hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, f);
    hipMemcpy(hf.data(), f, 1024, hipMemcpyDeviceToHost);
    printf("mfma f16 subnormal: D[0][0] = %g (expect 2^-20*1024 = %g), D[0][5] = %g, D[1][0] = %g\n", hf[0], 1024.0 / 1048576.0, hf[5 * 4], hf[1]);
    float *src; hipMalloc(&src, 1 << 20);
    std::vector<float> hs(1 << 18); for (size_t i = 0; i < hs.size(); ++i) hs[i] = (float)i;
    hipMemcpy(src, hs.data(), 1 << 20, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(oob_probe, dim3(1), dim3(64), 0, 0, src, f, 4096u);
    hipMemcpy(hf.data(), f, 512, hipMemcpyDeviceToHost);
    printf("buffer load, num_records 4096, soffset 1024: lane0 %g %g (expect 256 260)  lane1 %g %g (expect 0 0)  lane 62 %g %g (expect %g %g)\n",
           hf[0], hf[1], hf[2], hf[3], hf[124], hf[125], 256.0 + 62 * 4, 260.0 + 62 * 4);
    // cache-resident streaming
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {24, 48, 96, 512}) {
        const size_t n4 = mb * 1024 * 1024 / 16;
        f4 *a, *b; hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16);
        hipMemset(a, 0, n4 * 16); hipMemset(b, 0, n4 * 16);
        for (int mode = 1; mode <= 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(stream_k, dim3(2048), dim3(256), 0, 0, a, b, n4, mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (mode == 3 ? 2.0 : 1.0) * n4 * 16;
            printf("stream %4zu MB %s: %.1f us  %.2f TB/s\n", mb, mode == 1 ? "read " : mode == 2 ? "write" : "copy ", best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
        hipFree(a); hipFree(b);
    }
    return 0;
}
