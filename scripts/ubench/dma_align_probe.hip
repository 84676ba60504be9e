// Probe: does buffer_load_dwordx4 ... lds (LDS-DMA, 16 B per lane) accept an LDS base (M0) that is only 4- or 8-byte
// aligned?  Copies 1 KB to lds + ofs for ofs = 0, 4, 8, 12, 16 and checks the bytes.
// Build: hipcc --offload-arch=gfx950 -O3 dma_align_probe.hip -o dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define LDS3(T) __attribute__((address_space(3))) T
template <int OFS>
__global__ void k(const float *src, float *dst)
{
    __shared__ __attribute__((aligned(16))) char lds[4096];
    for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<float *>(lds)[i] = -1.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS3(void) *)(lds + 1024 + OFS), 16, (int)(threadIdx.x * 16), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) dst[i] = reinterpret_cast<float *>(lds)[i];
}
int main()
{
    float *src, *dst, h[1024], hs[1024];
    hipMalloc(&src, 4096); hipMalloc(&dst, 4096);
    for (int i = 0; i < 1024; ++i) hs[i] = (float)i;
    hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
    for (int ofs = 0; ofs <= 16; ofs += 4) {
        switch (ofs) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, src, dst); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, src, dst); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, src, dst); break;
        case 12: hipLaunchKernelGGL(k<12>, dim3(1), dim3(64), 0, 0, src, dst); break;
        default: hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, src, dst); break;
        }
        hipMemcpy(h, dst, 4096, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < 256; ++i) { if (h[256 + ofs / 4 + i] != (float)i) { ++bad; if (first < 0) first = i; } }
        printf("M0 offset %2d: %d of 256 dwords wrong (first %d: got %g %g %g %g %g); dword before %g after %g\n", ofs, bad, first,
               first >= 0 ? h[256 + ofs / 4 + first] : 0.f, h[256 + ofs / 4 + 1], h[256 + ofs / 4 + 2], h[256 + ofs / 4 + 3], h[256 + ofs / 4 + 4],
               h[256 + ofs / 4 - 1], h[256 + ofs / 4 + 256]);
    }
    return 0;
}
