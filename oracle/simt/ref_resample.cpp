/*
 * ref_resample.cpp -- runs the reference's resample2d __global__ kernels on the
 * CPU (TEST INFRASTRUCTURE ONLY; see cuda_on_cpu.h).  FN2_REF_SLICE is the device
 * code of networks/resample2d_package/resample2d_kernel.cu (before the first host
 * launcher).  Launch geometry restates resample2d_kernel.cu:200-242,:244-323
 * (ceil(n/512) x 512 threads; float only); shapes follow resample2d.py:16-18,:31-32.
 */
#include "cuda_on_cpu.h"
#include FN2_REF_SLICE

static long4 sz4(long a, long b, long c, long d) { return make_long4(a, b, c, d); }
static long4 st4(long b, long c, long d) { return make_long4(b * c * d, c * d, d, 1); }

extern "C" {
int fn2ref_resample_fwd_f32(const float *img, const float *flow, float *out, int B, int C, int Hi, int Wi,
                            int H, int W, int ks, int bilinear)
{
    const int n = B * C * H * W;
    simt::launch(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_resample2d_update_output<float>(n, img, sz4(B, C, Hi, Wi), st4(C, Hi, Wi), flow, sz4(B, 2, H, W),
                                               st4(2, H, W), out, sz4(B, C, H, W), st4(C, H, W), ks, bilinear != 0);
    });
    return 0;
}
int fn2ref_resample_bwd_f32(const float *img, const float *flow, const float *gout, float *gimg, float *gflow,
                            int B, int C, int Hi, int Wi, int H, int W, int ks, int bilinear)
{
    memset(gimg, 0, sizeof(float) * (size_t)B * C * Hi * Wi);
    memset(gflow, 0, sizeof(float) * (size_t)B * 2 * H * W);
    int n = B * C * H * W;
    simt::launch(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_resample2d_backward_input1<float>(n, img, sz4(B, C, Hi, Wi), st4(C, Hi, Wi), flow, sz4(B, 2, H, W),
                                                 st4(2, H, W), gout, sz4(B, C, H, W), st4(C, H, W), gimg,
                                                 sz4(B, C, Hi, Wi), st4(C, Hi, Wi), ks, bilinear != 0);
    });
    n = B * 2 * H * W;
    simt::launch(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_resample2d_backward_input2<float>(n, img, sz4(B, C, Hi, Wi), st4(C, Hi, Wi), flow, sz4(B, 2, H, W),
                                                 st4(2, H, W), gout, sz4(B, C, H, W), st4(C, H, W), gflow,
                                                 sz4(B, 2, H, W), st4(2, H, W), ks, bilinear != 0);
    });
    return 0;
}
}
