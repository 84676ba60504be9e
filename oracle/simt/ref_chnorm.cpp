/*
 * ref_chnorm.cpp -- runs the reference's channelnorm __global__ kernels on the CPU
 * (TEST INFRASTRUCTURE ONLY; see cuda_on_cpu.h).  FN2_REF_SLICE is the device code
 * of networks/channelnorm_package/channelnorm_kernel.cu (before the first host
 * launcher).  Launch geometry restates channelnorm_kernel.cu:98-129,:131-177.
 */
#include "cuda_on_cpu.h"
#include FN2_REF_SLICE

static long4 sz4(long a, long b, long c, long d) { return make_long4(a, b, c, d); }
static long4 st4(long b, long c, long d) { return make_long4(b * c * d, c * d, d, 1); }

template <typename T>
static int cn_fwd(const T *in, T *out, int B, int C, int H, int W)
{
    const int n = B * H * W;
    simt::launch(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_channelnorm_update_output<T>(n, in, sz4(B, C, H, W), st4(C, H, W), out, sz4(B, 1, H, W), st4(1, H, W), 2);
    });
    return 0;
}
template <typename T>
static int cn_bwd(const T *in, const T *out, const T *gout, T *gin, int B, int C, int H, int W)
{
    const int n = B * C * H * W;
    simt::launch(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_channelnorm_backward_input1<T>(n, in, sz4(B, C, H, W), st4(C, H, W), out, sz4(B, 1, H, W), st4(1, H, W),
                                              gout, sz4(B, 1, H, W), st4(1, H, W), gin, sz4(B, C, H, W), st4(C, H, W), 2);
    });
    return 0;
}
extern "C" {
int fn2ref_chnorm_fwd_f32(const float *in, float *out, int B, int C, int H, int W) { return cn_fwd<float>(in, out, B, C, H, W); }
int fn2ref_chnorm_fwd_f64(const double *in, double *out, int B, int C, int H, int W) { return cn_fwd<double>(in, out, B, C, H, W); }
int fn2ref_chnorm_bwd_f32(const float *in, const float *out, const float *go, float *gi, int B, int C, int H, int W) { return cn_bwd<float>(in, out, go, gi, B, C, H, W); }
int fn2ref_chnorm_bwd_f64(const double *in, const double *out, const double *go, double *gi, int B, int C, int H, int W) { return cn_bwd<double>(in, out, go, gi, B, C, H, W); }
}
