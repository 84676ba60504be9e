/*
 * ref_corr.cpp -- runs the reference's correlation __global__ kernels on the CPU
 * (TEST INFRASTRUCTURE ONLY; see cuda_on_cpu.h).  FN2_REF_SLICE is the device-code
 * part of networks/correlation_package/correlation_cuda_kernel.cu (everything
 * before the first host launcher), produced at build time by build_ref.sh.
 * The launch geometry below restates the host launchers
 * (correlation_cuda_kernel.cu:383-415 forward, :491-554 backward) and the
 * resize_/fill_(0) of correlation_cuda.cc:36-42,:106-114.
 */
#include "cuda_on_cpu.h"
#include FN2_REF_SLICE


template <typename T>
static int corr_fwd(const T *in1, const T *in2, T *out, int B, int C, int H, int W,
                    int pad, int k, int md, int s1, int s2)
{
    const int kr = (k - 1) / 2, br = kr + md;
    const int pH = H + 2 * pad, pW = W + 2 * pad;
    const int nOut = ((md / s2) * 2 + 1) * ((md / s2) * 2 + 1);
    const int oH = (int)ceil((float)(pH - 2 * br) / (float)s1);
    const int oW = (int)ceil((float)(pW - 2 * br) / (float)s1);
    if (oH < 1 || oW < 1) return -1;
    std::vector<T> r1((size_t)B * pH * pW * C, T(0)), r2((size_t)B * pH * pW * C, T(0));
    memset(out, 0, sizeof(T) * (size_t)B * nOut * oH * oW);
    T *pr1 = r1.data(), *pr2 = r2.data();
    simt::launch(dim3(B, H, W), dim3(THREADS_PER_BLOCK), [&] { channels_first<T>(in1, pr1, C, H, W, pad); });
    simt::launch(dim3(B, H, W), dim3(THREADS_PER_BLOCK), [&] { channels_first<T>(in2, pr2, C, H, W, pad); });
    simt::launch(dim3(B, oH, oW), dim3(THREADS_PER_BLOCK), [&] {
        correlation_forward<T>(out, nOut, oH, oW, pr1, C, H, W, pr2, pad, k, md, s1, s2);
    });
    return 0;
}

template <typename T>
static int corr_bwd(const T *in1, const T *in2, const T *gout, T *g1, T *g2, int B, int C, int H, int W,
                    int pad, int k, int md, int s1, int s2)
{
    const int kr = (k - 1) / 2, br = kr + md;
    const int pH = H + 2 * pad, pW = W + 2 * pad;
    const int nOut = ((md / s2) * 2 + 1) * ((md / s2) * 2 + 1);
    const int oH = (int)ceil((float)(pH - 2 * br) / (float)s1);
    const int oW = (int)ceil((float)(pW - 2 * br) / (float)s1);
    if (oH < 1 || oW < 1) return -1;
    std::vector<T> r1((size_t)B * pH * pW * C, T(0)), r2((size_t)B * pH * pW * C, T(0));
    memset(g1, 0, sizeof(T) * (size_t)B * C * H * W);
    memset(g2, 0, sizeof(T) * (size_t)B * C * H * W);
    T *pr1 = r1.data(), *pr2 = r2.data();
    simt::launch(dim3(B, H, W), dim3(THREADS_PER_BLOCK), [&] { channels_first<T>(in1, pr1, C, H, W, pad); });
    simt::launch(dim3(B, H, W), dim3(THREADS_PER_BLOCK), [&] { channels_first<T>(in2, pr2, C, H, W, pad); });
    for (int n = 0; n < B; ++n)
        simt::launch(dim3(H, W, C), dim3(THREADS_PER_BLOCK), [&] {
            correlation_backward_input1<T>(n, g1, C, H, W, gout, nOut, oH, oW, pr2, pad, k, md, s1, s2);
        });
    for (int n = 0; n < B; ++n)
        simt::launch(dim3(H, W, C), dim3(THREADS_PER_BLOCK), [&] {
            correlation_backward_input2<T>(n, g2, C, H, W, gout, nOut, oH, oW, pr1, pad, k, md, s1, s2);
        });
    return 0;
}

extern "C" {
int fn2ref_corr_fwd_f32(const float *a, const float *b, float *o, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{ return corr_fwd<float>(a, b, o, B, C, H, W, pad, k, md, s1, s2); }
int fn2ref_corr_fwd_f64(const double *a, const double *b, double *o, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{ return corr_fwd<double>(a, b, o, B, C, H, W, pad, k, md, s1, s2); }
int fn2ref_corr_bwd_f32(const float *a, const float *b, const float *go, float *g1, float *g2, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{ return corr_bwd<float>(a, b, go, g1, g2, B, C, H, W, pad, k, md, s1, s2); }
int fn2ref_corr_bwd_f64(const double *a, const double *b, const double *go, double *g1, double *g2, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{ return corr_bwd<double>(a, b, go, g1, g2, B, C, H, W, pad, k, md, s1, s2); }
}
