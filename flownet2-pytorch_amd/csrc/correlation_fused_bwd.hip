// correlation_fused_bwd.hip -- "next" row N1, training half: the backward pass of
//     cat((conv_redir, LeakyReLU_s(Correlation(in1, in2))), 1)                                   (FlowNetC.py:86-87, :92)
// for the correlation branch.  autograd runs three passes around the reference's layer: the slice of the concat gradient is
// copied out (.contiguous()), multiplied by the activation's derivative (leaky_relu_backward: a second read of the 43 MB volume
// plus the saved output), and handed to correlation_backward.  Here ONE streaming pass reads the gradient straight from its
// slice of the concat gradient (batch stride) and the forward's stored output from ITS slice of the concat buffer, applies
// the derivative -- g where the stored output is positive, s * g elsewhere: the output of LeakyReLU_s (s > 0) has the sign
// of its argument, so no mask tensor and no pre-activation copy are kept --, and writes the contiguous masked gradient the
// correlation backward kernels read (correlation_cuda_kernel.cu:150-334, here correlation_f16x2_bwd.hip and its fallbacks).
// The bits are those of autograd's composition: the same fp32 product s * g, the same kernels afterwards.
#include "corr_params.h"
#include <type_traits>

namespace fn2 {
namespace fb {

typedef float f4 __attribute__((ext_vector_type(4)));

// out / grad: slices of NCHW buffers, n_item contiguous elements per batch item, `obs` / `gbs` elements between items
template <class T>
__global__ __launch_bounds__(256) void mask_slice_kernel(const T *__restrict__ grad, long gbs, const T *__restrict__ out, long obs,
                                                         T *__restrict__ dst, long n_item, int B, float slope)
{
    const long total = (long)B * n_item;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / n_item, r = i - n * n_item;
        // the test in T (a stored double below the float range must not read as zero), the product in fp32 for half and
        // float tensors (what leaky_relu_backward computes) and in double for double ones
        typedef typename std::conditional<std::is_same<T, double>::value, double, float>::type op_t;
        const T o = out[n * obs + r];
        const op_t g = (op_t)grad[n * gbs + r];
        dst[i] = (T)(o > (T)0 ? g : g * (op_t)slope);
    }
}

// fp32, everything 16-byte aligned, n_item % 4 == 0: four elements per lane
__global__ __launch_bounds__(256) void mask_slice_f32x4(const float *__restrict__ grad, long gbs, const float *__restrict__ out, long obs,
                                                        float *__restrict__ dst, long n_item4, int B, float slope)
{
    const long total = (long)B * n_item4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / n_item4, r = i - n * n_item4;
        const f4 g = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(grad + n * gbs) + r);
        const f4 o = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(out + n * obs) + r);
        f4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = o[e] > 0.0f ? g[e] : g[e] * slope;
        reinterpret_cast<f4 *>(dst)[i] = m;   // re-read right away by the backward kernel: a plain store
    }
}

} // namespace fb
} // namespace fn2

extern "C" size_t fn2_correlation_backward_fused_workspace_bytes(int dtype, int B, int H, int W, int pad_size, int kernel_size,
                                                                 int max_displacement, int stride1, int stride2)
{
    int nOut = 0, oH = 0, oW = 0;
    if (fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut, &oH, &oW) != FN2_OK) return 0;
    return (size_t)B * nOut * oH * oW * fn2::dtype_size(dtype);
}

extern "C" int fn2_correlation_backward_fused(const void *in1, const void *in2, const void *out_act, int64_t out_batch_stride,
                                              const void *grad_cat, int64_t grad_batch_stride, float negative_slope,
                                              void *workspace, size_t workspace_bytes, void *grad_in1, void *grad_in2,
                                              int dtype, int B, int C, int H, int W, int pad_size, int kernel_size,
                                              int max_displacement, int stride1, int stride2, int algo, void *stream)
{
    using namespace fn2;
    const size_t es = dtype_size(dtype);
    if (!es) return FN2_EDTYPE;
    int nOut = 0, oH = 0, oW = 0;
    int rc = fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut, &oH, &oW);
    if (rc != FN2_OK) return rc;
    if (B < 0 || C < 1) return FN2_EINVAL;
    if (stride1 != 1) return FN2_EUNSUPPORTED;                  // as fn2_correlation_backward
    const long n_item = (long)nOut * oH * oW;
    // the derivative is read off the sign of the stored output: only an increasing activation (slope > 0) keeps that sign
    if (!(negative_slope > 0.0f) || out_batch_stride < n_item || grad_batch_stride < n_item) return FN2_EINVAL;
    if (B == 0) return FN2_OK;
    if (!in1 || !in2 || !out_act || !grad_cat || !workspace || !grad_in1 || !grad_in2) return FN2_EINVAL;
    if (workspace_bytes < (size_t)B * n_item * es) return FN2_EINVAL;
    if (!aligned(out_act, es) || !aligned(grad_cat, es) || !aligned(workspace, es)) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long total = (long)B * n_item;
    if (negative_slope == 1.0f && grad_batch_stride == n_item)   // identity activation on a contiguous gradient: nothing to do
        return fn2_correlation_backward_ex(in1, in2, grad_cat, grad_in1, grad_in2, dtype, B, C, H, W, pad_size, kernel_size,
                                           max_displacement, stride1, stride2, algo, stream);
    if (dtype == FN2_F32 && n_item % 4 == 0 && out_batch_stride % 4 == 0 && grad_batch_stride % 4 == 0 && aligned(out_act, 16) &&
        aligned(grad_cat, 16) && aligned(workspace, 16)) {
        long blocks = (total / 4 + 255) / 256;
        if (blocks > 256L * 16) blocks = 256L * 16;
        hipLaunchKernelGGL(fb::mask_slice_f32x4, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float *>(grad_cat),
                           (long)grad_batch_stride, static_cast<const float *>(out_act), (long)out_batch_stride,
                           static_cast<float *>(workspace), n_item / 4, B, negative_slope);
    } else {
        long blocks = (total + 255) / 256;
        if (blocks > 256L * 16) blocks = 256L * 16;
#define FN2_MASK(T)                                                                                                           \
    hipLaunchKernelGGL((fb::mask_slice_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const T *>(grad_cat), \
                       (long)grad_batch_stride, static_cast<const T *>(out_act), (long)out_batch_stride, static_cast<T *>(workspace), \
                       n_item, B, negative_slope)
        if (dtype == FN2_F32) FN2_MASK(float);
        else if (dtype == FN2_F16) FN2_MASK(half_t);
        else FN2_MASK(double);
#undef FN2_MASK
    }
    rc = launch_status();
    if (rc != FN2_OK) return rc;
    return fn2_correlation_backward_ex(in1, in2, workspace, grad_in1, grad_in2, dtype, B, C, H, W, pad_size, kernel_size,
                                       max_displacement, stride1, stride2, algo, stream);
}
