/*
 * flownet2_hip.h -- C ABI of libflownet2_hip.so: FlowNet2's three custom layers
 * (Correlation, Resample2d, ChannelNorm) as hand-written gfx950 (MI355X) HIP kernels.
 *
 * This is the drop-in boundary.  The reference reaches its kernels through three pybind
 * modules -- correlation_cuda, resample2d_cuda, channelnorm_cuda -- whose forward/backward
 * take at::Tensor& (reference correlation_cuda.cc:169-172, resample2d_cuda.cc:28-31,
 * channelnorm_cuda.cc:27-30) and call host launchers taking raw pointers, sizes, strides
 * and a stream (correlation_cuda_kernel.cuh:7-91, resample2d_kernel.cuh:5-18,
 * channelnorm_kernel.cuh:5-16).  The entry points below replace those launchers: plain
 * pointers and integers, no torch types.  The same-named pybind modules in
 * flownet2-pytorch_amd/csrc/binding/ sit on top (shape math, resize_, device guard,
 * error translation), see INTEGRATION.md.
 *
 * Conventions
 *  - all tensors are NCHW device memory of the element type `dtype` names;
 *  - `stream` is a hipStream_t (passed as void* to keep this header HIP-free); work is
 *    enqueued on it and never synchronised here (reference: at::cuda::getCurrentCUDAStream(),
 *    correlation_cuda.cc:76, resample2d_kernel.cu:221, channelnorm_kernel.cu:113);
 *  - the caller has made the right device current;
 *  - return value: FN2_OK (0), a negative FN2_E* code for a rejected call (nothing was
 *    launched), or a positive hipError_t from the launch (hipGetLastError()).  The
 *    reference's launchers return 0 on cudaGetLastError()!=success and the binding raises
 *    AT_ERROR("CUDA call failed") (correlation_cuda_kernel.cu:417-424, correlation_cuda.cc:81-83);
 *    the bindings here raise RuntimeError with fn2_strerror(code);
 *  - re-entrant, no global mutable state.
 */
#ifndef FLOWNET2_HIP_H
#define FLOWNET2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports, literally */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* 2 (round 5): + fn2_multiscale_loss, fn2_warp_diff_norm_cat_backward, fn2_warp_diff_norm,
 * fn2_warp_diff_norm_backward.  Additive: every version-1 entry point keeps its
 * signature and meaning; a caller built against version 1 runs unchanged on a version-2 library. */
/* 3 (round 6): + fn2_multiscale_loss_fused, fn2_multiscale_scale_grads; fn2_multiscale_workspace_bytes grew by the ticket counter's
 * 64-byte line (callers size their scratch with it, as before).  Additive as well. */
#define FN2_ABI_VERSION 3

/* element types (reference dispatch: AT_DISPATCH_FLOATING_TYPES_AND_HALF for correlation and
 * channelnorm -- correlation_cuda_kernel.cu:386-415, channelnorm_kernel.cu:111,152; float only
 * for resample2d -- resample2d_kernel.cu:221,269,298) */
enum { FN2_F32 = 0, FN2_F16 = 1, FN2_F64 = 2 };

enum {
    FN2_OK = 0,
    FN2_EINVAL = -1,      /* bad shape / parameter */
    FN2_EDTYPE = -2,      /* dtype not supported by this op */
    FN2_EALIGN = -3,      /* pointer not aligned to its element size */
    FN2_EUNSUPPORTED = -4 /* parameter combination the reference itself leaves undefined */
};

/* correlation algorithm selector for fn2_correlation_forward_ex */
enum {
    FN2_CORR_AUTO = 0,    /* fastest kernel whose preconditions hold */
    FN2_CORR_DIRECT = 1,  /* one-thread-per-output kernel, any parameters / dtype */
    FN2_CORR_MFMA_F32 = 2,   /* LDS-tiled v_mfma_f32_16x16x4_f32 kernels (f32, k=1, s1=1, s2=2, pad==md): bitwise an fmaf chain */
    FN2_CORR_MFMA_BF16X3 = 3, /* fp32 operands split exactly into 3 bf16 terms, 6 partial products on
                                v_mfma_f32_16x16x32_bf16, fp32 accumulate -- fp32-class accuracy (same error vs fp64 as
                                FN2_CORR_MFMA_F32).  Forward additionally needs W <= 64, C % 32 == 0, even md/2;
                                backward needs md/2 == 10, C % 64 == 0, W % 4 == 0, 16 B aligned tensors */
    FN2_CORR_MFMA_F16X2 = 4  /* forward AND backward (csrc/correlation_f16x2*.hip, csrc/f16x2_split.h): every fp32 operand x of a
                                task is written as x * 2^k = h + l with two f16 terms (round to nearest), 3 partial products on
                                v_mfma_f32_16x16x32_f16, fp32 accumulation; the power-of-two scale 2^k -- one per operand and
                                task, from the mean binary exponent of a 128..256-element sample of the operand -- is exact and
                                is removed exactly together with the 1/C.
                                Error bound per operand, relative to the operand's typical (geometric-mean) magnitude m:
                                max(2^-22 |x| / m, 2^-27), at ANY input magnitude (checked from 2^-27 to 2^13 and on a real
                                training gradient, tests/test_gpu_parity.py magnitude sweeps): fp32-class sums, measured
                                0.65x the fp32 MFMA kernel's error against fp64.  m is what the scale places at 2^-1: the
                                geometric mean of the sample's non-zero values (mean binary exponent).  Operands above
                                2^17 m (131072 m: the f16 maximum, 2^16, over 2^-1) do not fit the scaled f16; the outputs
                                they touch are recomputed by a plain fp32 fma chain (slow, exact semantics incl. inf / nan).  Operands far BELOW m (< 2^-27 m) lose relative accuracy:
                                block scaling is not scale-invariant within a task, unlike the reference's fp32 products.
                                BACKWARD (since round 4): the in1 / in2 operand carries one scale PER CHANNEL (rows of the
                                A matrix: removed exactly per gradient channel), so "m" above is the channel's own typical
                                magnitude there -- channels of very different scale (no BatchNorm in FlowNetC) each keep
                                fp32-class gradients (tests: channel magnitudes log-uniform over 10^+-3); gradOutput
                                keeps one scale per task.
                                Needs f32, k = 1, s1 = 1, s2 = 2, pad == md == 20, even H, W % 8 == 0 (any width: maps
                                wider than 64 pixels -- Sintel-size inputs -- run column-window variants of the same kernels,
                                csrc/correlation_f16x2_wide.hip / _bwd_wide.hip, ~1.5-1.7x the time per pixel),
                                C % 64 == 0, 16 B aligned tensors (forward also out_batch_stride % 4 == 0); the forward
                                launcher additionally declines H > 512 and B x tasks-per-item >= 65536.
                                Half tensors (dtype FN2_F16) with this selector or AUTO: the same tilings with ONE f16 product
                                per block (the operands are f16 as they are; exact fp32 products, fp32 sums, result rounded to
                                half): forward C % 128 == 0, any width (csrc/correlation_f16_fwd.hip); backward C % 64 == 0,
                                W <= 64 (csrc/correlation_f16_bwd.hip; wider maps: the general kernel).
                                What FN2_CORR_AUTO picks, forward and backward, for FlowNetC's cost volume; shapes the
                                launcher declines go on to FN2_CORR_MFMA_F32 / FN2_CORR_DIRECT under AUTO. */
};
/* any other algo value: FN2_EINVAL */

const char *fn2_strerror(int code);
int fn2_abi_version(void);

/* Shape math of correlation_forward_cuda (correlation_cuda.cc:19-34):
 * nOut = ((md/s2)*2+1)^2, oH = ceil((H + 2*pad - 2*((k-1)/2 + md)) / s1), likewise oW. */
int fn2_correlation_output_shape(int H, int W, int pad_size, int kernel_size, int max_displacement,
                                 int stride1, int stride2, int *nOut, int *oH, int *oW);

/* Replaces correlation_forward_cuda_kernel (correlation_cuda_kernel.cuh:7-41; kernels
 * correlation_cuda_kernel.cu:46-70 channels_first, :73-147 correlation_forward).
 *   in1, in2 : B x C x H x W contiguous       out : B x nOut x oH x oW contiguous, fully written
 * No rInput1/rInput2 scratch is needed (the reference's padded-NHWC copies are not
 * materialised).  corr_type_multiply is accepted by the reference and unused; it is not part
 * of this ABI.  Accumulation is fp32 for every dtype, as in the reference (:112,:124). */
int fn2_correlation_forward(const void *in1, const void *in2, void *out, int dtype,
                            int B, int C, int H, int W,
                            int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                            void *stream);
int fn2_correlation_forward_ex(const void *in1, const void *in2, void *out, int dtype,
                               int B, int C, int H, int W,
                               int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                               int algo, void *stream);

/* "Next" row N1: the same forward with the two passes that follow it in FlowNetC fused into the epilogue:
 * LeakyReLU(negative_slope) on the cost volume (FlowNetC.py:87; 1.0f = no activation) and the store into a channel slice
 * of a larger NCHW buffer, i.e. the torch.cat((conv_redir, corr), 1) input of conv3_1 (FlowNetC.py:92): `out` points at the
 * first correlation channel of batch item 0 and out_batch_stride (elements, >= nOut*oH*oW) is that buffer's batch stride.
 * fn2_correlation_forward_ex == this with out_batch_stride = nOut*oH*oW, negative_slope = 1. */
int fn2_correlation_forward_fused(const void *in1, const void *in2, void *out, int64_t out_batch_stride,
                                  float negative_slope, int dtype, int B, int C, int H, int W,
                                  int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                  int algo, void *stream);

/* Replaces correlation_backward_cuda_kernel (correlation_cuda_kernel.cuh:44-91; kernels
 * correlation_cuda_kernel.cu:150-241 backward_input1, :243-334 backward_input2, host loops
 * :522-554).  grad_out : B x nOut x oH x oW contiguous; grad_in1, grad_in2 : B x C x H x W,
 * fully written (no pre-zeroing needed, unlike the reference :113-114 of correlation_cuda.cc).
 * Like the reference this is only meaningful for stride1 == 1; other values
 * return FN2_EUNSUPPORTED. */
int fn2_correlation_backward(const void *in1, const void *in2, const void *grad_out,
                             void *grad_in1, void *grad_in2, int dtype,
                             int B, int C, int H, int W,
                             int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                             void *stream);
int fn2_correlation_backward_ex(const void *in1, const void *in2, const void *grad_out,
                                void *grad_in1, void *grad_in2, int dtype,
                                int B, int C, int H, int W,
                                int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                int algo, void *stream);

/* "Next" row N1, training half: the backward pass of the fused forward above -- of the correlation branch of
 * cat((conv_redir, LeakyReLU_s(Correlation(in1, in2))), 1) (FlowNetC.py:86-87, :92).  autograd runs three passes around the
 * reference's layer (slice of the concat gradient made contiguous, leaky_relu_backward, correlation backward); this entry
 * point reads the gradient where it lies and the activation's derivative off the SIGN of the stored forward output:
 *   out_act  : what fn2_correlation_forward_fused wrote -- pointer at the first correlation channel of batch item 0 inside
 *              the concat buffer, out_batch_stride elements between items (>= nOut*oH*oW)
 *   grad_cat : gradient wrt the concat buffer, pointer at the same channel, grad_batch_stride elements between items
 *   negative_slope > 0 (LeakyReLU; 1 = no activation): d/dx = 1 where out_act > 0, negative_slope elsewhere -- no mask
 *              tensor, no copy of the pre-activation values
 *   workspace: fn2_correlation_backward_fused_workspace_bytes(...) bytes of device scratch (the masked, contiguous
 *              gradient the backward kernels read: ONE streaming pass instead of autograd's two; overwritten)
 *   grad_in1, grad_in2 : B x C x H x W, fully written; bit-identical to the unfused composition on the same kernels. */
size_t fn2_correlation_backward_fused_workspace_bytes(int dtype, int B, int H, int W, int pad_size, int kernel_size,
                                                      int max_displacement, int stride1, int stride2);
int fn2_correlation_backward_fused(const void *in1, const void *in2, const void *out_act, int64_t out_batch_stride,
                                   const void *grad_cat, int64_t grad_batch_stride, float negative_slope,
                                   void *workspace, size_t workspace_bytes, void *grad_in1, void *grad_in2,
                                   int dtype, int B, int C, int H, int W,
                                   int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                   int algo, void *stream);

/* Replaces resample2d_kernel_forward (resample2d_kernel.cuh:5-10; kernel
 * resample2d_kernel.cu:15-72).  float32 only, like the reference.
 *   img  : B x C x Hi x Wi with element strides img_strides[4] (NULL = contiguous); the
 *          reference kernel honours input1's strides too (DIM3_INDEX), its Python wrapper
 *          just never passes a strided tensor (resample2d.py:9,48)
 *   flow : B x 2 x H x W contiguous (channel 0 = dx, 1 = dy)
 *   out  : B x C x H x W contiguous, fully written
 * kernel_size >= 1.  For kernel_size > 1 the reference sums the four corners over a kernel_size x kernel_size window of
 * offsets (:54-61) without a bounds test: near the lower / right border its shifted addresses run into the next row, the
 * next channel or past the tensor (the backward pass accumulates there).  Here the shifted indices are clamped to the image
 * instead: results are the reference's for every sample whose window stays inside, i.e. yT + ky <= Hi - 1 and
 * xR + kx <= Wi - 1 for all window offsets; at the border they differ from the reference's wrapped reads even where those
 * still fall inside its allocation.  FlowNet2 only uses kernel_size 1 (models.py:48-51), where there is no difference.
 * FN2_EINVAL for kernel_size < 1. */
int fn2_resample2d_forward(const float *img, const int64_t *img_strides, const float *flow, float *out,
                           int B, int C, int Hi, int Wi, int H, int W,
                           int kernel_size, int bilinear, void *stream);

/* Replaces resample2d_kernel_backward (resample2d_kernel.cuh:12-18; kernels
 * resample2d_kernel.cu:75-125 backward_input1, :127-198 backward_input2).
 *   grad_out : B x C x H x W contiguous
 *   grad_img : B x C x Hi x Wi contiguous, ACCUMULATED INTO with fp32 atomics -- the caller
 *              zero-fills it first, exactly as the reference's wrapper does (resample2d.py:31).  The order of
 *              the fp32 additions into a cell is unspecified, as with the reference's atomicAdd: results
 *              repeat to rounding level, not bit for bit (grad_flow is bit-exact and repeatable)
 *   grad_flow: B x 2 x H x W contiguous, fully written */
int fn2_resample2d_backward(const float *img, const int64_t *img_strides, const float *flow,
                            const float *grad_out, float *grad_img, float *grad_flow,
                            int B, int C, int Hi, int Wi, int H, int W,
                            int kernel_size, int bilinear, void *stream);

/* "Next" row N2: the warp -> difference -> channel norm -> concat sequence of FlowNet2
 * (models.py:133-138, repeated at :145-150, :157-161, :170-174) as ONE pass:
 *   pair : B x 2C x H x W contiguous (channels [0,C) = first image, [C,2C) = second image)
 *   flow : B x 2 x H x W contiguous
 *   out  : B x (3C+3) x H x W contiguous, fully written
 *        = cat(pair, Resample2d(pair[:, C:], flow), flow / div_flow, ChannelNorm(pair[:, :C] - warped)), dim 1
 * with the arithmetic of the unfused reference kernels (resample2d_kernel.cu:15-72, channelnorm_kernel.cu:18-60);
 * flow / div_flow is flow * (1.0f / div_flow), which is what PyTorch computes for a GPU tensor divided by a scalar.
 * float32; kernel_size 1. */
int fn2_warp_diff_norm_cat(const float *pair, const float *flow, float *out, float div_flow,
                           int B, int C, int H, int W, int bilinear, void *stream);

/* Row N2, training half: the backward pass of fn2_warp_diff_norm_cat -- what autograd computes through
 * resample -> difference -> ChannelNorm -> cat (models.py:133-138; resample2d_kernel.cu:75-198, channelnorm_kernel.cu:63-96) -- in
 * ONE kernel: g_diff = g_norm * diff / (norm + 1e-9) with diff = pair[:, :C] - warped (warped and norm are read back from the
 * forward's output), g_warped = grad_cat[:, 2C:3C] - g_diff feeds the scatter / gather of the Resample2d backward directly.
 *   out_cat   : the forward's output, B x (3C+3) x H x W contiguous
 *   grad_cat  : gradient of that output, contiguous
 *   grad_pair : NULL (the pair is the network's input: no scatter, no atomics), or B x 2C x H x W contiguous, fully written:
 *               [:, :C] = grad_cat[:, :C] + g_diff, [:, C:] = grad_cat[:, C:2C] + scatter of g_warped (fp32 atomics, order unspecified)
 *   grad_flow : B x 2 x H x W contiguous, fully written = Resample2d's flow gradient + grad_cat[:, 3C:3C+2] * (1 / div_flow);
 *               bit-identical to the composition of the unfused entry points
 * `bilinear` is accepted and ignored, as by both reference backward kernels.  float32, kernel_size 1.  C = 3 on tileable maps
 * (H >= 16, W >= 32, W % 4 == 0, 16-byte aligned pair) takes the LDS-window kernel, everything else one lane per pixel. */
int fn2_warp_diff_norm_cat_backward(const float *pair, const float *flow, const float *out_cat, const float *grad_cat,
                                    float *grad_pair, float *grad_flow, float div_flow, int B, int C, int H, int W,
                                    int bilinear, void *stream);

/* Row N2 without the concat -- the other two warp sites of FlowNet2 (models.py:157-161, :170-174):
 *   out_norm = ChannelNorm(pair[:, :C] - Resample2d(pair[:, C:], flow))      B x 1 x H x W, fully written
 * by the kernel of fn2_warp_diff_norm_cat storing only the norm plane (bit-identical to that plane), and its backward with respect
 * to the flow: g_warped = -grad_norm * diff / (norm + 1e-9) (channelnorm_kernel.cu:93) fed to the gather of the Resample2d backward
 * (resample2d_kernel.cu:127-198); the warped image is not stored by the forward and is recomputed from the same corners with the
 * forward's arithmetic (`bilinear` must be the forward's).  grad_flow: B x 2 x H x W, fully written, bit-identical to autograd
 * through the unfused entry points.  The pair gets no gradient here (it is the network's input in FlowNet2): compose
 * fn2_channelnorm_backward and fn2_resample2d_backward when it needs one.  float32, kernel_size 1.  (ABI v2) */
int fn2_warp_diff_norm(const float *pair, const float *flow, float *out_norm, int B, int C, int H, int W, int bilinear, void *stream);
int fn2_warp_diff_norm_backward(const float *pair, const float *flow, const float *norm, const float *grad_norm, float *grad_flow,
                                int B, int C, int H, int W, int bilinear, void *stream);

/* "Next" row N3: the training loss of FlowNet2 -- MultiScale with the L1 norm (losses.py:52-86) -- and
 * the EPE metric (losses.py:11-12) in one pass over the target flow instead of five AvgPool2d passes and ~35 launches.
 *   outputs[i] : B x 2 x (H/k_i) x (W/k_i) contiguous device tensors, k_i = start_scale << i, i < num_scales (host array
 *                of device pointers); start_scale a power of two <= 16, num_scales <= 5 with k_max / start_scale <= 16
 *   target     : B x 2 x H x W contiguous
 *   sums       : 2*num_scales floats on the device, fully written:
 *                sums[i]              = sum |out_i - AvgPool_ki(div_flow * target)|        (L1_i  = sums[i] / (B*2*H_i*W_i))
 *                sums[num_scales + i] = sum over pixels of the channel 2-norm of that     (EPE_i = ... / (B*H_i*W_i))
 *   grads      : NULL, or num_scales device tensors shaped like outputs[i], fully written with
 *                grad_scale * weights[i] / (B*2*H_i*W_i) * sign(out_i - t_i) = d(sum_i w_i L1_i)/d out_i * grad_scale
 *   weights    : host array of num_scales loss weights (only used for grads)
 *   workspace  : device scratch of fn2_multiscale_workspace_bytes(...) bytes (per-workgroup partial sums and a ticket counter:
 *                the last workgroup to finish adds the partial sums in a fixed order -- the result is deterministic)
 * float32. */
size_t fn2_multiscale_workspace_bytes(int B, int H, int W, int start_scale, int num_scales);
int fn2_multiscale_l1_epe(const float *const *outputs, const float *target, float *sums, float *const *grads,
                          const float *weights, float grad_scale, int B, int H, int W, int start_scale, int num_scales,
                          float div_flow, void *workspace, size_t workspace_bytes, void *stream);
/* The same with the norm of losses.py:62-67 selectable: norm = 1 is fn2_multiscale_l1_epe; norm = 2 is MultiScale(norm='L2'),
 * whose per-scale loss L2() (losses.py:21-26) is the expression of EPE (:12) -- `sums` is unchanged (the loss is then
 * sum_i w_i * sums[num_scales + i] / (B*H_i*W_i)) and grads[i] = grad_scale * weights[i] / (B*H_i*W_i) *
 * (out_i - t_i) / ||out_i - t_i||_2 per pixel, 0 where that norm is 0 (torch.norm's backward). */
int fn2_multiscale_loss(const float *const *outputs, const float *target, float *sums, float *const *grads,
                        const float *weights, float grad_scale, int norm, int B, int H, int W, int start_scale, int num_scales,
                        float div_flow, void *workspace, size_t workspace_bytes, void *stream);

/* Row N3 as the autograd node uses it (ABI v3): the same pass, with the weighted means themselves written by the kernel --
 *   loss_epe[0] = sum_i weights[i] * L_i   (norm 1: L_i = sums[i] / (B*2*H_i*W_i); norm 2: L_i = sums[n+i] / (B*H_i*W_i), losses.py:21-26)
 *   loss_epe[1] = sum_i weights[i] * sums[n+i] / (B*H_i*W_i)                                               (losses.py:77)
 * -- by the last workgroup to finish (a ticket counter at the end of `workspace`; the summation order is fixed, the result
 * deterministic), so that loss and metric cost ONE launch.  workspace_primed != 0: the caller guarantees that the workspace was
 * zero-filled once and has since only been used by this entry point on one stream at a time (the kernel leaves the counter at
 * zero); 0: the counter is cleared by a 4-byte memset node in front of the kernel (any scratch memory will do).  `weights` is
 * required. */
int fn2_multiscale_loss_fused(const float *const *outputs, const float *target, float *sums, float *loss_epe, float *const *grads,
                              const float *weights, float grad_scale, int norm, int B, int H, int W, int start_scale, int num_scales,
                              float div_flow, void *workspace, size_t workspace_bytes, int workspace_primed, void *stream);
/* grads[i][j] = unit_grads[i][j] * *scale for i < num_scales, j < numel[i], in one launch: the backward of the node above for an
 * incoming gradient that lives on the device (unit_grads = what fn2_multiscale_loss_fused wrote with grad_scale 1).  `scale`: device
 * pointer to one float; unit_grads / grads / numel: host arrays.  Out of place, so that a retained graph sees its saved gradients
 * unchanged. (ABI v3) */
int fn2_multiscale_scale_grads(const float *const *unit_grads, float *const *grads, const int64_t *numel, int num_scales,
                               const float *scale, void *stream);

/* Replaces channelnorm_kernel_forward (channelnorm_kernel.cuh:5-8; kernel
 * channelnorm_kernel.cu:18-60).  in : B x C x H x W contiguous, out : B x 1 x H x W contiguous.
 * norm_deg is accepted and ignored by the reference kernels (always L2); not part of this ABI. */
int fn2_channelnorm_forward(const void *in, void *out, int dtype, int B, int C, int H, int W, void *stream);

/* Replaces channelnorm_kernel_backward (channelnorm_kernel.cuh:10-16; kernel
 * channelnorm_kernel.cu:63-96).  grad_out is B x 1 x H x W with element strides
 * gout_strides[4] (NULL = contiguous): the reference ignores gradOutput's strides (:92) and
 * therefore mis-reads the non-contiguous slice autograd hands it (last bullet);
 * this entry point honours them. */
int fn2_channelnorm_backward(const void *in, const void *out, const void *grad_out, const int64_t *gout_strides,
                             void *grad_in, int dtype, int B, int C, int H, int W, void *stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FLOWNET2_HIP_H */
