// capi.hip -- C-ABI entry points of libflownet2_hip.so that are not tied to one kernel file:
// error strings, shape math and the correlation dispatcher (include/flownet2_hip.h).
#include <math.h>

#include "corr_params.h"
#include "fn2_debug.h"


extern "C" const char *fn2_strerror(int code)
{
    switch (code) {
    case FN2_OK: return "ok";
    case FN2_EINVAL: return "flownet2_hip: invalid shape or parameter";
    case FN2_EDTYPE: return "flownet2_hip: dtype not supported by this op";
    case FN2_EALIGN: return "flownet2_hip: pointer not aligned to its element size";
    case FN2_EUNSUPPORTED: return "flownet2_hip: parameter combination the reference leaves undefined";
    default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "flownet2_hip: unknown error";
}

extern "C" int fn2_abi_version(void) { return FN2_ABI_VERSION; }

// correlation_cuda.cc:19-34
extern "C" int fn2_correlation_output_shape(int H, int W, int pad_size, int kernel_size, int max_displacement,
                                            int stride1, int stride2, int *nOut, int *oH, int *oW)
{
    if (H < 1 || W < 1 || pad_size < 0 || kernel_size < 1 || max_displacement < 0 || stride1 < 1 || stride2 < 1)
        return FN2_EINVAL;
    const int kernel_radius = (kernel_size - 1) / 2;
    const int border_radius = kernel_radius + max_displacement;
    const int pH = H + 2 * pad_size, pW = W + 2 * pad_size;
    const int d = (max_displacement / stride2) * 2 + 1;
    const int oh = (int)ceilf((float)(pH - 2 * border_radius) / (float)stride1);
    const int ow = (int)ceilf((float)(pW - 2 * border_radius) / (float)stride1);
    if (oh < 1 || ow < 1) return FN2_EINVAL;
    if (nOut) *nOut = d * d;
    if (oH) *oH = oh;
    if (oW) *oW = ow;
    return FN2_OK;
}

static int corr_forward_impl(const void *in1, const void *in2, void *out, int64_t out_batch_stride,
                             float negative_slope, int dtype, int B, int C, int H, int W,
                             int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                             int algo, bool debug_variant, void *stream)
{
    using namespace fn2;
    const size_t es = dtype_size(dtype);
    if (!es) return FN2_EDTYPE;
    CorrP p;
    int rc = corr_make_params(p, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2);
    if (rc != FN2_OK) return rc;
    if (out_batch_stride < p.out_bs || !(negative_slope == negative_slope)) return FN2_EINVAL;
    p.out_bs = out_batch_stride;
    p.slope = negative_slope;
    if (B == 0) return FN2_OK;
    if (!in1 || !in2 || !out) return FN2_EINVAL;
    if (!aligned(in1, es) || !aligned(in2, es) || !aligned(out, es)) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // profiling instantiations (wrong or partial outputs by design) are only reachable through fn2_debug_* (fn2_debug.h)
    if (!debug_variant && (algo < FN2_CORR_AUTO || algo > FN2_CORR_MFMA_F16X2)) return FN2_EINVAL;
    // f16x2: two-term f16 split done once per staged value, 3 MFMAs per block product (correlation_f16x2.hip)
    const bool f16x2_ok = corr_f16x2_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2) &&
                          aligned(in1, 16) && aligned(in2, 16) && aligned(out, 16) && (out_batch_stride % 4 == 0);
    if (debug_variant && algo >= 5000) {
        if (!f16x2_ok) return FN2_EUNSUPPORTED;
        return corr_forward_f16x2(static_cast<const float *>(in1), static_cast<const float *>(in2), static_cast<float *>(out),
                                  p.out_bs, p.slope, B, C, H, W, algo - 5000, s);
    }
    // half tensors on FlowNetC's configuration: the single-product f16 kernel (correlation_f16_fwd.hip); AUTO or the f16x2 selector
    if (!debug_variant && (algo == FN2_CORR_AUTO || algo == FN2_CORR_MFMA_F16X2) &&
        corr_f16_fwd_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)) {
        rc = corr_forward_f16(in1, in2, out, p.out_bs, p.slope, B, C, H, W, s);
        if (!(algo == FN2_CORR_AUTO && (rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN))) return rc;
    }
    if (algo == FN2_CORR_MFMA_F16X2 && !f16x2_ok) return FN2_EUNSUPPORTED;
    if (algo == FN2_CORR_MFMA_F16X2 || (algo == FN2_CORR_AUTO && f16x2_ok)) {
        rc = corr_forward_f16x2(static_cast<const float *>(in1), static_cast<const float *>(in2), static_cast<float *>(out),
                                p.out_bs, p.slope, B, C, H, W, 0, s);
        // automatic selection: a shape the launcher declines (task table / index limits; nothing launched) goes on to the
        // next kernel, as below
        if (!(algo == FN2_CORR_AUTO && (rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN))) return rc;
    }
    const bool mfma_ok = corr_mfma_f32_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                                                  stride2) && aligned(in1, 8) && aligned(in2, 8) && aligned(out, 8) &&
                         (out_batch_stride % 2 == 0);
    const bool wants_mfma = (algo == FN2_CORR_MFMA_F32 || algo == FN2_CORR_MFMA_BF16X3 || debug_variant);
    if (wants_mfma && !mfma_ok) return FN2_EUNSUPPORTED;
    if (wants_mfma || (algo == FN2_CORR_AUTO && mfma_ok)) {
        rc = corr_forward_mfma_f32(static_cast<const float *>(in1), static_cast<const float *>(in2),
                                   static_cast<float *>(out), p.out_bs, p.slope, B, C, H, W, max_displacement,
                                   algo, s); // 0 auto, 2 fp32 MFMA, 3 bf16x3, >= 100 profiling instantiations (debug only)
        // automatic selection: a shape the tiled kernels decline (nothing launched) goes to the general kernel
        if (!(algo == FN2_CORR_AUTO && (rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN))) return rc;
    }
    if (algo != FN2_CORR_AUTO && algo != FN2_CORR_DIRECT) return FN2_EINVAL;
    // double tensors on FlowNetC's configuration: v_mfma_f64_16x16x4_f64 (correlation_mfma_f64.hip); AUTO only -- FN2_CORR_DIRECT
    // keeps selecting the one-thread-per-output kernel
    if (algo == FN2_CORR_AUTO && corr_mfma_f64_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2) &&
        aligned(in1, 16) && aligned(in2, 16) && aligned(out, 16) && (out_batch_stride % 2 == 0)) {
        rc = corr_forward_mfma_f64(static_cast<const double *>(in1), static_cast<const double *>(in2), static_cast<double *>(out), p.out_bs,
                                   (double)p.slope, B, C, H, W, s);
        if (!(rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN)) return rc;
    }
    return corr_forward_direct(in1, in2, out, dtype, p, s);
}

extern "C" int fn2_correlation_forward_fused(const void *in1, const void *in2, void *out, int64_t out_batch_stride,
                                             float negative_slope, int dtype, int B, int C, int H, int W,
                                             int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                             int algo, void *stream)
{
    return corr_forward_impl(in1, in2, out, out_batch_stride, negative_slope, dtype, B, C, H, W, pad_size, kernel_size,
                             max_displacement, stride1, stride2, algo, false, stream);
}

extern "C" int fn2_correlation_forward_ex(const void *in1, const void *in2, void *out, int dtype,
                                          int B, int C, int H, int W,
                                          int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                          int algo, void *stream)
{
    int nOut = 0, oH = 0, oW = 0;
    const int rc = fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut,
                                                &oH, &oW);
    if (rc != FN2_OK) return rc;
    return fn2_correlation_forward_fused(in1, in2, out, (int64_t)nOut * oH * oW, 1.0f, dtype, B, C, H, W, pad_size,
                                         kernel_size, max_displacement, stride1, stride2, algo, stream);
}

extern "C" int fn2_correlation_forward(const void *in1, const void *in2, void *out, int dtype,
                                       int B, int C, int H, int W,
                                       int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                       void *stream)
{
    return fn2_correlation_forward_ex(in1, in2, out, dtype, B, C, H, W, pad_size, kernel_size, max_displacement,
                                      stride1, stride2, FN2_CORR_AUTO, stream);
}

static int corr_backward_impl(const void *in1, const void *in2, const void *grad_out,
                              void *grad_in1, void *grad_in2, int dtype,
                              int B, int C, int H, int W,
                              int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                              int algo, bool debug_variant, void *stream)
{
    using namespace fn2;
    const size_t es = dtype_size(dtype);
    if (!es) return FN2_EDTYPE;
    CorrP p;
    int rc = corr_make_params(p, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2);
    if (rc != FN2_OK) return rc;
    // the reference's backward indexes (and writes) out of bounds for stride1 != 1
    if (stride1 != 1) return FN2_EUNSUPPORTED;
    if (B == 0) return FN2_OK;
    if (!in1 || !in2 || !grad_out || !grad_in1 || !grad_in2) return FN2_EINVAL;
    if (!aligned(in1, es) || !aligned(in2, es) || !aligned(grad_out, es) || !aligned(grad_in1, es) ||
        !aligned(grad_in2, es))
        return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!debug_variant && (algo < FN2_CORR_AUTO || algo > FN2_CORR_MFMA_F16X2)) return FN2_EINVAL;
    // f16x2: one-time two-term f16 split, gathered G operand (correlation_f16x2_bwd.hip)
    const bool f16x2_ok = corr_bwd_f16x2_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2) &&
                          aligned(in1, 16) && aligned(in2, 16) && aligned(grad_out, 16) && aligned(grad_in1, 16) && aligned(grad_in2, 16);
    if (debug_variant && algo >= 6000) {
        if (!f16x2_ok) return FN2_EUNSUPPORTED;
        return corr_backward_f16x2(static_cast<const float *>(in1), static_cast<const float *>(in2), static_cast<const float *>(grad_out),
                                   static_cast<float *>(grad_in1), static_cast<float *>(grad_in2), B, C, H, W, algo - 6000, s);
    }
    // half tensors on FlowNetC's configuration: the single-product f16 kernel (correlation_f16_bwd.hip); AUTO or the f16x2 selector
    if (!debug_variant && (algo == FN2_CORR_AUTO || algo == FN2_CORR_MFMA_F16X2) &&
        corr_f16_bwd_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)) {
        rc = corr_backward_f16(in1, in2, grad_out, grad_in1, grad_in2, B, C, H, W, s);
        if (!(algo == FN2_CORR_AUTO && (rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN))) return rc;
    }
    if (!debug_variant && algo == FN2_CORR_MFMA_F16X2 && !f16x2_ok) return FN2_EUNSUPPORTED;
    if (!debug_variant && (algo == FN2_CORR_MFMA_F16X2 || (algo == FN2_CORR_AUTO && f16x2_ok))) {
        rc = corr_backward_f16x2(static_cast<const float *>(in1), static_cast<const float *>(in2), static_cast<const float *>(grad_out),
                                 static_cast<float *>(grad_in1), static_cast<float *>(grad_in2), B, C, H, W, 0, s);
        if (!(algo == FN2_CORR_AUTO && (rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN))) return rc;
    }
    const bool mfma_ok = corr_bwd_mfma_f32_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                                                      stride2) &&
                         aligned(in1, 8) && aligned(in2, 8) && aligned(grad_out, 8);
    const bool wants_mfma = (algo == FN2_CORR_MFMA_F32 || algo == FN2_CORR_MFMA_BF16X3 || debug_variant);
    if (wants_mfma && !mfma_ok) return FN2_EUNSUPPORTED;
    if (wants_mfma || (algo == FN2_CORR_AUTO && mfma_ok)) {
        // internal tune: 0 = automatic (bf16x3 where its extra preconditions hold), 6 = fp32 MFMA, 4 = bf16x3 or
        // FN2_EUNSUPPORTED, algo - 100 = profiling variants
        const int tune = algo == FN2_CORR_MFMA_F32 ? 6 : algo == FN2_CORR_MFMA_BF16X3 ? 4 : debug_variant ? algo - 100 : 0;
        rc = corr_backward_mfma_f32(static_cast<const float *>(in1), static_cast<const float *>(in2),
                                    static_cast<const float *>(grad_out), static_cast<float *>(grad_in1),
                                    static_cast<float *>(grad_in2), B, C, H, W, max_displacement, tune, s);
        if (!(algo == FN2_CORR_AUTO && (rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN))) return rc;
    }
    if (algo != FN2_CORR_AUTO && algo != FN2_CORR_DIRECT) return FN2_EINVAL;
    if (algo == FN2_CORR_AUTO && corr_mfma_f64_applicable(dtype, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2) &&
        aligned(in1, 16) && aligned(in2, 16) && aligned(grad_out, 16)) {
        rc = corr_backward_mfma_f64(static_cast<const double *>(in1), static_cast<const double *>(in2), static_cast<const double *>(grad_out),
                                    static_cast<double *>(grad_in1), static_cast<double *>(grad_in2), B, C, H, W, s);
        if (!(rc == FN2_EUNSUPPORTED || rc == FN2_EALIGN)) return rc;
    }
    return corr_backward_direct(in1, in2, grad_out, grad_in1, grad_in2, dtype, p, s);
}

extern "C" int fn2_correlation_backward_ex(const void *in1, const void *in2, const void *grad_out,
                                           void *grad_in1, void *grad_in2, int dtype,
                                           int B, int C, int H, int W,
                                           int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                           int algo, void *stream)
{
    return corr_backward_impl(in1, in2, grad_out, grad_in1, grad_in2, dtype, B, C, H, W, pad_size, kernel_size,
                              max_displacement, stride1, stride2, algo, false, stream);
}

extern "C" int fn2_correlation_backward(const void *in1, const void *in2, const void *grad_out,
                                        void *grad_in1, void *grad_in2, int dtype,
                                        int B, int C, int H, int W,
                                        int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                        void *stream)
{
    return fn2_correlation_backward_ex(in1, in2, grad_out, grad_in1, grad_in2, dtype, B, C, H, W, pad_size,
                                       kernel_size, max_displacement, stride1, stride2, FN2_CORR_AUTO, stream);
}

#ifdef FN2_DEBUG_BUILD   // libflownet2_hip_debug.so only (build.py): the product library exports none of this
// ---- profiling / ablation instantiations (fn2_debug.h): not part of the public ABI, outputs may be wrong by design
extern "C" int fn2_debug_correlation_forward(const void *in1, const void *in2, void *out, int dtype, int B, int C, int H, int W,
                                             int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                             int variant, void *stream)
{
    if (variant < 100) return FN2_EINVAL;
    int nOut = 0, oH = 0, oW = 0;
    const int rc = fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut, &oH, &oW);
    if (rc != FN2_OK) return rc;
    return corr_forward_impl(in1, in2, out, (int64_t)nOut * oH * oW, 1.0f, dtype, B, C, H, W, pad_size, kernel_size,
                             max_displacement, stride1, stride2, variant, true, stream);
}

extern "C" int fn2_debug_correlation_backward(const void *in1, const void *in2, const void *grad_out, void *grad_in1,
                                              void *grad_in2, int dtype, int B, int C, int H, int W, int pad_size,
                                              int kernel_size, int max_displacement, int stride1, int stride2, int variant,
                                              void *stream)
{
    if (variant < 100) return FN2_EINVAL;
    return corr_backward_impl(in1, in2, grad_out, grad_in1, grad_in2, dtype, B, C, H, W, pad_size, kernel_size,
                              max_displacement, stride1, stride2, variant, true, stream);
}

extern "C" void fn2_debug_set_buffer(void *device_ptr) { fn2::corr_f16x2_set_debug_buffer(device_ptr); }

// Streaming-copy probe (bench.py's `copy_ceiling_GBps`): 16 bytes per lane, grid-stride, four independent loads in flight per
// lane -- the float4 copy MI355X_MICROARCH.md quotes 6.29 TB/s for (read + write bytes).
typedef float scf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(scf4 *__restrict__ dst, const scf4 *__restrict__ src, size_t n16, int nt)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nt) {
        for (; i + 3 * stride < n16; i += 4 * stride) {
            const scf4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
            const scf4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
            __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
            __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
        }
    } else {
        for (; i + 3 * stride < n16; i += 4 * stride) {
            const scf4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
            dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// Clock probe (bench.py's `box.mfma_probe_TFLOPs`): every SIMD of the chip runs a register-only stream of independent
// v_mfma_f32_16x16x32_f16 from 4 waves -- no memory, no LDS -- so the achieved rate is (matrix-pipe rate) x (the shader clock this
// box sustains under load): the boxes of the pool differ by 20 % on the whole step, this number says whether it is the clock.
__global__ __launch_bounds__(1024) void mfma_probe_kernel(float *sink, int iters)
{
    typedef float pf4 __attribute__((ext_vector_type(4)));
    typedef _Float16 ph8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    pf4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (pf4){0.0f, 0.0f, 0.0f, 0.0f};
    ph8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}
// launches the probe; returns the number of MFMA FLOP it executes in *flop (0 on error)
extern "C" int fn2_debug_mfma_probe(void *sink, int iters, int workgroups, double *flop, void *stream)
{
    if (!sink || iters < 1 || workgroups < 1 || !flop) return FN2_EINVAL;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)workgroups), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       static_cast<float *>(sink), iters);
    *flop = (double)workgroups * 16.0 * iters * 8.0 * 16384.0;   // 16 waves x iters x 8 MFMAs x 2 * 16 * 16 * 32 FLOP
    return fn2::launch_status();
}

// Where do the workgroups of a 1-D grid run?  out[b] = HW_REG_XCC_ID of workgroup b.  The kernels' tile orders (xcd_remap) assume
// "workgroup b runs on XCD b % 8" for SPEED only (neighbouring tiles share an L2); bench.py reports the census with its box probes.
__global__ void xcc_census_kernel(int *out)
{
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 15;
}
extern "C" int fn2_debug_xcc_census(int *out, int workgroups, void *stream)
{
    if (!out || workgroups < 1) return FN2_EINVAL;
    hipLaunchKernelGGL(xcc_census_kernel, dim3((unsigned)workgroups), dim3(64), 0, static_cast<hipStream_t>(stream), out);
    return fn2::launch_status();
}

extern "C" int fn2_debug_stream_copy(void *dst, const void *src, size_t bytes, int blocks, int nontemporal, void *stream)
{
    if (!dst || !src || (bytes % 16) || !fn2::aligned(dst, 16) || !fn2::aligned(src, 16) || blocks < 1) return FN2_EINVAL;
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<scf4 *>(dst), static_cast<const scf4 *>(src), bytes / 16, nontemporal);
    return fn2::launch_status();
}
#endif
