// fn2_debug.h -- profiling / ablation entry points of libflownet2_hip.so.  NOT part of the public C ABI
// (include/flownet2_hip.h): the instantiations they select skip MFMAs, loads or stores, or dump timestamps into the
// output, so their results are wrong by design.  Used by scripts/ (micro-benchmarks) and a few parity tests of
// alternative tilings.
//   forward variants : 100 .. 4999  instantiations of correlation_mfma.hip (see corr_forward_mfma_f32)
//                      5000 + v     correlation_f16x2.hip with profiling switches v (1 no MFMA, 2 no global loads,
//                                   4 no stores, 8 no operand reads, 16 no split / LDS staging writes)
//   backward variants: 100 + v      instantiations of correlation_mfma_bwd.hip
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the libraries are built with -fvisibility=hidden */
int fn2_debug_correlation_forward(const void *in1, const void *in2, void *out, int dtype, int B, int C, int H, int W,
                                  int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                  int variant, void *stream);
int fn2_debug_correlation_backward(const void *in1, const void *in2, const void *grad_out, void *grad_in1, void *grad_in2,
                                   int dtype, int B, int C, int H, int W, int pad_size, int kernel_size,
                                   int max_displacement, int stride1, int stride2, int variant, void *stream);
/* resample2d with profiling switches in `flags` (bit 8: untiled kernels, bits 9-11: backward ablations, bits 12-13: tile
 * height 1 = 48, 2 = 32, 3 = 64 rows); the public entry points only look at bilinear != 0 */
int fn2_debug_resample2d_forward(const float *img, const int64_t *img_strides, const float *flow, float *out,
                                 int B, int C, int Hi, int Wi, int H, int W, int kernel_size, int bilinear, int flags, void *stream);
int fn2_debug_resample2d_backward(const float *img, const int64_t *img_strides, const float *flow, const float *grad_out,
                                  float *grad_img, float *grad_flow, int B, int C, int Hi, int Wi, int H, int W,
                                  int kernel_size, int bilinear, int flags, void *stream);
/* device buffer (>= 64 KB) that forward variant 5064 dumps its s_memtime stamps into; NULL = none */
void fn2_debug_set_buffer(void *device_ptr);
/* float4 grid-stride device-to-device copy of `bytes` (multiple of 16) with `blocks` workgroups of 256 lanes: the streaming
 * ceiling bench.py reports next to the 8 TB/s spec peak */
int fn2_debug_stream_copy(void *dst, const void *src, size_t bytes, int blocks, int nontemporal, void *stream);
/* register-only f16 MFMA stream on every SIMD (`workgroups` x 16 waves x iters x 8 MFMAs): a probe of the shader clock the box
 * sustains; *flop receives the FLOP count, `sink` is a device buffer of >= 4 KB that is never written */
int fn2_debug_mfma_probe(void *sink, int iters, int workgroups, double *flop, void *stream);
/* out[b] = HW_REG_XCC_ID of workgroup b of a 1-D grid of `workgroups` (device array of ints) */
int fn2_debug_xcc_census(int *out, int workgroups, void *stream);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
