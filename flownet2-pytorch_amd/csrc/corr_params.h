// corr_params.h -- correlation parameter block shared by the dispatcher and the kernel files.
#pragma once
#include "fn2_common.h"

namespace fn2 {

struct CorrP {
    int B, C, H, W;            // input1 / input2 shape (NCHW)
    int pad, k, md, s1, s2;    // pad_size, kernel_size, max_displacement, stride1, stride2
    int kr, dr, D, nOut, oH, oW;
    long out_bs;               // elements between batch items of the output (nOut*oH*oW unless it is a slice of a larger buffer)
    float slope;               // LeakyReLU negative slope applied to the output (1 = none)
};

int corr_make_params(CorrP &p, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_forward_direct(const void *in1, const void *in2, void *out, int dtype, const CorrP &p, hipStream_t s);
int corr_backward_direct(const void *in1, const void *in2, const void *gout, void *g1, void *g2, int dtype,
                         const CorrP &p, hipStream_t s);
bool corr_mfma_f32_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_forward_mfma_f32(const float *in1, const float *in2, float *out, long out_bs, float slope, int B, int C, int H,
                          int W, int md, int tune, hipStream_t s);

bool corr_f16x2_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_forward_f16x2(const float *in1, const float *in2, float *out, long out_bs, float slope, int B, int C, int H, int W,
                       int variant, hipStream_t s);
int corr_forward_f16x2_wide(const float *in1, const float *in2, float *out, long out_bs, float slope, int B, int C, int H, int W,
                            hipStream_t s);   // W > 64 (correlation_f16x2_wide.hip)

// half tensors (correlation_f16_fwd.hip): one f16 MFMA per block product, no split
bool corr_f16_fwd_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_forward_f16(const void *in1, const void *in2, void *out, long out_bs, float slope, int B, int C, int H, int W, hipStream_t s);

// half tensors, backward (correlation_f16_bwd.hip): one f16 MFMA per product, fp32 sums
bool corr_f16_bwd_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_backward_f16(const void *in1, const void *in2, const void *gout, void *g1, void *g2, int B, int C, int H, int W, hipStream_t s);

void corr_f16x2_set_debug_buffer(void *p);
void *corr_f16x2_get_debug_buffer();
bool corr_bwd_f16x2_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_backward_f16x2(const float *in1, const float *in2, const float *gout, float *g1, float *g2, int B, int C, int H, int W,
                        int variant, hipStream_t s);
int corr_backward_f16x2_wide(const float *in1, const float *in2, const float *gout, float *g1, float *g2, int B, int C, int H, int W,
                             hipStream_t s);   // W > 64 (correlation_f16x2_bwd_wide.hip)

bool corr_bwd_mfma_f32_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_backward_mfma_f32(const float *in1, const float *in2, const float *gout, float *g1, float *g2,
                           int B, int C, int H, int W, int md, int tune, hipStream_t s);

// double tensors on the fp64 matrix cores (correlation_mfma_f64.hip): FlowNetC's configuration, md = 20
bool corr_mfma_f64_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2);
int corr_forward_mfma_f64(const double *in1, const double *in2, double *out, long out_bs, double slope, int B, int C, int H, int W, hipStream_t s);
int corr_backward_mfma_f64(const double *in1, const double *in2, const double *gout, double *g1, double *g2, int B, int C, int H, int W, hipStream_t s);

} // namespace fn2
