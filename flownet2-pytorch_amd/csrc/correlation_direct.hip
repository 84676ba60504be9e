// correlation_direct.hip -- parameter-general correlation kernels (any pad / kernel_size /
// max_displacement / strides, f32 / f16 / f64) for gfx950.
//
// Replaces the arithmetic of the reference's channels_first + correlation_forward
// (correlation_cuda_kernel.cu:46-70, :73-147) and correlation_backward_input1/2 (:150-241,
// :243-334) for every configuration the LDS/MFMA-tiled fast path (correlation_mfma.hip) does
// not cover.  No padded-NHWC scratch copies: the zero padding is a bounds test on the NCHW
// inputs.  One lane per output element with x fastest, so for a fixed channel the 64 lanes of
// a wave read one contiguous run of in1 and one (shifted) contiguous run of in2 -- coalesced
// without any transpose.  The accumulator is fp32 for every dtype, as in the reference
// forward (:112,:124); the backward accumulates in fp32 (f32, f16) or fp64 (f64) -- the
// reference accumulates f16 in f16 there (:229), this is the more accurate superset.
//
// Where the reference reads outside its padded buffers (kernel_size > 1 with
// md - (md/s2)*s2 < (k-1)/2, or pad < the displacement reach in the backward), those reads
// are defined as 0 here (the CPU checker under tests/ defines them the same way).
#include "corr_params.h"

namespace fn2 {


template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

// ---------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(256) void corr_fwd_direct(const T *__restrict__ in1, const T *__restrict__ in2,
                                                       T *__restrict__ out, CorrP p, long total)
{
    const long HW = (long)p.H * p.W;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int bx = (int)(idx % p.oW);
        long r = idx / p.oW;
        const int by = (int)(r % p.oH);
        r /= p.oH;
        const int tc = (int)(r % p.nOut);
        const int n = (int)(r / p.nOut);
        const int tj = tc / p.D - p.dr, ti = tc % p.D - p.dr;
        // padded coordinates (:90-91,:109-110) shifted back to image coordinates
        const int y1 = by * p.s1 + p.md - p.pad, x1 = bx * p.s1 + p.md - p.pad;
        const int y2 = y1 + tj * p.s2, x2 = x1 + ti * p.s2;
        const T *a = in1 + (long)n * p.C * HW;
        const T *b = in2 + (long)n * p.C * HW;
        float acc = 0.0f;
        for (int j = -p.kr; j <= p.kr; ++j) {
            const int ya = y1 + j, yb = y2 + j;
            if (ya < 0 || ya >= p.H || yb < 0 || yb >= p.H) continue; // zero padding
            for (int i = -p.kr; i <= p.kr; ++i) {
                const int xa = x1 + i, xb = x2 + i;
                if (xa < 0 || xa >= p.W || xb < 0 || xb >= p.W) continue;
                const T *pa = a + (long)ya * p.W + xa;
                const T *pb = b + (long)yb * p.W + xb;
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f; // 4 independent chains for ILP
                int c = 0;
                for (; c + 4 <= p.C; c += 4) {
                    const T p0 = pa[(long)(c + 0) * HW] * pb[(long)(c + 0) * HW]; // product in T (:124)
                    const T p1 = pa[(long)(c + 1) * HW] * pb[(long)(c + 1) * HW];
                    const T p2 = pa[(long)(c + 2) * HW] * pb[(long)(c + 2) * HW];
                    const T p3 = pa[(long)(c + 3) * HW] * pb[(long)(c + 3) * HW];
                    s0 += (float)p0; s1 += (float)p1; s2 += (float)p2; s3 += (float)p3;
                }
                for (; c < p.C; ++c) s0 += (float)(T)(pa[(long)c * HW] * pb[(long)c * HW]);
                acc += (s0 + s1) + (s2 + s3);
            }
        }
        const int nelems = p.k * p.k * p.C;
        float res = acc / nelems; // (:143)
        if (p.slope != 1.0f) res = res > 0.0f ? res : (float)(T)res * p.slope;   // fused LeakyReLU (FlowNetC.py:87) on the stored value
        out[(long)n * p.out_bs + (idx - (long)n * p.nOut * p.oH * p.oW)] = (T)res;
    }
}

// ---------------------------------------------------------------- backward
// One lane per (n, c, y, x) input element; both gradients in one pass over grad_out.
template <typename T>
__global__ __launch_bounds__(256) void corr_bwd_direct(const T *__restrict__ in1, const T *__restrict__ in2,
                                                       const T *__restrict__ gout, T *__restrict__ g1,
                                                       T *__restrict__ g2, CorrP p, long total)
{
    typedef typename Acc<T>::type A;
    const long HW = (long)p.H * p.W, oHW = (long)p.oH * p.oW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int bx = (int)(idx % p.W);
        long r = idx / p.W;
        const int by = (int)(r % p.H);
        r /= p.H;
        const int c = (int)(r % p.C);
        const int n = (int)(r / p.C);
        const int y = by * p.s1 + p.pad, x = bx * p.s1 + p.pad; // padded coords (:161-162)
        const T *go = gout + (long)n * p.nOut * oHW;
        const T *a = in1 + ((long)n * p.C + c) * HW;
        const T *b = in2 + ((long)n * p.C + c) * HW;

        // ---- gradInput1 (:171-192): one window for all tc
        A sum1 = 0;
        {
            int xmin = (x - p.kr - p.md) / p.s1, ymin = (y - p.kr - p.md) / p.s1; // C truncating division
            int xmax = (x + p.kr - p.md) / p.s1, ymax = (y + p.kr - p.md) / p.s1;
            const bool skip = (xmax < 0 || ymax < 0 || xmin >= p.oW || ymin >= p.oH) || (xmin > xmax || ymin > ymax);
            if (!skip) {
                xmin = max(0, xmin); xmax = min(p.oW - 1, xmax);
                ymin = max(0, ymin); ymax = min(p.oH - 1, ymax);
                for (int tc = 0; tc < p.nOut; ++tc) {
                    const int i2 = (tc % p.D - p.dr) * p.s2, j2 = (tc / p.D - p.dr) * p.s2;
                    const int yy = y + j2 - p.pad, xx = x + i2 - p.pad; // image coords of rInput2[y+j2, x+i2]
                    if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) continue;
                    const A v2 = (A)b[(long)yy * p.W + xx];
                    A w = 0;
                    for (int j = ymin; j <= ymax; ++j)
                        for (int i = xmin; i <= xmax; ++i) w += (A)go[(long)tc * oHW + (long)j * p.oW + i];
                    sum1 += w * v2;
                }
            }
        }
        // ---- gradInput2 (:286-321): the window moves with tc
        A sum2 = 0;
        for (int tc = 0; tc < p.nOut; ++tc) {
            const int i2 = (tc % p.D - p.dr) * p.s2, j2 = (tc / p.D - p.dr) * p.s2;
            int xmin = (x - p.kr - p.md - i2) / p.s1, ymin = (y - p.kr - p.md - j2) / p.s1;
            int xmax = (x + p.kr - p.md - i2) / p.s1, ymax = (y + p.kr - p.md - j2) / p.s1;
            if (xmax < 0 || ymax < 0 || xmin >= p.oW || ymin >= p.oH) continue;
            if (xmin > xmax || ymin > ymax) continue;
            xmin = max(0, xmin); xmax = min(p.oW - 1, xmax);
            ymin = max(0, ymin); ymax = min(p.oH - 1, ymax);
            const int yy = y - j2 - p.pad, xx = x - i2 - p.pad; // image coords of rInput1[y-j2, x-i2]
            if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) continue;
            const A v1 = (A)a[(long)yy * p.W + xx];
            A w = 0;
            for (int j = ymin; j <= ymax; ++j)
                for (int i = xmin; i <= xmax; ++i) w += (A)go[(long)tc * oHW + (long)j * p.oW + i];
            sum2 += w * v1;
        }
        const A nelems = (A)(p.k * p.k * p.C);
        g1[idx] = (T)(sum1 / nelems);
        g2[idx] = (T)(sum2 / nelems);
    }
}

static inline unsigned stream_grid(long nthreads, long cap_blocks)
{
    long blocks = (nthreads + 255) / 256;
    if (blocks > cap_blocks) blocks = cap_blocks;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

int corr_make_params(CorrP &p, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    if (B < 0 || C < 1 || H < 1 || W < 1 || pad < 0 || k < 1 || md < 0 || s1 < 1 || s2 < 1) return FN2_EINVAL;
    p.B = B; p.C = C; p.H = H; p.W = W;
    p.pad = pad; p.k = k; p.md = md; p.s1 = s1; p.s2 = s2;
    p.kr = (k - 1) / 2;
    p.dr = md / s2;
    p.D = 2 * p.dr + 1;
    int rc = fn2_correlation_output_shape(H, W, pad, k, md, s1, s2, &p.nOut, &p.oH, &p.oW);
    p.out_bs = (long)p.nOut * p.oH * p.oW;
    p.slope = 1.0f;
    return rc;
}

template <typename T>
static int fwd_direct_launch(const void *in1, const void *in2, void *out, const CorrP &p, hipStream_t s)
{
    const long total = (long)p.B * p.nOut * p.oH * p.oW;
    if (total == 0) return FN2_OK;
    hipLaunchKernelGGL(corr_fwd_direct<T>, dim3(stream_grid(total, 256L * 64)), dim3(256), 0, s,
                       static_cast<const T *>(in1), static_cast<const T *>(in2), static_cast<T *>(out), p, total);
    return launch_status();
}

template <typename T>
static int bwd_direct_launch(const void *in1, const void *in2, const void *gout, void *g1, void *g2, const CorrP &p,
                             hipStream_t s)
{
    const long total = (long)p.B * p.C * p.H * p.W;
    if (total == 0) return FN2_OK;
    hipLaunchKernelGGL(corr_bwd_direct<T>, dim3(stream_grid(total, 256L * 64)), dim3(256), 0, s,
                       static_cast<const T *>(in1), static_cast<const T *>(in2), static_cast<const T *>(gout),
                       static_cast<T *>(g1), static_cast<T *>(g2), p, total);
    return launch_status();
}

int corr_forward_direct(const void *in1, const void *in2, void *out, int dtype, const CorrP &p, hipStream_t s)
{
    switch (dtype) {
    case FN2_F32: return fwd_direct_launch<float>(in1, in2, out, p, s);
    case FN2_F16: return fwd_direct_launch<half_t>(in1, in2, out, p, s);
    case FN2_F64: return fwd_direct_launch<double>(in1, in2, out, p, s);
    default: return FN2_EDTYPE;
    }
}

int corr_backward_direct(const void *in1, const void *in2, const void *gout, void *g1, void *g2, int dtype,
                         const CorrP &p, hipStream_t s)
{
    switch (dtype) {
    case FN2_F32: return bwd_direct_launch<float>(in1, in2, gout, g1, g2, p, s);
    case FN2_F16: return bwd_direct_launch<half_t>(in1, in2, gout, g1, g2, p, s);
    case FN2_F64: return bwd_direct_launch<double>(in1, in2, gout, g1, g2, p, s);
    default: return FN2_EDTYPE;
    }
}

} // namespace fn2
