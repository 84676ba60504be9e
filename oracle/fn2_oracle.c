/*
 * fn2_oracle.c -- CPU oracle for the FlowNet2 custom-layer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  A plain-C restatement of the arithmetic of the
 * reference's CUDA kernels (NVIDIA/flownet2-pytorch, the three networks/<op>_package .cu files).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (flownet2-pytorch_amd/) never does.
 *
 * Parity pin: the reference ships no tests or golden vectors.
 * The pin used instead is the reference's own kernel source executed on the
 * CPU: oracle/simt/ compiles the __global__ functions of the three .cu files,
 * from where they lie under /root/reference, against a CPU SIMT shim
 * (oracle/_ref/libfn2_ref.so) and the .npz files in tests/golden hold its outputs; this
 * file is checked against both (tests/test_oracle_vs_ref.py,
 * tests/test_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* shape math of correlation_forward_cuda (correlation_cuda.cc:19-34) */
int fn2o_corr_shapes(int H, int W, int pad, int k, int md, int s1, int s2,
                     int *pH, int *pW, int *nOut, int *oH, int *oW)
{
    if (k < 1 || s1 < 1 || s2 < 1 || pad < 0 || md < 0) return -1;
    const int kernel_radius = (k - 1) / 2;
    const int border_radius = kernel_radius + md;
    *pH = H + 2 * pad;
    *pW = W + 2 * pad;
    *nOut = ((md / s2) * 2 + 1) * ((md / s2) * 2 + 1);
    *oH = (int)ceilf((float)(*pH - 2 * border_radius) / (float)s1);
    *oW = (int)ceilf((float)(*pW - 2 * border_radius) / (float)s1);
    if (*oH < 1 || *oW < 1) return -1;
    return 0;
}

#define T float
#define SUF f32
#include "fn2_oracle_body.h"
#undef T
#undef SUF

#define T double
#define SUF f64
#include "fn2_oracle_body.h"
#undef T
#undef SUF

/* ------------------------------------------------------------------------- */
/* resample2d -- float only in the reference (resample2d_kernel.cu:221,269,298) */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* int(floor(xf)) / int(floor(xf)+1) as the reference forms them, made safe for
 * non-finite or huge coordinates (where the CUDA float->int conversion
 * saturates and the C one is undefined). */
static inline int f2i_sat(float v)
{
    if (!(v == v)) return 0; /* NaN -> 0 like cvt.rzi.s32.f32 */
    if (v >= 2147483520.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

/* kernel_resample2d_update_output<float> (resample2d_kernel.cu:15-72).
 * img: B x C x Hi x Wi (contiguous), flow: B x 2 x H x W, out: B x C x H x W
 * (shape rule of Resample2dFunction.forward, resample2d.py:16-18).
 * kernel_size > 1 reads out of bounds in the reference; here those reads are
 * clamped to the image (documented divergence; only kernel_size=1 is used). */
int fn2o_resample_fwd_f32(const float *img, const float *flow, float *out,
                          int B, int C, int Hi, int Wi, int H, int W,
                          int kernel_size, int bilinear)
{
    const long HW = (long)H * W, HWi = (long)Hi * Wi;
#pragma omp parallel for schedule(static)
    for (long index = 0; index < (long)B * C * HW; ++index) {
        const int x = (int)(index % W), y = (int)((index / W) % H);
        const int c = (int)((index / HW) % C), b = (int)(index / (HW * C));
        const float dx = flow[((long)b * 2 + 0) * HW + (long)y * W + x];
        const float dy = flow[((long)b * 2 + 1) * HW + (long)y * W + x];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
        const float *I = img + ((long)b * C + c) * HWi;
        float val = 0.0f;
        if (bilinear) {
            /* indices clamped with the OUTPUT dims (:49-52) */
            const int xL = clampi(f2i_sat(floorf(xf)), 0, W - 1);
            const int xR = clampi(f2i_sat(floorf(xf) + 1), 0, W - 1);
            const int yT = clampi(f2i_sat(floorf(yf)), 0, H - 1);
            const int yB = clampi(f2i_sat(floorf(yf) + 1), 0, H - 1);
            for (int fy = 0; fy < kernel_size; ++fy)
                for (int fx = 0; fx < kernel_size; ++fx) {
                    const int yt = clampi(yT + fy, 0, Hi - 1), yb = clampi(yB + fy, 0, Hi - 1);
                    const int xl = clampi(xL + fx, 0, Wi - 1), xr = clampi(xR + fx, 0, Wi - 1);
                    /* weights formed in double ("1." literals), product cast to
                     * float, float accumulation (:56-59) */
                    val += (float)((1. - alpha) * (1. - beta) * I[(long)yt * Wi + xl]);
                    val += (float)((alpha) * (1. - beta) * I[(long)yt * Wi + xr]);
                    val += (float)((1. - alpha) * (beta)*I[(long)yb * Wi + xl]);
                    val += (float)((alpha) * (beta)*I[(long)yb * Wi + xr]);
                }
            out[index] = val;
        } else {
            /* floor(xf + 0.5): double add, double floor (:66-67) */
            const double xn = floor((double)xf + 0.5), yn = floor((double)yf + 0.5);
            int xN = (xn != xn) ? 0 : (xn >= 2147483647.0 ? 2147483647 : (xn <= -2147483648.0 ? (-2147483647 - 1) : (int)xn));
            int yN = (yn != yn) ? 0 : (yn >= 2147483647.0 ? 2147483647 : (yn <= -2147483648.0 ? (-2147483647 - 1) : (int)yn));
            xN = clampi(xN, 0, W - 1);
            yN = clampi(yN, 0, H - 1);
            out[index] = I[(long)clampi(yN, 0, Hi - 1) * Wi + clampi(xN, 0, Wi - 1)];
        }
    }
    return 0;
}

/* kernel_resample2d_backward_input1<float> (:75-125): scatter-add into gimg,
 * which this function zero-fills first (the reference's caller passes a
 * zero-filled tensor, resample2d.py:31).  The reference scatters with
 * atomicAdd in an undefined order; the oracle adds in thread-index order, so
 * comparisons against it are tolerance-based (race row).
 * kernel_resample2d_backward_input2<float> (:127-198): gather. */
int fn2o_resample_bwd_f32(const float *img, const float *flow, const float *gout,
                          float *gimg, float *gflow,
                          int B, int C, int Hi, int Wi, int H, int W,
                          int kernel_size, int bilinear)
{
    (void)bilinear; /* both backward kernels ignore the flag */
    const long HW = (long)H * W, HWi = (long)Hi * Wi;
    memset(gimg, 0, sizeof(float) * (size_t)B * C * HWi);
    /* ---- input1: serial over the thread index to keep a defined order; parallel
     * over (b, c) planes, which never collide. */
#pragma omp parallel for schedule(static)
    for (long plane = 0; plane < (long)B * C; ++plane) {
        const int b = (int)(plane / C);
        float *G = gimg + plane * HWi;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float dx = flow[((long)b * 2 + 0) * HW + (long)y * W + x];
                const float dy = flow[((long)b * 2 + 1) * HW + (long)y * W + x];
                const float xf = (float)x + dx, yf = (float)y + dy;
                /* alpha = xf - int(xf): truncation, not floor (:105-106) */
                const float alpha = xf - (float)f2i_sat(xf), beta = yf - (float)f2i_sat(yf);
                /* corner indices clamped with INPUT1 dims (:108-114) */
                const int xL = clampi(f2i_sat(floorf(xf)), 0, Wi - 1);
                const int xR = clampi(f2i_sat(floorf(xf) + 1), 0, Wi - 1);
                const int yT = clampi(f2i_sat(floorf(yf)), 0, Hi - 1);
                const int yB = clampi(f2i_sat(floorf(yf) + 1), 0, Hi - 1);
                const float go = gout[plane * HW + (long)y * W + x];
                for (int fy = 0; fy < kernel_size; ++fy)
                    for (int fx = 0; fx < kernel_size; ++fx) {
                        const int yt = clampi(yT + fy, 0, Hi - 1), yb = clampi(yB + fy, 0, Hi - 1);
                        const int xl = clampi(xL + fx, 0, Wi - 1), xr = clampi(xR + fx, 0, Wi - 1);
                        G[(long)yt * Wi + xl] += (1 - alpha) * (1 - beta) * go; /* float math (:118-121) */
                        G[(long)yt * Wi + xr] += (alpha) * (1 - beta) * go;
                        G[(long)yb * Wi + xl] += (1 - alpha) * (beta)*go;
                        G[(long)yb * Wi + xr] += (alpha) * (beta)*go;
                    }
            }
    }
    /* ---- input2 (flow): one value per (b, c in {0,1}, y, x); indices clamped
     * with the FLOW dims (:163-166); float accumulation in the order written. */
    const int kernel_rad = (kernel_size - 1) / 2;
#pragma omp parallel for schedule(static)
    for (long index = 0; index < (long)B * 2 * HW; ++index) {
        const int x = (int)(index % W), y = (int)((index / W) % H);
        const int c = (int)((index / HW) % 2), b = (int)(index / (HW * 2));
        const float dx = flow[((long)b * 2 + 0) * HW + (long)y * W + x];
        const float dy = flow[((long)b * 2 + 1) * HW + (long)y * W + x];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const int xL = clampi(f2i_sat(floorf(xf)), 0, W - 1);
        const int xR = clampi(f2i_sat(floorf(xf) + 1), 0, W - 1);
        const int yT = clampi(f2i_sat(floorf(yf)), 0, H - 1);
        const int yB = clampi(f2i_sat(floorf(yf) + 1), 0, H - 1);
        float output = 0.0f;
        if (c % 2) { /* d/d(dy) (:168-179) */
            const float gamma = 1 - (xf - floorf(xf));
            for (int i = 0; i <= 2 * kernel_rad; ++i)
                for (int j = 0; j <= 2 * kernel_rad; ++j)
                    for (int ch = 0; ch < C; ++ch) {
                        const float go = gout[((long)b * C + ch) * HW + (long)y * W + x];
                        const float *I = img + ((long)b * C + ch) * HWi;
                        const int yb = clampi(yB + j, 0, Hi - 1), yt = clampi(yT + j, 0, Hi - 1);
                        const int xl = clampi(xL + i, 0, Wi - 1), xr = clampi(xR + i, 0, Wi - 1);
                        output += (gamma)*go * I[(long)yb * Wi + xl];
                        output -= (gamma)*go * I[(long)yt * Wi + xl];
                        output += (1 - gamma) * go * I[(long)yb * Wi + xr];
                        output -= (1 - gamma) * go * I[(long)yt * Wi + xr];
                    }
        } else { /* d/d(dx) (:181-192) */
            const float gamma = 1 - (yf - floorf(yf));
            for (int i = 0; i <= 2 * kernel_rad; ++i)
                for (int j = 0; j <= 2 * kernel_rad; ++j)
                    for (int ch = 0; ch < C; ++ch) {
                        const float go = gout[((long)b * C + ch) * HW + (long)y * W + x];
                        const float *I = img + ((long)b * C + ch) * HWi;
                        const int yb = clampi(yB + j, 0, Hi - 1), yt = clampi(yT + j, 0, Hi - 1);
                        const int xl = clampi(xL + i, 0, Wi - 1), xr = clampi(xR + i, 0, Wi - 1);
                        output += (gamma)*go * I[(long)yt * Wi + xr];
                        output -= (gamma)*go * I[(long)yt * Wi + xl];
                        output += (1 - gamma) * go * I[(long)yb * Wi + xr];
                        output -= (1 - gamma) * go * I[(long)yb * Wi + xl];
                    }
        }
        gflow[index] = output;
    }
    return 0;
}
