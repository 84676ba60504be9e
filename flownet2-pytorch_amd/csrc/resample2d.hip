// resample2d.hip -- Resample2d (bilinear / nearest flow warp) forward and backward for gfx950.
//
// Replaces reference kernels kernel_resample2d_update_output (resample2d_kernel.cu:15-72),
// kernel_resample2d_backward_input1 (:75-125) and kernel_resample2d_backward_input2 (:127-198).
//
// HBM-bound gathers.  Unlike the reference (one thread per output ELEMENT: the flow is re-read
// and the weights re-derived once per channel), one lane owns PX consecutive output pixels and
// walks the image channels: flow is read once with a 16 B load, weights / corner offsets are
// formed once, every output plane gets a coalesced 16 B store.  Algorithmic bytes (fp32):
//   fwd  (2C + 2) * B*H*W * 4      bwd  (3C + 4) * B*H*W * 4  (+ the caller's zero-fill of grad_img)
// The backward pass is ONE kernel (the reference launches two, each re-reading flow, grad_out and
// the image): per pixel it scatters grad_out into the four corners of grad_img with hardware fp32
// atomics and forms both flow-gradient components from the same loaded corners.
// Arithmetic order follows the reference so the gather results are bit-identical to its source
// semantics: bilinear weights in double, each term rounded to float, float accumulation.
#include <type_traits>

#include "fn2_common.h"
#include "fn2_debug.h"
#include "corr_params.h"

namespace fn2 {

typedef float __attribute__((ext_vector_type(4))) f4;

struct ImgStrides { long b, c, h, w; };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// int(floor(xf)) with CUDA's saturating float->int conversion (cvt.rzi.s32.f32; NaN -> 0).
__device__ __forceinline__ int f2i_sat(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483520.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
__device__ __forceinline__ int d2i_sat(double v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

// ---------------------------------------------------------------- forward
template <int PX>
__global__ __launch_bounds__(256) void resample_fwd_kernel(const float *__restrict__ img, ImgStrides is,
                                                           const float *__restrict__ flow, float *__restrict__ out,
                                                           int C, int Hi, int Wi, int H, int W, long ngroups,
                                                           int bilinear)
{
    const long HW = (long)H * W;
    const int gpr = W / PX; // groups per row (W % PX == 0 when PX > 1)
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < ngroups; g += (long)gridDim.x * blockDim.x) {
        const int x0 = (int)(g % gpr) * PX;
        const long row = g / gpr;
        const int y = (int)(row % H);
        const int b = (int)(row / H);
        const long fo = (long)b * 2 * HW + (long)y * W + x0;
        float dx[PX], dy[PX];
        if constexpr (PX == 4) {
            const f4 vx = *reinterpret_cast<const f4 *>(flow + fo);
            const f4 vy = *reinterpret_cast<const f4 *>(flow + fo + HW);
#pragma unroll
            for (int i = 0; i < 4; ++i) { dx[i] = vx[i]; dy[i] = vy[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < PX; ++i) { dx[i] = flow[fo + i]; dy[i] = flow[fo + HW + i]; }
        }
        long o00[PX], o01[PX], o10[PX], o11[PX];
        double w00[PX], w01[PX], w10[PX], w11[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const float xf = (float)(x0 + i) + dx[i], yf = (float)y + dy[i];
            if (bilinear) {
                const float fx = floorf(xf), fy = floorf(yf);
                const float alpha = xf - fx, beta = yf - fy; // (:45-46)
                // indices clamped with the OUTPUT dims (:49-52), then to the image (defensive)
                const int xL = clampi(clampi(f2i_sat(fx), 0, W - 1), 0, Wi - 1);
                const int xR = clampi(clampi(f2i_sat(fx + 1.0f), 0, W - 1), 0, Wi - 1);
                const int yT = clampi(clampi(f2i_sat(fy), 0, H - 1), 0, Hi - 1);
                const int yB = clampi(clampi(f2i_sat(fy + 1.0f), 0, H - 1), 0, Hi - 1);
                o00[i] = yT * is.h + xL * is.w;
                o01[i] = yT * is.h + xR * is.w;
                o10[i] = yB * is.h + xL * is.w;
                o11[i] = yB * is.h + xR * is.w;
                const double a = (double)alpha, be = (double)beta; // "1." literals -> double (:56-59)
                w00[i] = (1. - a) * (1. - be);
                w01[i] = a * (1. - be);
                w10[i] = (1. - a) * be;
                w11[i] = a * be;
            } else {
                // nearest: floor(xf + 0.5) in double (:66-67)
                const int xN = clampi(clampi(d2i_sat(floor((double)xf + 0.5)), 0, W - 1), 0, Wi - 1);
                const int yN = clampi(clampi(d2i_sat(floor((double)yf + 0.5)), 0, H - 1), 0, Hi - 1);
                o00[i] = yN * is.h + xN * is.w;
                o01[i] = o10[i] = o11[i] = o00[i];
                w00[i] = 1.;
                w01[i] = w10[i] = w11[i] = 0.;
            }
        }
        const float *ib = img + (long)b * is.b;
        float *ob = out + (long)b * C * HW + (long)y * W + x0;
        for (int c = 0; c < C; ++c) {
            const float *ic = ib + (long)c * is.c;
            float v[PX];
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                if (bilinear) {
                    float val = 0.0f;
                    val = val + (float)(w00[i] * (double)ic[o00[i]]);
                    val = val + (float)(w01[i] * (double)ic[o01[i]]);
                    val = val + (float)(w10[i] * (double)ic[o10[i]]);
                    val = val + (float)(w11[i] * (double)ic[o11[i]]);
                    v[i] = val;
                } else {
                    v[i] = ic[o00[i]];
                }
            }
            if constexpr (PX == 4) {
                f4 r = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f4 *>(ob + (long)c * HW) = r;
            } else {
#pragma unroll
                for (int i = 0; i < PX; ++i) ob[(long)c * HW + i] = v[i];
            }
        }
    }
}

// ---------------------------------------------------------------- backward (fused input1 + input2)
// One lane per output pixel, lanes along x (coalesced flow / grad_out reads, coalesced grad_flow
// writes).  grad_img scatter: 4 fp32 hardware atomics per (pixel, channel); neighbouring lanes
// mostly hit neighbouring addresses, which the memory pipeline merges per cache line.
__global__ __launch_bounds__(256) void resample_bwd_kernel(const float *__restrict__ img, ImgStrides is,
                                                           const float *__restrict__ flow,
                                                           const float *__restrict__ gout,
                                                           float *__restrict__ gimg, float *__restrict__ gflow,
                                                           int C, int Hi, int Wi, int H, int W, long npix)
{
    const long HW = (long)H * W, HWi = (long)Hi * Wi;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < npix; g += (long)gridDim.x * blockDim.x) {
        const int x = (int)(g % W);
        const long row = g / W;
        const int y = (int)(row % H);
        const int b = (int)(row / H);
        const long p = (long)y * W + x;
        const float dx = flow[(long)b * 2 * HW + p], dy = flow[(long)b * 2 * HW + HW + p];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float fx = floorf(xf), fy = floorf(yf);
        const int ixL = f2i_sat(fx), ixR = f2i_sat(fx + 1.0f), iyT = f2i_sat(fy), iyB = f2i_sat(fy + 1.0f);

        // ---- grad_img: weights use truncation, alpha = xf - int(xf) (:105-106); corners clamped
        //      with the INPUT1 dims (:108-114); all float math (:118-121).
        const float alpha = xf - (float)f2i_sat(xf), beta = yf - (float)f2i_sat(yf);
        const int sxL = clampi(ixL, 0, Wi - 1), sxR = clampi(ixR, 0, Wi - 1);
        const int syT = clampi(iyT, 0, Hi - 1), syB = clampi(iyB, 0, Hi - 1);
        const float s00 = (1 - alpha) * (1 - beta), s01 = alpha * (1 - beta);
        const float s10 = (1 - alpha) * beta, s11 = alpha * beta;

        // ---- grad_flow: corners clamped with the FLOW dims (:163-166) (then to the image, defensive)
        const int gxL = clampi(clampi(ixL, 0, W - 1), 0, Wi - 1), gxR = clampi(clampi(ixR, 0, W - 1), 0, Wi - 1);
        const int gyT = clampi(clampi(iyT, 0, H - 1), 0, Hi - 1), gyB = clampi(clampi(iyB, 0, H - 1), 0, Hi - 1);
        const float gam_y = 1 - (xf - fx); // c == 1 branch: "gamma = 1 - (xf - floor(xf))" (:169)
        const float gam_x = 1 - (yf - fy); // c == 0 branch (:182)
        float out_dx = 0.0f, out_dy = 0.0f;

        for (int ch = 0; ch < C; ++ch) {
            const float go = gout[((long)b * C + ch) * HW + p];
            float *G = gimg + ((long)b * C + ch) * HWi;
            unsafeAtomicAdd(G + (long)syT * Wi + sxL, s00 * go);
            unsafeAtomicAdd(G + (long)syT * Wi + sxR, s01 * go);
            unsafeAtomicAdd(G + (long)syB * Wi + sxL, s10 * go);
            unsafeAtomicAdd(G + (long)syB * Wi + sxR, s11 * go);

            const float *I = img + (long)b * is.b + (long)ch * is.c;
            const float iTL = I[gyT * is.h + gxL * is.w], iTR = I[gyT * is.h + gxR * is.w];
            const float iBL = I[gyB * is.h + gxL * is.w], iBR = I[gyB * is.h + gxR * is.w];
            // d/d(dy)  (:172-177)
            out_dy = out_dy + (gam_y * go) * iBL;
            out_dy = out_dy - (gam_y * go) * iTL;
            out_dy = out_dy + ((1 - gam_y) * go) * iBR;
            out_dy = out_dy - ((1 - gam_y) * go) * iTR;
            // d/d(dx)  (:185-190)
            out_dx = out_dx + (gam_x * go) * iTR;
            out_dx = out_dx - (gam_x * go) * iTL;
            out_dx = out_dx + ((1 - gam_x) * go) * iBR;
            out_dx = out_dx - ((1 - gam_x) * go) * iBL;
        }
        gflow[(long)b * 2 * HW + p] = out_dx;
        gflow[(long)b * 2 * HW + HW + p] = out_dy;
    }
}

// ---------------------------------------------------------------- kernel_size > 1 (window sums)
// The reference adds the four corners at every offset (fy, fx) of a kernel_size x kernel_size window
// (resample2d_kernel.cu:54-61, :116-123) and, in the flow gradient, at the offsets 0 .. 2 * ((kernel_size-1)/2)
// (:171-178, :184-191) -- without any bounds test, i.e. it reads and accumulates outside its tensors near the lower
// and right borders.  Here the shifted indices are clamped to the image (the checker restates it the same way); wherever
// the reference stays inside its tensors the results are the reference's.  FlowNet2 only uses kernel_size 1
// (models.py:48,51), so these are plain one-lane-per-pixel kernels with the reference's operation order.
__global__ __launch_bounds__(256) void resample_fwd_ks_kernel(const float *__restrict__ img, ImgStrides is,
                                                              const float *__restrict__ flow, float *__restrict__ out,
                                                              int C, int Hi, int Wi, int H, int W, long npix, int ks, int bilinear)
{
    const long HW = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < npix; g += (long)gridDim.x * blockDim.x) {
        const int x = (int)(g % W);
        const long row = g / W;
        const int y = (int)(row % H);
        const int b = (int)(row / H);
        const long p = (long)y * W + x;
        const float dx = flow[(long)b * 2 * HW + p], dy = flow[(long)b * 2 * HW + HW + p];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float *ib = img + (long)b * is.b;
        float *ob = out + (long)b * C * HW + p;
        if (!bilinear) {   // nearest ignores kernel_size (:64-69)
            const int xN = clampi(clampi(d2i_sat(floor((double)xf + 0.5)), 0, W - 1), 0, Wi - 1);
            const int yN = clampi(clampi(d2i_sat(floor((double)yf + 0.5)), 0, H - 1), 0, Hi - 1);
            for (int c = 0; c < C; ++c) ob[(long)c * HW] = ib[(long)c * is.c + yN * is.h + xN * is.w];
            continue;
        }
        const float fx0 = floorf(xf), fy0 = floorf(yf);
        const double a = (double)(xf - fx0), be = (double)(yf - fy0);
        const int xL = clampi(f2i_sat(fx0), 0, W - 1), xR = clampi(f2i_sat(fx0 + 1.0f), 0, W - 1);
        const int yT = clampi(f2i_sat(fy0), 0, H - 1), yB = clampi(f2i_sat(fy0 + 1.0f), 0, H - 1);
        const double w00 = (1. - a) * (1. - be), w01 = a * (1. - be), w10 = (1. - a) * be, w11 = a * be;
        for (int c = 0; c < C; ++c) {
            const float *ic = ib + (long)c * is.c;
            float val = 0.0f;
            for (int fy = 0; fy < ks; ++fy)
                for (int fx = 0; fx < ks; ++fx) {
                    const long yt = clampi(yT + fy, 0, Hi - 1) * is.h, yb = clampi(yB + fy, 0, Hi - 1) * is.h;
                    const long xl = clampi(xL + fx, 0, Wi - 1) * is.w, xr = clampi(xR + fx, 0, Wi - 1) * is.w;
                    val = val + (float)(w00 * (double)ic[yt + xl]);
                    val = val + (float)(w01 * (double)ic[yt + xr]);
                    val = val + (float)(w10 * (double)ic[yb + xl]);
                    val = val + (float)(w11 * (double)ic[yb + xr]);
                }
            ob[(long)c * HW] = val;
        }
    }
}

__global__ __launch_bounds__(256) void resample_bwd_ks_kernel(const float *__restrict__ img, ImgStrides is,
                                                              const float *__restrict__ flow, const float *__restrict__ gout,
                                                              float *__restrict__ gimg, float *__restrict__ gflow,
                                                              int C, int Hi, int Wi, int H, int W, long npix, int ks)
{
    const long HW = (long)H * W, HWi = (long)Hi * Wi;
    const int span = 2 * ((ks - 1) / 2);   // the flow gradient walks offsets 0 .. 2 * kernel_rad (:171, :184)
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < npix; g += (long)gridDim.x * blockDim.x) {
        const int x = (int)(g % W);
        const long row = g / W;
        const int y = (int)(row % H);
        const int b = (int)(row / H);
        const long p = (long)y * W + x;
        const float dx = flow[(long)b * 2 * HW + p], dy = flow[(long)b * 2 * HW + HW + p];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float fx0 = floorf(xf), fy0 = floorf(yf);
        const int ixL = f2i_sat(fx0), ixR = f2i_sat(fx0 + 1.0f), iyT = f2i_sat(fy0), iyB = f2i_sat(fy0 + 1.0f);
        // grad_img (:105-123): truncation weights, corners clamped with the image dims, float math
        const float alpha = xf - (float)f2i_sat(xf), beta = yf - (float)f2i_sat(yf);
        const int sxL = clampi(ixL, 0, Wi - 1), sxR = clampi(ixR, 0, Wi - 1);
        const int syT = clampi(iyT, 0, Hi - 1), syB = clampi(iyB, 0, Hi - 1);
        const float s00 = (1 - alpha) * (1 - beta), s01 = alpha * (1 - beta), s10 = (1 - alpha) * beta, s11 = alpha * beta;
        for (int ch = 0; ch < C; ++ch) {
            const float go = gout[((long)b * C + ch) * HW + p];
            float *G = gimg + ((long)b * C + ch) * HWi;
            for (int fy = 0; fy < ks; ++fy)
                for (int fx = 0; fx < ks; ++fx) {
                    const long yt = (long)clampi(syT + fy, 0, Hi - 1) * Wi, yb = (long)clampi(syB + fy, 0, Hi - 1) * Wi;
                    const int xl = clampi(sxL + fx, 0, Wi - 1), xr = clampi(sxR + fx, 0, Wi - 1);
                    unsafeAtomicAdd(G + yt + xl, s00 * go);
                    unsafeAtomicAdd(G + yt + xr, s01 * go);
                    unsafeAtomicAdd(G + yb + xl, s10 * go);
                    unsafeAtomicAdd(G + yb + xr, s11 * go);
                }
        }
        // grad_flow (:163-192): corners clamped with the flow dims, loops i (x offset), j (y offset), channel
        const int gxL = clampi(ixL, 0, W - 1), gxR = clampi(ixR, 0, W - 1), gyT = clampi(iyT, 0, H - 1), gyB = clampi(iyB, 0, H - 1);
        const float gam_y = 1 - (xf - fx0), gam_x = 1 - (yf - fy0);
        float out_dx = 0.0f, out_dy = 0.0f;
        for (int i = 0; i <= span; ++i)
            for (int j = 0; j <= span; ++j)
                for (int ch = 0; ch < C; ++ch) {
                    const float go = gout[((long)b * C + ch) * HW + p];
                    const float *I = img + (long)b * is.b + (long)ch * is.c;
                    const long yb = clampi(gyB + j, 0, Hi - 1) * is.h, yt = clampi(gyT + j, 0, Hi - 1) * is.h;
                    const long xl = clampi(gxL + i, 0, Wi - 1) * is.w, xr = clampi(gxR + i, 0, Wi - 1) * is.w;
                    const float iTL = I[yt + xl], iTR = I[yt + xr], iBL = I[yb + xl], iBR = I[yb + xr];
                    out_dy = out_dy + (gam_y * go) * iBL;
                    out_dy = out_dy - (gam_y * go) * iTL;
                    out_dy = out_dy + ((1 - gam_y) * go) * iBR;
                    out_dy = out_dy - ((1 - gam_y) * go) * iTR;
                    out_dx = out_dx + (gam_x * go) * iTR;
                    out_dx = out_dx - (gam_x * go) * iTL;
                    out_dx = out_dx + ((1 - gam_x) * go) * iBR;
                    out_dx = out_dx - ((1 - gam_x) * go) * iBL;
                }
        gflow[(long)b * 2 * HW + p] = out_dx;
        gflow[(long)b * 2 * HW + HW + p] = out_dy;
    }
}

// ---------------------------------------------------------------- tiled kernels (LDS windows)
// With an arbitrary flow every lane of a wave gathers from / scatters to a different cache line; the per-CU
// vector-memory path then moves a 64-128 B line per 4 useful bytes, and device-scope fp32 atomics to scattered
// addresses leave the XCD (the per-XCD L2s are not coherent) at a few tens of G atomics/s.  The tiled kernels
// give a workgroup a TH x TW tile of output / source pixels and an LDS window covering the tile +- R pixels:
//   forward : the image window is loaded with coalesced 16 B loads, one channel at a time, and the four
//             bilinear corners are gathered from LDS; samples whose corners leave the window read global memory
//   backward: per channel, an image window (for the flow gradient's corner differences) and an accumulation
//             window: the four-corner scatter goes to LDS (compare-and-swap adds), only out-of-window targets use global
//             atomics, and the window is flushed as contiguous rows (exact zeros skipped) -- the windows of
//             neighbouring tiles overlap, so the flush accumulates too.
// Arithmetic and operation order are those of the reference kernels (see resample_fwd_kernel /
// resample_bwd_kernel above, which remain as the fallback for small images and strided pixel layouts).
// fp32 accumulation into LDS.  The native ds_add_f32 runs at ~0.33 lane-atomics/clk/CU on gfx950 (measured,
// scripts/ubench/lds_atomics.hip: integer LDS atomics reach 7-11), a compare-and-swap loop on the same word
// reaches ~2.5 -- 7x faster for the scattered, rarely colliding adds of the warp gradient.
__device__ __forceinline__ void lds_add_f32(float *addr, float v)
{
    unsigned *a = reinterpret_cast<unsigned *>(addr);
    unsigned old = *a, assumed;
    do {
        assumed = old;
        old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v));
    } while (old != assumed);
}

// fp64 cell += fp32 value: ds_add_f64 without return (the product was formed in fp32 as the reference forms it; the
// conversion is exact)
__device__ __forceinline__ void lds_add_f64(double *addr, float v)
{
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)addr, (double)v);
}
__device__ __forceinline__ void lds_add_f64(float *, float) {}   // never called (ACC 0 instantiations)

// FUSE (models.py:133-138): `img` is the second image of a B x 2C x H x W pair tensor `pair`, and the
// kernel writes cat((pair, warped, flow / div_flow, ||pair[:, :C] - warped||_2), 1) = B x (3C+3) x H x W in one pass:
// the warped channel, the copy of both images (the second one comes from the LDS window), the squared difference
// accumulated in channel order as channelnorm_kernel.cu:41-52 does, then the flow and the norm planes.
template <int TH, int TW, int R, int FUSE = 0>   // 0: Resample2d; 1: row N2's concat; 2: the norm plane only (models.py:157-161)
__global__ __launch_bounds__(1024, 8) void resample_fwd_tiled(const float *__restrict__ img, ImgStrides is,
                                                            const float *__restrict__ flow, float *__restrict__ out,
                                                            int C, int Hi, int Wi, int H, int W, int tiles_x, int tiles_y,
                                                            int bilinear, const float *__restrict__ pair = nullptr,
                                                            float div_flow = 1.0f)
{
    // 1024 threads, <= 64 VGPRs: two workgroups (32 waves) per CU.  Per-pixel state is 4 registers: the four corners are
    // base + {0, dx, dy*stride, both} (clamping can only merge neighbours) and the double-precision weights are rebuilt per
    // channel from alpha/beta with the reference's expressions.  The window is double-buffered: channel c+1 is loaded to
    // registers before channel c is gathered and written to the other buffer afterwards -- one barrier per channel.
    constexpr int NT = 1024, WH = TH + 2 * R, WW = TW + 2 * R, PPT = TH * TW / NT;
    __shared__ __attribute__((aligned(16))) float win[2][WH * WW];
    enum { LIVE = 1, IN_WIN = 2, DX = 4, DY = 8 };

    const int tid = threadIdx.x;
    int t = (int)xcd_remap(blockIdx.x, gridDim.x);   // an XCD's workgroups take consecutive tiles: neighbours share its L2
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int X0 = tx * TW, Y0 = ty * TH, wx0 = X0 - R, wy0 = Y0 - R;
    const long HW = (long)H * W;

    float alpha[PPT], beta[PPT];
    float ssq[FUSE ? PPT : 1];
    int base[PPT], flags[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        if (FUSE) ssq[k] = 0.0f;
        alpha[k] = beta[k] = 0.0f;
        base[k] = flags[k] = 0;
        if (!((x < W) && (y < H))) continue;
        const long p = (long)y * W + x;
        const float dx = flow[(long)b * 2 * HW + p], dy = flow[(long)b * 2 * HW + HW + p];
        const float xf = (float)x + dx, yf = (float)y + dy;
        int xL, xR, yT, yB;
        if (bilinear) {
            const float fx = floorf(xf), fy = floorf(yf);
            alpha[k] = xf - fx; beta[k] = yf - fy;                     // (:45-46)
            xL = clampi(clampi(f2i_sat(fx), 0, W - 1), 0, Wi - 1);     // clamped with the OUTPUT dims (:49-52)
            xR = clampi(clampi(f2i_sat(fx + 1.0f), 0, W - 1), 0, Wi - 1);
            yT = clampi(clampi(f2i_sat(fy), 0, H - 1), 0, Hi - 1);
            yB = clampi(clampi(f2i_sat(fy + 1.0f), 0, H - 1), 0, Hi - 1);
        } else {
            xL = xR = clampi(clampi(d2i_sat(floor((double)xf + 0.5)), 0, W - 1), 0, Wi - 1);   // (:66-67)
            yT = yB = clampi(clampi(d2i_sat(floor((double)yf + 0.5)), 0, H - 1), 0, Hi - 1);
        }
        const int lxL = xL - wx0, lxR = xR - wx0, lyT = yT - wy0, lyB = yB - wy0;
        const bool in = (lxL >= 0) && (lxR < WW) && (lyT >= 0) && (lyB < WH);
        base[k] = in ? lyT * WW + lxL : yT * (int)is.h + xL * (int)is.w;
        flags[k] = LIVE | (in ? IN_WIN : 0) | (xR != xL ? DX : 0) | (yB != yT ? DY : 0);
    }

    // window rows are contiguous in the image (pixel stride 1 and Wi % 4 == 0 are launcher preconditions): 16 B per
    // lane, a 4-px group is entirely inside or outside the image
    constexpr int NW = (WH * (WW / 4) + NT - 1) / NT;
    f4 wreg[NW];
    auto win_load = [&](const float *I) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            int i = tid + NT * j;
            asm volatile("" : "+v"(i));    // keep the address arithmetic inside the channel loop (register budget)
            const int ly = i / (WW / 4), lx = (i - ly * (WW / 4)) * 4;
            const int gy = wy0 + ly, gx = wx0 + lx;
            f4 v = (f4){0.0f, 0.0f, 0.0f, 0.0f};
            if (i < WH * (WW / 4) && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
                v = *reinterpret_cast<const f4 *>(I + (long)gy * is.h + gx);
            wreg[j] = v;
        }
    };
    auto win_write = [&](float *dst) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + NT * j;
            if (i < WH * (WW / 4)) *reinterpret_cast<f4 *>(dst + 4 * i) = wreg[j];
        }
    };
    if (C > 0) { win_load(img + (long)b * is.b); win_write(win[0]); }
    __syncthreads();

    for (int c = 0; c < C; ++c) {
        const float *I = img + (long)b * is.b + (long)c * is.c;
        const float *wc = win[c & 1];
        if (c + 1 < C) win_load(I + is.c);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int fl = flags[k], o = base[k], idx = tid + NT * k;
            if (!(fl & LIVE)) continue;
            asm volatile("" : "+v"(fl), "+v"(o), "+v"(idx));   // as above
            const int x = X0 + idx % TW, y = Y0 + idx / TW;
            float i00, i01, i10, i11;
            if (fl & IN_WIN) {
                const int ox = (fl & DX) ? 1 : 0, oy = (fl & DY) ? WW : 0;
                i00 = wc[o]; i01 = wc[o + ox]; i10 = wc[o + oy]; i11 = wc[o + oy + ox];
            } else {
                const int ox = (fl & DX) ? (int)is.w : 0, oy = (fl & DY) ? (int)is.h : 0;
                i00 = I[o]; i01 = I[o + ox]; i10 = I[o + oy]; i11 = I[o + oy + ox];
            }
            float val;
            if (bilinear) {
                const double a = (double)alpha[k], be = (double)beta[k];   // "1." literals -> double (:56-59)
                val = 0.0f;
                val = val + (float)(((1. - a) * (1. - be)) * (double)i00);
                val = val + (float)((a * (1. - be)) * (double)i01);
                val = val + (float)(((1. - a) * be) * (double)i10);
                val = val + (float)((a * be) * (double)i11);
            } else {
                val = i00;
            }
            if (!FUSE) store_out(out + ((long)b * C + c) * HW + (y * W + x), val);
            else {
                const int pix = y * W + x;
                const float v0 = pair[((long)b * 2 * C + c) * HW + pix];
                if (FUSE == 1) {
                    float *ob = out + (long)b * (3 * C + 3) * HW + pix;
                    const float v1 = wc[(y - wy0) * WW + (x - wx0)];        // the pixel itself is always inside the window
                    store_out(ob + (long)c * HW, v0);
                    store_out(ob + (long)(C + c) * HW, v1);
                    store_out(ob + (long)(2 * C + c) * HW, val);
                }
                const float d = v0 - val;                               // models.py:134
                ssq[k] = ssq[k] + d * d;                                // channelnorm_kernel.cu:47-50
            }
        }
        if (c + 1 < C) win_write(win[(c + 1) & 1]);
        __syncthreads();
    }
    if (FUSE) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int idx = tid + NT * k;
            const int x = X0 + idx % TW, y = Y0 + idx / TW;
            if (!(flags[k] & LIVE)) continue;
            const int pix = y * W + x;
            if (FUSE == 2) {                                            // only ||first image - warped||: B x 1 x H x W
                store_out(out + (long)b * HW + pix, __fsqrt_rn(ssq[k]));
                continue;
            }
            float *ob = out + (long)b * (3 * C + 3) * HW + pix;
            // models.py:138 `flow / self.div_flow` on a GPU tensor: PyTorch multiplies by the fp32 reciprocal of a scalar divisor
            const float inv_div = 1.0f / div_flow;
            store_out(ob + (long)(3 * C) * HW, flow[(long)b * 2 * HW + pix] * inv_div);
            store_out(ob + (long)(3 * C + 1) * HW, flow[(long)b * 2 * HW + HW + pix] * inv_div);
            store_out(ob + (long)(3 * C + 2) * HW, __fsqrt_rn(ssq[k]));                      // channelnorm_kernel.cu:52
        }
    }
}

// Forward, 4 ADJACENT pixels per thread: the flow arrives as two 16-byte loads and every output plane leaves as one 16-byte store per
// thread (resample_fwd_tiled: 4-byte accesses strided by the workgroup size -- four times the vector-memory instructions for the
// same bytes, and the CU's vector-memory path retires instructions, not bytes: ~33 cycles each inside a kernel).
// TH x 64 tile, TH * 16 threads; window double-buffered per channel as in resample_fwd_tiled.
template <int TH, int R>
__global__ __launch_bounds__(TH * 16, (2 * TH * 16 + 255) / 256) void resample_fwd_tiled4(const float *__restrict__ img, ImgStrides is,
                                                                                        const float *__restrict__ flow, float *__restrict__ out,
                                                                                        int C, int Hi, int Wi, int H, int W, int tiles_x, int tiles_y,
                                                                                        int bilinear)
{
    constexpr int TW = 64, NT = TH * 16, WH = TH + 2 * R, WW = TW + 2 * R, NW = (WH * (WW / 4) + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float win[2][WH * WW];
    enum { IN_WIN = 2, DX = 4, DY = 8 };
    const int tid = threadIdx.x;
    int t = (int)xcd_remap(blockIdx.x, gridDim.x);   // an XCD's workgroups take consecutive tiles: neighbours share its L2
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int X0 = tx * TW, Y0 = ty * TH, wx0 = X0 - R, wy0 = Y0 - R;
    const long HW = (long)H * W;
    const int x0 = X0 + 4 * (tid & 15), y = Y0 + (tid >> 4);
    const bool live = x0 < W && y < H;                        // W % 4 == 0: a group is entirely inside or outside
    const long p0 = live ? (long)y * W + x0 : 0;
    const f4 vdx = *reinterpret_cast<const f4 *>(flow + (long)b * 2 * HW + p0);
    const f4 vdy = *reinterpret_cast<const f4 *>(flow + (long)b * 2 * HW + HW + p0);
    f4 wreg[NW];
    auto win_load = [&](const float *I) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            int i = tid + NT * j;
            asm volatile("" : "+v"(i));
            const int ly = i / (WW / 4), lx = (i - ly * (WW / 4)) * 4;
            const int gy = wy0 + ly, gx = wx0 + lx;
            f4 v = (f4){0.0f, 0.0f, 0.0f, 0.0f};
            if (i < WH * (WW / 4) && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
                v = *reinterpret_cast<const f4 *>(I + (long)gy * is.h + gx);
            wreg[j] = v;
        }
    };
    auto win_write = [&](float *dst) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + NT * j;
            if (i < WH * (WW / 4)) *reinterpret_cast<f4 *>(dst + 4 * i) = wreg[j];
        }
    };
    if (C > 0) win_load(img + (long)b * is.b);
    float alpha[4], beta[4];
    int base[4], flags[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xf = (float)(x0 + k) + vdx[k], yf = (float)y + vdy[k];
        int xL, xR, yT, yB;
        alpha[k] = beta[k] = 0.0f;
        if (bilinear) {
            const float fx = floorf(xf), fy = floorf(yf);
            alpha[k] = xf - fx; beta[k] = yf - fy;                     // (:45-46)
            xL = clampi(clampi(f2i_sat(fx), 0, W - 1), 0, Wi - 1);     // clamped with the OUTPUT dims (:49-52)
            xR = clampi(clampi(f2i_sat(fx + 1.0f), 0, W - 1), 0, Wi - 1);
            yT = clampi(clampi(f2i_sat(fy), 0, H - 1), 0, Hi - 1);
            yB = clampi(clampi(f2i_sat(fy + 1.0f), 0, H - 1), 0, Hi - 1);
        } else {
            xL = xR = clampi(clampi(d2i_sat(floor((double)xf + 0.5)), 0, W - 1), 0, Wi - 1);   // (:66-67)
            yT = yB = clampi(clampi(d2i_sat(floor((double)yf + 0.5)), 0, H - 1), 0, Hi - 1);
        }
        const int lxL = xL - wx0, lxR = xR - wx0, lyT = yT - wy0, lyB = yB - wy0;
        const bool in = (lxL >= 0) && (lxR < WW) && (lyT >= 0) && (lyB < WH);
        base[k] = in ? lyT * WW + lxL : yT * (int)is.h + xL * (int)is.w;
        flags[k] = (in ? IN_WIN : 0) | (xR != xL ? DX : 0) | (yB != yT ? DY : 0);
    }
    if (C > 0) win_write(win[0]);
    __syncthreads();
    for (int c = 0; c < C; ++c) {
        const float *I = img + (long)b * is.b + (long)c * is.c;
        const float *wc = win[c & 1];
        if (c + 1 < C) win_load(I + is.c);
        f4 res;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int fl = flags[k], o = base[k];
            asm volatile("" : "+v"(fl), "+v"(o));
            float i00, i01, i10, i11;
            if (fl & IN_WIN) {
                const int ox = (fl & DX) ? 1 : 0, oy = (fl & DY) ? WW : 0;
                i00 = wc[o]; i01 = wc[o + ox]; i10 = wc[o + oy]; i11 = wc[o + oy + ox];
            } else {
                const int ox = (fl & DX) ? (int)is.w : 0, oy = (fl & DY) ? (int)is.h : 0;
                i00 = I[o]; i01 = I[o + ox]; i10 = I[o + oy]; i11 = I[o + oy + ox];
            }
            float val;
            if (bilinear) {
                const double a = (double)alpha[k], be = (double)beta[k];   // "1." literals -> double (:56-59)
                val = 0.0f;
                val = val + (float)(((1. - a) * (1. - be)) * (double)i00);
                val = val + (float)((a * (1. - be)) * (double)i01);
                val = val + (float)(((1. - a) * be) * (double)i10);
                val = val + (float)((a * be) * (double)i11);
            } else {
                val = i00;
            }
            res[k] = val;
        }
        if (live) store_out(reinterpret_cast<f4 *>(out + ((long)b * C + c) * HW + p0), res);
        if (c + 1 < C) win_write(win[(c + 1) & 1]);
        __syncthreads();
    }
}

// Forward with ALL image channels of the window resident in LDS (C == NC, typically 3): a workgroup owns a TH x TW tile,
// loads the NC windows at once (one barrier in the whole kernel instead of one per channel), forms the corner offsets and the
// double-precision weights once per pixel and gathers the NC channels back to back.
// One tile of the all-channel forward: rows [Y0, Y0 + TH) x columns [X0, X0 + TW) of item b; `win` holds NC windows of
// (TH + 2R) x (TW + 2R) floats.  The caller provides the barrier before `win` is reused.
template <int TH, int TW, int R, int NC, int NT>
__device__ __forceinline__ void fwd_tile_all(float *__restrict__ win, const float *__restrict__ img, const ImgStrides is,
                                             const float *__restrict__ flow, float *__restrict__ out, int Hi, int Wi, int H, int W,
                                             int b, int X0, int Y0, int ylim, int bilinear, unsigned long long *ts = nullptr)
{
    // `ts` (profiling, debug library): four wall-clock stamps (100 MHz) -- entry, windows written to LDS, barrier, stores issued
    constexpr int WH = TH + 2 * R, WW = TW + 2 * R, PPT = TH * TW / NT, NW = (WH * (WW / 4) + NT - 1) / NT;
    static_assert(TH * TW % NT == 0, "whole pixels per thread");
    const int tid = threadIdx.x;
    if (ts) ts[0] = wall_clock64();
    const int wx0 = X0 - R, wy0 = Y0 - R;
    const long HW = (long)H * W;
    // the flow of the thread's pixels and every window group: all requested before anything is used
    float fdx[PPT], fdy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        const long p = (x < W && y < ylim) ? (long)y * W + x : 0;
        fdx[k] = flow[(long)b * 2 * HW + p]; fdy[k] = flow[(long)b * 2 * HW + HW + p];
    }
    f4 wreg[NC][NW];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + NT * j;
            const int ly = i / (WW / 4), lx = (i - ly * (WW / 4)) * 4;
            const int gy = wy0 + ly, gx = wx0 + lx;
            f4 v = (f4){0.0f, 0.0f, 0.0f, 0.0f};
            if (i < WH * (WW / 4) && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
                v = *reinterpret_cast<const f4 *>(img + (long)b * is.b + (long)c * is.c + (long)gy * is.h + gx);
            wreg[c][j] = v;
        }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + NT * j;
            if (i < WH * (WW / 4)) *reinterpret_cast<f4 *>(win + c * (WH * WW) + 4 * i) = wreg[c][j];
        }
    if (ts) ts[1] = wall_clock64();
    __syncthreads();
    if (ts) ts[2] = wall_clock64();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        if (!((x < W) && (y < ylim))) continue;
        const float xf = (float)x + fdx[k], yf = (float)y + fdy[k];
        int xL, xR, yT, yB;
        float alpha = 0.0f, beta = 0.0f;
        if (bilinear) {
            const float fx = floorf(xf), fy = floorf(yf);
            alpha = xf - fx; beta = yf - fy;                           // (:45-46)
            xL = clampi(clampi(f2i_sat(fx), 0, W - 1), 0, Wi - 1);     // clamped with the OUTPUT dims (:49-52)
            xR = clampi(clampi(f2i_sat(fx + 1.0f), 0, W - 1), 0, Wi - 1);
            yT = clampi(clampi(f2i_sat(fy), 0, H - 1), 0, Hi - 1);
            yB = clampi(clampi(f2i_sat(fy + 1.0f), 0, H - 1), 0, Hi - 1);
        } else {
            xL = xR = clampi(clampi(d2i_sat(floor((double)xf + 0.5)), 0, W - 1), 0, Wi - 1);   // (:66-67)
            yT = yB = clampi(clampi(d2i_sat(floor((double)yf + 0.5)), 0, H - 1), 0, Hi - 1);
        }
        const int lxL = xL - wx0, lxR = xR - wx0, lyT = yT - wy0, lyB = yB - wy0;
        const bool in = (lxL >= 0) && (lxR < WW) && (lyT >= 0) && (lyB < WH);
        const double a = (double)alpha, be = (double)beta;             // "1." literals -> double (:56-59)
        const double w00 = (1. - a) * (1. - be), w01 = a * (1. - be), w10 = (1. - a) * be, w11 = a * be;
        const int o = in ? lyT * WW + lxL : yT * (int)is.h + xL * (int)is.w;
        const int ox = (xR != xL) ? (in ? 1 : (int)is.w) : 0, oy = (yB != yT) ? (in ? WW : (int)is.h) : 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float i00, i01, i10, i11;
            if (in) {
                const float *wc = win + c * (WH * WW);
                i00 = wc[o]; i01 = wc[o + ox]; i10 = wc[o + oy]; i11 = wc[o + oy + ox];
            } else {
                const float *I = img + (long)b * is.b + (long)c * is.c;
                i00 = I[o]; i01 = I[o + ox]; i10 = I[o + oy]; i11 = I[o + oy + ox];
            }
            float val;
            if (bilinear) {
                val = 0.0f;
                val = val + (float)(w00 * (double)i00);
                val = val + (float)(w01 * (double)i01);
                val = val + (float)(w10 * (double)i10);
                val = val + (float)(w11 * (double)i11);
            } else {
                val = i00;
            }
            store_out(out + ((long)b * NC + c) * HW + (y * W + x), val);
        }
    }
    if (ts) ts[3] = wall_clock64();
}

template <int TH, int TW, int R, int NC, int WPE = 4, int NTH = 1024>
__global__ __launch_bounds__(NTH, WPE) void resample_fwd_tiled_all(const float *__restrict__ img, ImgStrides is,
                                                                const float *__restrict__ flow, float *__restrict__ out,
                                                                int Hi, int Wi, int H, int W, int tiles_x, int tiles_y, int bilinear,
                                                                unsigned long long *dbg = nullptr)
{
    __shared__ __attribute__((aligned(16))) float win[NC * (TH + 2 * R) * (TW + 2 * R)];
    int t = (int)xcd_remap(blockIdx.x, gridDim.x);   // an XCD's workgroups take consecutive tiles: neighbours share its L2
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    unsigned long long ts[4] = {0, 0, 0, 0};
    fwd_tile_all<TH, TW, R, NC, NTH>(win, img, is, flow, out, Hi, Wi, H, W, t / tiles_y, tx * TW, ty * TH, H, bilinear, dbg ? ts : nullptr);
    if (dbg && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *d = dbg + (long)blockIdx.x * 8;
        for (int i = 0; i < 4; ++i) d[i] = ts[i];
        d[4] = wall_clock64();
    }
}

// The backward windows FOLLOW the flow.  A tile's accumulation / image window is tile +- R px around where the tile's pixels LAND,
// not around the tile: real flow fields are locally smooth but not small (FlowNet2 warps by the estimated motion, tens of pixels on
// Sintel), and with a window centred on the tile a translation of (25, -18) px sends every corner to the global-atomic path -- 760 us
// instead of 38 for the same field without the translation (8 x 3 x 384 x 512).  The offset is a robust centre of the flow at four
// pixels of the tile (its quadrant centres): per component the mean of the two middle values if they
// agree, so that one or two outliers among the four change nothing -- the plain mean moved 4 % of the tiles of the SURVEY's white-noise flow (1 % of its
// values are x20) far away from their pixels: 51 -> 77 us.  It is rounded -- to a multiple of 4 px in x, so that the 16-byte groups of
// the image window stay aligned -- and zero below 8 px, which leaves small flows exactly where they were.  Only the placement of the
// window depends on it; which pixels take the LDS path never changes a result beyond the order of the fp32 sums of grad_input1.
struct TileFlowSample { float v; };   // lane l holds component (l >> 2) & 1 of sample pixel l & 3
__device__ __forceinline__ void tile_flow_sample(const float *__restrict__ flow_b, long HW, int X0, int Y0, int TW, int TH, int H, int W,
                                                 TileFlowSample &S)
{
    // ONE vector load per wave (eight scalar loads per wave through the scalar cache cost the kernel 4 us)
    const int l = threadIdx.x & 7, q = l & 3;
    const int sx = min(X0 + TW / 4 + (q & 1) * (TW / 2), W - 1), sy = min(Y0 + TH / 4 + (q >> 1) * (TH / 2), H - 1);
    S.v = flow_b[((l >> 2) ? HW : 0) + (long)sy * W + sx];
}
// the two middle values of four: their mean if they agree within 8 px, else "no estimate" (two of the four were outliers)
__device__ __forceinline__ float middle_of_four(const float v[4])
{
    const float a = fminf(v[0], v[1]), b = fmaxf(v[0], v[1]), c = fminf(v[2], v[3]), d = fmaxf(v[2], v[3]);
    const float m1 = fmaxf(a, c), m2 = fminf(b, d);
    return fabsf(m1 - m2) < 8.0f ? 0.5f * (m1 + m2) : 0.0f;
}
__device__ __forceinline__ void tile_window_offset(const TileFlowSample &S, int &offx, int &offy, int xq = 4)
{
    float sx[4], sy[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sx[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(S.v), q));
        sy[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(S.v), 4 + q));
    }
    const float mx = middle_of_four(sx), my = middle_of_four(sy);
    offx = (fabsf(mx) >= 8.0f && fabsf(mx) < 1.0e6f) ? xq * (int)rintf(mx / (float)xq) : 0;   // xq = 4 or 16: exact divisions
    offy = (fabsf(my) >= 8.0f && fabsf(my) < 1.0e6f) ? (int)rintf(my) : 0;
}

template <int TH, int TW, int R, int NT, int WPE, int ACC = 0>
__global__ __launch_bounds__(NT, WPE) void resample_bwd_tiled(const float *__restrict__ img, ImgStrides is,
                                                            const float *__restrict__ flow,
                                                            const float *__restrict__ gout,
                                                            float *__restrict__ gimg, float *__restrict__ gflow,
                                                            int C, int Hi, int Wi, int H, int W, int tiles_x, int tiles_y,
                                                            int abl)   // abl: profiling switches (0 in production)
{
    // Two workgroups per CU (<= 64 VGPRs, 2 x 49 KB LDS): one workgroup's window loads overlap the other's LDS scatter.
    // Per-pixel state is therefore kept small: the four corners are base + {0, dx, dy*stride, both} with dx, dy in {0,1}
    // (clamping can only merge neighbours), and the bilinear weights are recomputed per channel from alpha/beta with the
    // reference's expressions.
    constexpr int WH = TH + 2 * R, WW = TW + 2 * R, WWP = WW + 1, PPT = TH * TW / NT;
    __shared__ __attribute__((aligned(16))) float iwin[WH * WW];   // image window
    // accumulation window (+1: rows on different banks).  ACC 0: fp32 cells, added to with a compare-and-swap loop (lds_add_f32);
    // ACC 1: fp64 cells, added to with ds_add_f64 -- no return value, no retry loop, no round trip per add (the returning LDS
    // atomic is what bounds the CAS loop, scripts/ubench/lds_cas_pipelined.hip); the sum is rounded to fp32 once, at the flush.
    typedef std::conditional_t<ACC == 1, double, float> acc_t;
    __shared__ acc_t awin[WH * WWP];
    enum { LIVE = 1, S_IN = 2, G_IN = 4, S_DX = 8, S_DY = 16, G_DX = 32, G_DY = 64 };

    const int tid = threadIdx.x;
    int t = (int)xcd_remap(blockIdx.x, gridDim.x);   // an XCD's workgroups take consecutive tiles: neighbours share its L2
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int X0 = tx * TW, Y0 = ty * TH;
    const long HW = (long)H * W, HWi = (long)Hi * Wi;
    TileFlowSample tfs;
    tile_flow_sample(flow + (long)b * 2 * HW, HW, X0, Y0, TW, TH, H, W, tfs);
    int wx0, wy0;   // set below, once the thread's own loads have been requested

    // per-pixel state (flow is read once)
    float alpha[PPT], beta[PPT], gam_x[PPT], gam_y[PPT], out_dx[PPT], out_dy[PPT];
    int sbase[PPT], gbase[PPT], flags[PPT];
    // image-window staging: a thread owns NW 4-px groups; loaded to registers (so the loads of channel c+1 are in flight
    // while channel c's accumulation window is flushed), then written to LDS
    constexpr int NW = (WH * (WW / 4) + NT - 1) / NT;
    f4 wreg[NW];
    auto win_load = [&](const float *I) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            int i = tid + NT * j;
            asm volatile("" : "+v"(i));    // keep the address arithmetic inside the channel loop (register budget)
            const int ly = i / (WW / 4), lx = (i - ly * (WW / 4)) * 4;
            const int gy = wy0 + ly, gx = wx0 + lx;
            f4 v = (f4){0.0f, 0.0f, 0.0f, 0.0f};
            // Wi % 4 == 0 (launcher): a 4-px group is entirely inside or outside the image
            if (i < WH * (WW / 4) && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
                v = *reinterpret_cast<const f4 *>(I + (long)gy * is.h + gx);
            wreg[j] = v;
        }
    };
    auto win_write = [&]() {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + NT * j;
            if (i < WH * (WW / 4)) *reinterpret_cast<f4 *>(iwin + 4 * i) = wreg[j];
        }
    };
    // all flow values of the thread are requested before the first one is used (one memory round trip instead of PPT); the
    // window's place follows from the tile's flow sample, requested before them, and the first image window goes out right behind
    float fdx[PPT], fdy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        const bool in = (x < W) && (y < H);
        const long p = in ? (long)y * W + x : 0;
        fdx[k] = flow[(long)b * 2 * HW + p]; fdy[k] = flow[(long)b * 2 * HW + HW + p];
    }
    wx0 = X0 - R; wy0 = Y0 - R;
    if (C > 0) win_load(img + (long)b * is.b);
    __builtin_amdgcn_sched_barrier(0);
    {   // requested where the tile is, and once more if the flow sample says its pixels land elsewhere (never for flows below 8 px)
        int offx, offy;
        tile_window_offset(tfs, offx, offy);
        if (offx | offy) {
            wx0 += offx; wy0 += offy;
            if (C > 0) win_load(img + (long)b * is.b);
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        out_dx[k] = out_dy[k] = 0.0f;
        alpha[k] = beta[k] = gam_x[k] = gam_y[k] = 0.0f;
        sbase[k] = gbase[k] = 0;
        flags[k] = 0;
        if (!((x < W) && (y < H))) continue;
        int fl = LIVE;
        const float dx = fdx[k], dy = fdy[k];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float fx = floorf(xf), fy = floorf(yf);
        const int ixL = f2i_sat(fx), ixR = f2i_sat(fx + 1.0f), iyT = f2i_sat(fy), iyB = f2i_sat(fy + 1.0f);
        alpha[k] = xf - (float)f2i_sat(xf); beta[k] = yf - (float)f2i_sat(yf);   // truncation (:105-106)
        gam_y[k] = 1 - (xf - fx);   // c == 1 branch (:169)
        gam_x[k] = 1 - (yf - fy);   // c == 0 branch (:182)
        {   // scatter corners: clamped with the INPUT1 dims (:108-114)
            const int xL = clampi(ixL, 0, Wi - 1), xR = clampi(ixR, 0, Wi - 1);
            const int yT = clampi(iyT, 0, Hi - 1), yB = clampi(iyB, 0, Hi - 1);
            const int lxL = xL - wx0, lxR = xR - wx0, lyT = yT - wy0, lyB = yB - wy0;
            const bool in = (lxL >= 0) && (lxR < WW) && (lyT >= 0) && (lyB < WH);
            sbase[k] = in ? lyT * WWP + lxL : yT * Wi + xL;
            fl |= (in ? S_IN : 0) | (xR != xL ? S_DX : 0) | (yB != yT ? S_DY : 0);
        }
        {   // gather corners: clamped with the FLOW dims (:163-166), then to the image
            const int xL = clampi(clampi(ixL, 0, W - 1), 0, Wi - 1), xR = clampi(clampi(ixR, 0, W - 1), 0, Wi - 1);
            const int yT = clampi(clampi(iyT, 0, H - 1), 0, Hi - 1), yB = clampi(clampi(iyB, 0, H - 1), 0, Hi - 1);
            const int lxL = xL - wx0, lxR = xR - wx0, lyT = yT - wy0, lyB = yB - wy0;
            const bool in = (lxL >= 0) && (lxR < WW) && (lyT >= 0) && (lyB < WH);
            gbase[k] = in ? lyT * WW + lxL : yT * (int)is.h + xL * (int)is.w;
            fl |= (in ? G_IN : 0) | (xR != xL ? G_DX : 0) | (yB != yT ? G_DY : 0);
        }
        flags[k] = fl;
        __builtin_amdgcn_sched_barrier(0);
    }

    for (int i = tid; i < WH * WWP; i += NT) awin[i] = (acc_t)0;
    win_write();
    __syncthreads();

    for (int c = 0; c < C; ++c) {
        const float *I = img + (long)b * is.b + (long)c * is.c;
        float *G = gimg + ((long)b * C + c) * HWi;
        float gov[PPT];   // the channel's grad_out values of the thread: requested together (one round trip, not PPT)
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int idx = tid + NT * k;
            asm volatile("" : "+v"(idx));
            const int x = X0 + idx % TW, y = Y0 + idx / TW;
            gov[k] = (flags[k] & LIVE) ? gout[((long)b * C + c) * HW + (y * W + x)] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int fl = flags[k], sb = sbase[k], gb = gbase[k];
            if (!(fl & LIVE)) continue;
            // opaque to the optimiser: otherwise every corner address of every pixel is hoisted out of the channel
            // loop and the kernel no longer fits the 64 VGPRs two workgroups per CU need
            asm volatile("" : "+v"(fl), "+v"(sb), "+v"(gb));
            const float go = gov[k];
            const float s00 = (1 - alpha[k]) * (1 - beta[k]), s01 = alpha[k] * (1 - beta[k]);
            const float s10 = (1 - alpha[k]) * beta[k], s11 = alpha[k] * beta[k];
            if (abl & 2) {
            } else if (fl & S_IN) {
                const int ox = (fl & S_DX) ? 1 : 0, oy = (fl & S_DY) ? WWP : 0;
                if constexpr (ACC == 1) {
                    lds_add_f64(awin + sb, s00 * go);
                    lds_add_f64(awin + sb + ox, s01 * go);
                    lds_add_f64(awin + sb + oy, s10 * go);
                    lds_add_f64(awin + sb + oy + ox, s11 * go);
                } else {
                    lds_add_f32(awin + sb, s00 * go);
                    lds_add_f32(awin + sb + ox, s01 * go);
                    lds_add_f32(awin + sb + oy, s10 * go);
                    lds_add_f32(awin + sb + oy + ox, s11 * go);
                }
            } else {
                const int ox = (fl & S_DX) ? 1 : 0, oy = (fl & S_DY) ? Wi : 0;
                unsafeAtomicAdd(G + sb, s00 * go);
                unsafeAtomicAdd(G + sb + ox, s01 * go);
                unsafeAtomicAdd(G + sb + oy, s10 * go);
                unsafeAtomicAdd(G + sb + oy + ox, s11 * go);
            }
            float iTL, iTR, iBL, iBR;
            if (abl & 4) { iTL = iTR = iBL = iBR = go; }
            else if (fl & G_IN) {
                const int ox = (fl & G_DX) ? 1 : 0, oy = (fl & G_DY) ? WW : 0;
                iTL = iwin[gb]; iTR = iwin[gb + ox]; iBL = iwin[gb + oy]; iBR = iwin[gb + oy + ox];
            } else {
                const int ox = (fl & G_DX) ? (int)is.w : 0, oy = (fl & G_DY) ? (int)is.h : 0;
                iTL = I[gb]; iTR = I[gb + ox]; iBL = I[gb + oy]; iBR = I[gb + oy + ox];
            }
            out_dy[k] = out_dy[k] + (gam_y[k] * go) * iBL;       // (:172-177)
            out_dy[k] = out_dy[k] - (gam_y[k] * go) * iTL;
            out_dy[k] = out_dy[k] + ((1 - gam_y[k]) * go) * iBR;
            out_dy[k] = out_dy[k] - ((1 - gam_y[k]) * go) * iTR;
            out_dx[k] = out_dx[k] + (gam_x[k] * go) * iTR;       // (:185-190)
            out_dx[k] = out_dx[k] - (gam_x[k] * go) * iTL;
            out_dx[k] = out_dx[k] + ((1 - gam_x[k]) * go) * iBR;
            out_dx[k] = out_dx[k] - ((1 - gam_x[k]) * go) * iBL;
            __builtin_amdgcn_sched_barrier(0);   // one pixel at a time: keeps the live set small
        }
        __syncthreads();
        if (c + 1 < C) win_load(I + is.c);
        // flush the accumulation window as contiguous rows (exact zeros skipped) and clear it for the next channel
        {
            int ly = tid / WW, lx = tid - ly * WW;     // element tid, then steps of NT without divisions
#pragma unroll 2
            for (int i = tid; i < WH * WW; i += NT) {
                const int gx = wx0 + lx, gy = wy0 + ly;
                const acc_t v = awin[ly * WWP + lx];
                if (v != (acc_t)0) {
                    awin[ly * WWP + lx] = (acc_t)0;
                    if (!(abl & 1) && gx >= 0 && gx < Wi && gy >= 0 && gy < Hi) unsafeAtomicAdd(G + gy * Wi + gx, (float)v);
                }
                ly += NT / WW; lx += NT % WW;
                if (lx >= WW) { lx -= WW; ++ly; }
            }
        }
        if (c + 1 < C) win_write();
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (!(flags[k] & LIVE)) continue;
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        const long p = (long)y * W + x;
        store_out(gflow + (long)b * 2 * HW + p, out_dx[k]);
        store_out(gflow + (long)b * 2 * HW + HW + p, out_dy[k]);
    }
}

// ---------------------------------------------------------------- backward, three channels in one scatter
// resample_bwd_tiled above walks the channels one at a time: per channel it rebuilds the weights and corner addresses of
// every pixel, scatters, flushes, and reloads the image window -- seven barriers for C = 3.  FlowNet2 only warps 3-channel
// images (models.py:133-174), so this variant keeps ALL THREE accumulation windows in LDS (fp32 cells, 3 x 24.8 KB = 74.5 KB:
// still two workgroups per CU): a pixel's weights and addresses are formed once and its adds issued back to back --
// channels 0 and 1 of a cell share one 64-bit word updated by ONE compare-and-swap, channel 2 has a word of its own: 8 LDS
// atomics per pixel instead of 12 --; one flush pass; then the three image windows take the same LDS bytes and grad_flow is
// gathered for all channels.  Four barriers.  Arithmetic and operation order per output as resample_bwd_tiled with fp32 cells
// (the kernel of round 3); grad_flow is bit-identical to it.
// Measured (scripts/resample_micro.py, 8 x 3 x 384 x 512, us; white-noise / smooth flow): loads, barriers and zeroing alone
// 17 (resample_bwd_tiled: 26), + scatter 14 / 10, + flush 19 / 12: the flush is the largest part (why: the atomic-request bound in
// the kernel's own comment below).  (Half of the workgroups running the gather first, so that their flushes pass
// under the others' scatters, gained 3 us on the smooth flow and nothing on the white-noise one, and the kernel's time moved by
// +-3 us with the mere layout of that second path's code: one order, straight-line text.)
__device__ __forceinline__ void lds_add_f32x2(unsigned long long *a, float v0, float v1)
{
    // (starting from an assumed 0 instead of this read -- a third of the adds find an untouched cell -- measured 1 us slower:
    // a failed compare-and-swap costs more than the read it replaces, profiles/r05_e_caszero.log)
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        const float lo = __uint_as_float((unsigned)assumed) + v0, hi = __uint_as_float((unsigned)(assumed >> 32)) + v1;
        old = atomicCAS(a, assumed, ((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo));
    } while (old != assumed);
}

// The 12 adds of one pixel into the three accumulation windows.  `pile`: a corner was clamped to the image border in a wave where
// at least 8 lanes were -- clamping sends whole runs of pixels to ONE cell (a translation of 25 px puts 25 lanes of a wave on the last
// column; the stray outlier of a noisy flow does not count: a wave that takes this path pays for it with all its lanes).  A
// compare-and-swap loop retries once per colliding lane and a round trip each -- 790 us for the translated field of tile_window_offset's note --, so these
// pixels use the LDS's own fp32 add: 0.33 lane-atomics/clk/CU, 7x slower than the loop without collisions, but collisions are
// resolved inside the LDS unit.  It may touch one half of a 64-bit pair another lane updates by compare-and-swap: both are single
// LDS operations, and the swap fails and retries if the word changed under it.
__device__ __forceinline__ void c3_add_pixel(unsigned long long *a01, float *a2, int sb, int ox, int oy, float s00, float s01,
                                             float s10, float s11, float g0, float g1, float g2, bool pile)
{
    if (!pile) {
        lds_add_f32x2(a01 + sb, s00 * g0, s00 * g1);
        lds_add_f32x2(a01 + sb + ox, s01 * g0, s01 * g1);
        lds_add_f32x2(a01 + sb + oy, s10 * g0, s10 * g1);
        lds_add_f32x2(a01 + sb + oy + ox, s11 * g0, s11 * g1);
        lds_add_f32(a2 + sb, s00 * g2);
        lds_add_f32(a2 + sb + ox, s01 * g2);
        lds_add_f32(a2 + sb + oy, s10 * g2);
        lds_add_f32(a2 + sb + oy + ox, s11 * g2);
    } else {
        // corners that clamping merged into one cell are added once (their weights summed first: the reference issues them as
        // separate atomics in no particular order)
        typedef __attribute__((address_space(3))) float lds_float;
        lds_float *p01 = (lds_float *)a01, *p2 = (lds_float *)a2;
        float a = s00, b = s01, c = s10, d = s11;
        if (!ox) { a += b; c += d; }
        if (!oy) { a += c; b += d; }
        auto add = [&](int cell, float w) {
            __builtin_amdgcn_ds_faddf(p01 + 2 * cell, w * g0, 0, 0, false);
            __builtin_amdgcn_ds_faddf(p01 + 2 * cell + 1, w * g1, 0, 0, false);
            __builtin_amdgcn_ds_faddf(p2 + cell, w * g2, 0, 0, false);
        };
        add(sb, a);
        if (ox) add(sb + ox, b);
        if (oy) add(sb + oy, c);
        if (ox && oy) add(sb + oy + ox, d);
    }
}

// ---------------------------------------------------------------- backward, three channels: the kernel
// The scheme described above (all three accumulation windows in LDS, one scatter, one flush, then the image windows and the gather),
// its inputs in one argument block, and two options (round 5):
//   FUSED (row N2, models.py:133-138 differentiated): the gradient of the warped image is formed on the fly from the concat
//     gradient -- g_warped = g_cat[6:9] - g_norm * diff / (norm + 1e-9) (channelnorm_kernel.cu:93), diff = first image - warped
//     image read back from the forward's concat buffer --, the first image's gradient g_cat[0:3] + g_diff is written on the way, and
//     the flow gradient gets the flow / div_flow term added;
//   SCATTER = false when the image pair needs no gradient (it is the network's input): gather only, no LDS atomics, no global
//     atomics, 73.7 KB of LDS.
// The window is aligned to 64 bytes in x (tile_window_offset rounds to 16 px here): a flushed row is 6 atomic requests, never 7.
// What bounds the scattering form (scripts/ubench/atomic_rate.hip, flush_probe.hip): global fp32 atomics on gfx950 are
// performed outside the XCD whatever their scope -- workgroup and agent scope have ONE encoding, and the rate does not depend
// on which XCDs share an image -- at ~20.5 G REQUESTS/s chip-wide, a request being the lanes of one instruction that fall into one
// aligned 64-byte segment (16 consecutive floats cost what a single float costs).  8 x 3 x 384 x 512 with the white-noise flow:
// 0.59 M flush requests + 0.20 M from the 18 k far pixels = 39 us of atomic-unit time, which cannot start before the first
// windows are complete (timeline, scripts/resample_timeline.py: loads 3 us, then the scatter -- 33 k LDS compare-and-swap pairs per CU
// at 2.5 per clock: 8-13 us): 50-53 us.  (Delaying the second workgroup of every CU by 2-10 us so that one scatters while the other
// flushes changes nothing: 53.1-54.7 us, profiles/r05_d_resample_stagger.log.)  Measured and dropped in round 5 (scripts/attic/resample_bwd_fill_farlist.hip.txt,
// profiles/r05_b_resample_fill_farlist.log): far pixels collected in an LDS list and issued with the two corners of a row in
// adjacent lanes (half the far requests, but issued in the flush phase instead of trickling out while the atomic unit idles:
// 59.6 us against 52.7) and the zero fill of grad_input1 folded into the kernel behind claim / completion counters (four
// dependent memory round trips per workgroup cost more than the 5.7 us fill they replace: 69.0 us against 58.1 with the fill).
// Arithmetic and operation order per output as resample_bwd_tiled with fp32 cells; grad_flow is bit-identical to it.
struct C3xArgs {
    const float *img; ImgStrides is;          // the image that was warped (FUSED: pair + 3 HW, batch stride 6 HW)
    const float *flow;
    const float *gout;                        // !FUSED: B x 3 x H x W gradient of the warped image
    const float *gcat, *pair, *outcat;        // FUSED: B x 12 x H x W concat gradient, B x 6 x H x W images, B x 12 x H x W forward output
    float *gpair0;                            // FUSED: B x 6 x H x W gradient of the pair (channels 0..2 written here), or null
    float *gimg; long gimg_bs;                // scatter target, 3 planes of Hi x Wi per item, gimg_bs floats between items
    float *gflow;
    int B, Hi, Wi, H, W, tiles_x, tiles_y, abl;
    int bilinear;                             // FUSED == 2: how the forward sampled (its value is recomputed)
    float inv_div_flow;
    unsigned long long *dbg;                  // profiling (abl & 8): the debug library's stamp buffer (fn2_debug_set_buffer), else null
};

template <int FUSED>
__device__ __forceinline__ void c3x_load_go(const C3xArgs &p, int b, long pix, long HW, float go[3], bool write_gpair0)
{
    if constexpr (FUSED == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) go[c] = p.gout[((long)b * 3 + c) * HW + pix];
    } else if constexpr (FUSED == 2) {   // the first image's pixel: the gradient is formed at the gather, from the recomputed warp
#pragma unroll
        for (int c = 0; c < 3; ++c) go[c] = p.pair[((long)b * 6 + c) * HW + pix];
    } else {
        const float gn = p.gcat[((long)b * 12 + 11) * HW + pix], nrm = p.outcat[((long)b * 12 + 11) * HW + pix];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float diff = p.pair[((long)b * 6 + c) * HW + pix] - p.outcat[((long)b * 12 + 6 + c) * HW + pix];   // (models.py:134)
            const float gd = chnorm_grad(gn, diff, nrm);
            go[c] = p.gcat[((long)b * 12 + 6 + c) * HW + pix] - gd;
            if (write_gpair0 && p.gpair0) store_out(p.gpair0 + ((long)b * 6 + c) * HW + pix, p.gcat[((long)b * 12 + c) * HW + pix] + gd);
        }
    }
}

// FUSED: 0 = Resample2d's backward; 1 = row N2's concat (above); 2 = the norm-only form (models.py:157-161, :170-174:
// ||first image - warped||_2 with no concat): gcat / outcat are the B x 1 x H x W gradient of the norm and the norm, the warped image
// was never stored and is recomputed at the gather from the same corners with the forward's arithmetic; gather only (SCATTER false).
template <int TH, int TW, int R, int NT, int FUSED, bool SCATTER>
__global__ __launch_bounds__(NT, 8) void resample_bwd_c3x(const C3xArgs p)
{
    static_assert(FUSED != 2 || !SCATTER, "the norm-only form is gather-only");
    constexpr int WH = TH + 2 * R, WW = TW + 2 * R, WWP = WW + 1, PPT = TH * TW / NT, CELLS = WH * WWP, C = 3;
    constexpr int NW = (WH * (WW / 4) + NT - 1) / NT;
    constexpr int WIN_BYTES = SCATTER ? CELLS * 12 : 3 * WH * WW * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[WIN_BYTES];
    float *const aw = reinterpret_cast<float *>(smem);
    unsigned long long *const a01 = reinterpret_cast<unsigned long long *>(smem);
    float *const a2 = reinterpret_cast<float *>(smem + CELLS * 8);
    float *const iwin = reinterpret_cast<float *>(smem);
    static_assert(3 * WH * WW * 4 <= WIN_BYTES, "image windows must fit the accumulation windows' bytes");
    enum { LIVE = 1, G_IN = 4, G_DX = 32, G_DY = 64 };

    const int tid = threadIdx.x;
    int t = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int X0 = tx * TW, Y0 = ty * TH;
    const int H = p.H, W = p.W, Hi = p.Hi, Wi = p.Wi;
    const long HW = (long)H * W, HWi = (long)Hi * Wi;
    const ImgStrides is = p.is;
    const float *const img = p.img;
    float *const gimg_b = p.gimg + (long)b * p.gimg_bs;
    TileFlowSample tfs;
    tile_flow_sample(p.flow + (long)b * 2 * HW, HW, X0, Y0, TW, TH, H, W, tfs);
    // profiling (abl & 8, debug library only): wall-clock stamps (100 MHz) of the workgroup's phases, dumped into the debug buffer
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) __attribute__((always_inline)) { if ((p.abl & 8) && p.dbg) ts[i] = wall_clock64(); };
    stamp(0);

    float fdx[PPT], fdy[PPT], go[PPT][C];
    float gnv[FUSED == 2 ? PPT : 1], nrv[FUSED == 2 ? PPT : 1], alv[FUSED == 2 ? PPT : 1], btv[FUSED == 2 ? PPT : 1];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        const bool live = (x < W) && (y < H);
        const long pix = live ? (long)y * W + x : 0;
        fdx[k] = p.flow[(long)b * 2 * HW + pix]; fdy[k] = p.flow[(long)b * 2 * HW + HW + pix];
        c3x_load_go<FUSED>(p, b, pix, HW, go[k], live);
        if (FUSED == 2) { gnv[k] = p.gcat[(long)b * HW + pix]; nrv[k] = p.outcat[(long)b * HW + pix]; }
    }
    if (SCATTER)
        for (int i = tid; i < CELLS * 3; i += NT) aw[i] = 0.0f;
    __builtin_amdgcn_sched_barrier(0);   // all of that goes out before the wave waits for its flow sample
    int offx, offy;
    // 16-px steps where a window is flushed (a flushed row is then exactly six 64-byte segments); the gather-only instantiations have no
    // flush to align and keep the finer 4-px steps (2 instead of 8 px of the margin lost at worst: ADVICE r5)
    tile_window_offset(tfs, offx, offy, SCATTER ? 16 : 4);
    const int wx0 = X0 - R + offx, wy0 = Y0 - R + offy;
    if (SCATTER) __syncthreads();
    stamp(1);

    float gam_x[PPT], gam_y[PPT];
    int gbase[PPT], flags[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        gam_x[k] = gam_y[k] = 0.0f;
        gbase[k] = flags[k] = 0;
        if (!((x < W) && (y < H))) continue;
        int fl = LIVE;
        const float xf = (float)x + fdx[k], yf = (float)y + fdy[k];
        const float fx = floorf(xf), fy = floorf(yf);
        const int ixL = f2i_sat(fx), ixR = f2i_sat(fx + 1.0f), iyT = f2i_sat(fy), iyB = f2i_sat(fy + 1.0f);
        gam_y[k] = 1 - (xf - fx);   // c == 1 branch (:169)
        gam_x[k] = 1 - (yf - fy);   // c == 0 branch (:182)
        if (FUSED == 2) { alv[k] = xf - fx; btv[k] = yf - fy; }   // the forward's alpha, beta (resample2d_kernel.cu:45-46)
        {   // gather corners: clamped with the FLOW dims (:163-166), then to the image
            const int xL = clampi(clampi(ixL, 0, W - 1), 0, Wi - 1), xR = clampi(clampi(ixR, 0, W - 1), 0, Wi - 1);
            const int yT = clampi(clampi(iyT, 0, H - 1), 0, Hi - 1), yB = clampi(clampi(iyB, 0, H - 1), 0, Hi - 1);
            const int lxL = xL - wx0, lxR = xR - wx0, lyT = yT - wy0, lyB = yB - wy0;
            const bool in = (lxL >= 0) && (lxR < WW) && (lyT >= 0) && (lyB < WH);
            gbase[k] = in ? lyT * WW + lxL : yT * (int)is.h + xL * (int)is.w;
            fl |= (in ? G_IN : 0) | (xR != xL ? G_DX : 0) | (yB != yT ? G_DY : 0);
        }
        flags[k] = fl;
        if (!SCATTER || (p.abl & 2)) continue;
        // scatter: weights by truncation (:105-106), corners clamped with the INPUT1 dims (:108-114)
        const float alpha = xf - (float)f2i_sat(xf), beta = yf - (float)f2i_sat(yf);
        const float s00 = (1 - alpha) * (1 - beta), s01 = alpha * (1 - beta), s10 = (1 - alpha) * beta, s11 = alpha * beta;
        const int xL = clampi(ixL, 0, Wi - 1), xR = clampi(ixR, 0, Wi - 1), yT = clampi(iyT, 0, Hi - 1), yB = clampi(iyB, 0, Hi - 1);
        const int lxL = xL - wx0, lxR = xR - wx0, lyT = yT - wy0, lyB = yB - wy0;
        if ((lxL >= 0) && (lxR < WW) && (lyT >= 0) && (lyB < WH)) {
            const int sb = lyT * WWP + lxL, ox = (xR != xL) ? 1 : 0, oy = (yB != yT) ? WWP : 0;
            const bool moved = (xL != ixL) || (xR != ixR) || (yT != iyT) || (yB != iyB);   // a corner was clamped to the border
            c3_add_pixel(a01, a2, sb, ox, oy, s00, s01, s10, s11, go[k][0], go[k][1], go[k][2],
                         moved && __popcll(__ballot(moved)) >= 8);
        } else {
            // far pixel: 12 atomics now, while the atomic unit has nothing else to do (the flushes come later)
            const int sb = yT * Wi + xL, ox = (xR != xL) ? 1 : 0, oy = (yB != yT) ? Wi : 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float *G = gimg_b + (long)c * HWi + sb;
                unsafeAtomicAdd(G, s00 * go[k][c]);
                unsafeAtomicAdd(G + ox, s01 * go[k][c]);
                unsafeAtomicAdd(G + oy, s10 * go[k][c]);
                unsafeAtomicAdd(G + oy + ox, s11 * go[k][c]);
            }
        }
    }
    if (SCATTER) __syncthreads();
    stamp(2);

    // the three image windows are requested now: their latency passes under the flush
    f4 wreg[C][NW];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + NT * j;
            const int ly = i / (WW / 4), lx = (i - ly * (WW / 4)) * 4;
            const int gy = wy0 + ly, gx = wx0 + lx;
            f4 v = (f4){0.0f, 0.0f, 0.0f, 0.0f};
            if (i < WH * (WW / 4) && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
                v = *reinterpret_cast<const f4 *>(img + (long)b * is.b + (long)c * is.c + (long)gy * is.h + gx);
            wreg[c][j] = v;
        }
    if constexpr (SCATTER) {
        if (!(p.abl & 1)) {
            int ly = tid / WW, lx = tid - ly * WW;
#pragma unroll 2
            for (int i = tid; i < WH * WW; i += NT) {
                const int gx = wx0 + lx, gy = wy0 + ly;
                const int cell = ly * WWP + lx;
                float v0, v1, v2;
                {
                    const unsigned long long u = a01[cell];
                    v0 = __uint_as_float((unsigned)u); v1 = __uint_as_float((unsigned)(u >> 32)); v2 = a2[cell];
                }
                if (gx >= 0 && gx < Wi && gy >= 0 && gy < Hi) {
                    float *G = gimg_b + gy * Wi + gx;
                    if (v0 != 0.0f) unsafeAtomicAdd(G, v0);
                    if (v1 != 0.0f) unsafeAtomicAdd(G + HWi, v1);
                    if (v2 != 0.0f) unsafeAtomicAdd(G + 2 * HWi, v2);
                }
                ly += NT / WW; lx += NT % WW;
                if (lx >= WW) { lx -= WW; ++ly; }
            }
        }
        stamp(3);      // flush atomics issued (not retired: they return nothing)
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + NT * j;
            if (i < WH * (WW / 4)) *reinterpret_cast<f4 *>(iwin + c * (WH * WW) + 4 * i) = wreg[c][j];
        }
    __syncthreads();
    stamp(4);

#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int fl = flags[k], gb = gbase[k];
        if (!(fl & LIVE)) continue;
        float out_dx = 0.0f, out_dy = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float g = go[k][c];
            float iTL, iTR, iBL, iBR;
            if (p.abl & 4) { iTL = iTR = iBL = iBR = g; }
            else if (fl & G_IN) {
                const float *wc = iwin + c * (WH * WW);
                const int ox = (fl & G_DX) ? 1 : 0, oy = (fl & G_DY) ? WW : 0;
                iTL = wc[gb]; iTR = wc[gb + ox]; iBL = wc[gb + oy]; iBR = wc[gb + oy + ox];
            } else {
                const float *I = img + (long)b * is.b + (long)c * is.c;
                const int ox = (fl & G_DX) ? (int)is.w : 0, oy = (fl & G_DY) ? (int)is.h : 0;
                iTL = I[gb]; iTR = I[gb + ox]; iBL = I[gb + oy]; iBR = I[gb + oy + ox];
            }
            if constexpr (FUSED == 2) {
                // the forward's sample (resample2d_kernel.cu:56-59 / :66-69) from the same four corners, then ChannelNorm's gradient of
                // the difference (channelnorm_kernel.cu:93) with the sign of d(diff)/d(warped)
                float val;
                if (p.bilinear) {
                    const double a = (double)alv[k], be = (double)btv[k];
                    val = 0.0f;
                    val = val + (float)(((1. - a) * (1. - be)) * (double)iTL);
                    val = val + (float)((a * (1. - be)) * (double)iTR);
                    val = val + (float)(((1. - a) * be) * (double)iBL);
                    val = val + (float)((a * be) * (double)iBR);
                } else {
                    val = alv[k] >= 0.5f ? (btv[k] >= 0.5f ? iBR : iTR) : (btv[k] >= 0.5f ? iBL : iTL);
                }
                g = 0.0f - chnorm_grad(gnv[k], g - val, nrv[k]);
            }
            out_dy = out_dy + (gam_y[k] * g) * iBL;       // (:172-177)
            out_dy = out_dy - (gam_y[k] * g) * iTL;
            out_dy = out_dy + ((1 - gam_y[k]) * g) * iBR;
            out_dy = out_dy - ((1 - gam_y[k]) * g) * iTR;
            out_dx = out_dx + (gam_x[k] * g) * iTR;       // (:185-190)
            out_dx = out_dx - (gam_x[k] * g) * iTL;
            out_dx = out_dx + ((1 - gam_x[k]) * g) * iBR;
            out_dx = out_dx - ((1 - gam_x[k]) * g) * iBL;
        }
        const int idx = tid + NT * k;
        const int x = X0 + idx % TW, y = Y0 + idx / TW;
        const long pix = (long)y * W + x;
        if (FUSED == 1) {   // + the gradient through flow / div_flow = flow * (1 / div_flow) (models.py:137)
            out_dx = out_dx + p.gcat[((long)b * 12 + 9) * HW + pix] * p.inv_div_flow;
            out_dy = out_dy + p.gcat[((long)b * 12 + 10) * HW + pix] * p.inv_div_flow;
        }
        store_out(p.gflow + (long)b * 2 * HW + pix, out_dx);
        store_out(p.gflow + (long)b * 2 * HW + HW + pix, out_dy);
    }
    if ((p.abl & 8) && p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's atomics and stores have left the CU's queue
        stamp(5);
        unsigned long long *d = p.dbg + (long)blockIdx.x * 8;
        for (int i = 0; i < 6; ++i) d[i] = ts[i];
        d[6] = (unsigned long long)xcd_remap(blockIdx.x, gridDim.x);   // the tile this workgroup took
    }
}

// Row N2's backward for shapes the tiled kernel does not take (any C, any size): one lane per pixel, corners from global memory,
// 4 atomics per (pixel, channel) onto a gradient the host initialised with the concat gradient's slice.
__global__ __launch_bounds__(256) void warp_diff_norm_cat_bwd_kernel(const float *__restrict__ pair, const float *__restrict__ flow,
                                                                     const float *__restrict__ outcat, const float *__restrict__ gcat,
                                                                     float *__restrict__ gpair, float *__restrict__ gflow,
                                                                     int C, int H, int W, long npix, float inv_div_flow, int norm_only, int bilinear)
{
    // norm_only (models.py:157-161 differentiated): outcat / gcat are the B x 1 x H x W norm and its gradient; the warped image is not
    // stored and is recomputed with the forward's arithmetic; no concat terms
    const long HW = (long)H * W;
    const int CC = 3 * C + 3;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < npix; g += (long)gridDim.x * blockDim.x) {
        const int x = (int)(g % W);
        const long row = g / W;
        const int y = (int)(row % H), b = (int)(row / H);
        const long pix = (long)y * W + x;
        const float xf = (float)x + flow[(long)b * 2 * HW + pix], yf = (float)y + flow[(long)b * 2 * HW + HW + pix];
        const float fx = floorf(xf), fy = floorf(yf);
        const int ixL = f2i_sat(fx), ixR = f2i_sat(fx + 1.0f), iyT = f2i_sat(fy), iyB = f2i_sat(fy + 1.0f);
        const int xL = clampi(ixL, 0, W - 1), xR = clampi(ixR, 0, W - 1), yT = clampi(iyT, 0, H - 1), yB = clampi(iyB, 0, H - 1);
        const float gam_y = 1 - (xf - fx), gam_x = 1 - (yf - fy);
        const float alpha = xf - (float)f2i_sat(xf), beta = yf - (float)f2i_sat(yf);
        const float s00 = (1 - alpha) * (1 - beta), s01 = alpha * (1 - beta), s10 = (1 - alpha) * beta, s11 = alpha * beta;
        const float gn = norm_only ? gcat[(long)b * HW + pix] : gcat[((long)b * CC + 3 * C + 2) * HW + pix];
        const float nrm = norm_only ? outcat[(long)b * HW + pix] : outcat[((long)b * CC + 3 * C + 2) * HW + pix];
        float out_dx = 0.0f, out_dy = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float *I = pair + ((long)b * 2 * C + C + c) * HW;
            float warped;
            if (!norm_only) warped = outcat[((long)b * CC + 2 * C + c) * HW + pix];
            else if (bilinear) {
                const double a = (double)(xf - fx), be = (double)(yf - fy);
                warped = 0.0f;
                warped = warped + (float)(((1. - a) * (1. - be)) * (double)I[yT * W + xL]);
                warped = warped + (float)((a * (1. - be)) * (double)I[yT * W + xR]);
                warped = warped + (float)(((1. - a) * be) * (double)I[yB * W + xL]);
                warped = warped + (float)((a * be) * (double)I[yB * W + xR]);
            } else {
                const int xN = clampi(d2i_sat(floor((double)xf + 0.5)), 0, W - 1), yN = clampi(d2i_sat(floor((double)yf + 0.5)), 0, H - 1);
                warped = I[yN * W + xN];
            }
            const float diff = pair[((long)b * 2 * C + c) * HW + pix] - warped;
            const float gd = chnorm_grad(gn, diff, nrm);
            const float go = norm_only ? 0.0f - gd : gcat[((long)b * CC + 2 * C + c) * HW + pix] - gd;
            if (gpair) {
                gpair[((long)b * 2 * C + c) * HW + pix] = gcat[((long)b * CC + c) * HW + pix] + gd;
                float *G = gpair + ((long)b * 2 * C + C + c) * HW;
                unsafeAtomicAdd(G + yT * W + xL, s00 * go);
                unsafeAtomicAdd(G + yT * W + xR, s01 * go);
                unsafeAtomicAdd(G + yB * W + xL, s10 * go);
                unsafeAtomicAdd(G + yB * W + xR, s11 * go);
            }
            const float iTL = I[yT * W + xL], iTR = I[yT * W + xR], iBL = I[yB * W + xL], iBR = I[yB * W + xR];
            out_dy = out_dy + (gam_y * go) * iBL;
            out_dy = out_dy - (gam_y * go) * iTL;
            out_dy = out_dy + ((1 - gam_y) * go) * iBR;
            out_dy = out_dy - ((1 - gam_y) * go) * iTR;
            out_dx = out_dx + (gam_x * go) * iTR;
            out_dx = out_dx - (gam_x * go) * iTL;
            out_dx = out_dx + ((1 - gam_x) * go) * iBR;
            out_dx = out_dx - ((1 - gam_x) * go) * iBL;
        }
        if (norm_only) { gflow[(long)b * 2 * HW + pix] = out_dx; gflow[(long)b * 2 * HW + HW + pix] = out_dy; continue; }
        gflow[(long)b * 2 * HW + pix] = out_dx + gcat[((long)b * CC + 3 * C) * HW + pix] * inv_div_flow;
        gflow[(long)b * 2 * HW + HW + pix] = out_dy + gcat[((long)b * CC + 3 * C + 1) * HW + pix] * inv_div_flow;
    }
}

// N2 for shapes the tiled kernel does not take: one lane per pixel, corners gathered from global memory.
__global__ __launch_bounds__(256) void warp_diff_norm_cat_kernel(const float *__restrict__ pair, const float *__restrict__ flow,
                                                                 float *__restrict__ out, int C, int H, int W, long npix,
                                                                 int bilinear, float div_flow, int norm_only)
{
    const long HW = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < npix; g += (long)gridDim.x * blockDim.x) {
        const int x = (int)(g % W);
        const long row = g / W;
        const int y = (int)(row % H), b = (int)(row / H);
        const long pix = (long)y * W + x;
        const float dx = flow[(long)b * 2 * HW + pix], dy = flow[(long)b * 2 * HW + HW + pix];
        const float xf = (float)x + dx, yf = (float)y + dy;
        int xL, xR, yT, yB;
        double a = 0., be = 0.;
        if (bilinear) {
            const float fx = floorf(xf), fy = floorf(yf);
            a = (double)(xf - fx); be = (double)(yf - fy);
            xL = clampi(f2i_sat(fx), 0, W - 1); xR = clampi(f2i_sat(fx + 1.0f), 0, W - 1);
            yT = clampi(f2i_sat(fy), 0, H - 1); yB = clampi(f2i_sat(fy + 1.0f), 0, H - 1);
        } else {
            xL = xR = clampi(d2i_sat(floor((double)xf + 0.5)), 0, W - 1);
            yT = yB = clampi(d2i_sat(floor((double)yf + 0.5)), 0, H - 1);
        }
        float *ob = out + (long)b * (3 * C + 3) * HW + pix;
        float ssq = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float *I = pair + ((long)b * 2 * C + C + c) * HW;
            float val;
            if (bilinear) {
                val = 0.0f;
                val = val + (float)(((1. - a) * (1. - be)) * (double)I[(long)yT * W + xL]);
                val = val + (float)((a * (1. - be)) * (double)I[(long)yT * W + xR]);
                val = val + (float)(((1. - a) * be) * (double)I[(long)yB * W + xL]);
                val = val + (float)((a * be) * (double)I[(long)yB * W + xR]);
            } else val = I[(long)yT * W + xL];
            const float v0 = pair[((long)b * 2 * C + c) * HW + pix];
            if (!norm_only) {
                ob[(long)c * HW] = v0;
                ob[(long)(C + c) * HW] = I[pix];
                ob[(long)(2 * C + c) * HW] = val;
            }
            const float d = v0 - val;
            ssq = ssq + d * d;
        }
        if (norm_only) { out[(long)b * HW + pix] = __fsqrt_rn(ssq); continue; }
        const float inv_div = 1.0f / div_flow;   // as PyTorch divides a GPU tensor by a scalar
        ob[(long)(3 * C) * HW] = dx * inv_div;
        ob[(long)(3 * C + 1) * HW] = dy * inv_div;
        ob[(long)(3 * C + 2) * HW] = __fsqrt_rn(ssq);
    }
}

// Tile height of the tiled kernels (two 1024-thread workgroups per CU = 512 resident tiles): the height that needs fewer
// window rows over all rounds; 48 turns FlowNet2's 8 x 384 x 512 into exactly one round.
static inline int tile_height(int B, int H, int tiles_x)
{
    const long slots = 512;
    const long t32 = (long)B * tiles_x * ((H + 31) / 32), t48 = (long)B * tiles_x * ((H + 47) / 48);
    const long c32 = ((t32 + slots - 1) / slots) * (32 + 32), c48 = ((t48 + slots - 1) / slots) * (48 + 32);
    return c48 < c32 ? 48 : 32;
}

static inline unsigned stream_grid(long nthreads)
{
    long blocks = (nthreads + 255) / 256;
    const long cap = 256L * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

} // namespace fn2

// argument block of resample_bwd_c3x for 32 x 64 tiles (the fields of the fused form are set by its caller)
static fn2::C3xArgs c3x_args(const float *img, fn2::ImgStrides is, const float *flow, float *gimg, long gimg_bs, float *gflow,
                             int B, int Hi, int Wi, int H, int W, int tiles_x, int abl)
{
    fn2::C3xArgs a;
    a.img = img; a.is = is; a.flow = flow; a.gout = nullptr; a.gcat = a.pair = a.outcat = nullptr; a.gpair0 = nullptr;
    a.gimg = gimg; a.gimg_bs = gimg_bs; a.gflow = gflow;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.H = H; a.W = W; a.tiles_x = tiles_x; a.tiles_y = (H + 31) / 32; a.abl = abl;
    a.inv_div_flow = 0.0f; a.bilinear = 1;
#ifdef FN2_DEBUG_BUILD
    a.dbg = static_cast<unsigned long long *>(fn2::corr_f16x2_get_debug_buffer());
#else
    a.dbg = nullptr;
#endif
    return a;
}
static unsigned c3x_grid(const fn2::C3xArgs &a) { return (unsigned)((long)a.B * a.tiles_x * a.tiles_y); }

// `bilinear`: bit 0 = bilinear (else nearest); bits 8.. = profiling switches that only fn2_debug_resample2d_* set
// (bit 8: untiled kernels, bits 9-11: backward ablations, bits 12-13: tile height)
static int resample2d_forward_impl(const float *img, const int64_t *img_strides, const float *flow, float *out,
                                   int B, int C, int Hi, int Wi, int H, int W,
                                   int kernel_size, int bilinear, void *stream)
{
    using namespace fn2;
    if (B < 0 || C < 0 || Hi < 1 || Wi < 1 || H < 0 || W < 0) return FN2_EINVAL;
    if (kernel_size < 1) return FN2_EINVAL;
    if ((long)B * C * H * W == 0) return FN2_OK;
    if (!img || !flow || !out) return FN2_EINVAL;
    if (!aligned(img, 4) || !aligned(flow, 4) || !aligned(out, 4)) return FN2_EALIGN;
    ImgStrides is;
    if (img_strides) { is.b = img_strides[0]; is.c = img_strides[1]; is.h = img_strides[2]; is.w = img_strides[3]; }
    else { is.b = (long)C * Hi * Wi; is.c = (long)Hi * Wi; is.h = Wi; is.w = 1; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long npix = (long)B * H * W;
    if (kernel_size != 1) {
        hipLaunchKernelGGL(resample_fwd_ks_kernel, dim3(stream_grid(npix)), dim3(256), 0, s, img, is, flow, out, C, Hi, Wi, H, W,
                           npix, kernel_size, (bilinear & 1) ? 1 : 0);
        return launch_status();
    }
    // tiled path: image rows contiguous and 16 B aligned, same size as the flow, large enough to tile
    const bool tiled_ok = (is.w == 1) && (is.h % 4 == 0) && (is.c % 4 == 0) && (is.b % 4 == 0) && aligned(img, 16) &&
                          (Hi == H) && (Wi == W) && (W % 4 == 0) && (H >= 16) && (W >= 32) && !(bilinear & 0x100);
    if (tiled_ok) {
        constexpr int TW = 64;
        const int tiles_x = (W + TW - 1) / TW;
#define FN2_RF(TH)                                                                                                     \
    do {                                                                                                               \
        const int tiles_y = (H + TH - 1) / TH;                                                                         \
        hipLaunchKernelGGL((resample_fwd_tiled<TH, TW, 16>), dim3((unsigned)((long)B * tiles_x * tiles_y)), dim3(1024), \
                           0, s, img, is, flow, out, C, Hi, Wi, H, W, tiles_x, tiles_y, (bilinear & 1) ? 1 : 0);       \
    } while (0)
        if (((bilinear >> 14) & 3) == 1 && aligned(flow, 16) && aligned(out, 16)) {   // profiling: 4 adjacent pixels per thread, 48 x 64 tiles
            const int tiles_y = (H + 47) / 48;
            hipLaunchKernelGGL((resample_fwd_tiled4<48, 16>), dim3((unsigned)((long)B * tiles_x * tiles_y)), dim3(768), 0, s,
                               img, is, flow, out, C, Hi, Wi, H, W, tiles_x, tiles_y, (bilinear & 1) ? 1 : 0);
            return launch_status();
        }
        if (((bilinear >> 14) & 3) == 2 && aligned(flow, 16) && aligned(out, 16)) {   // ... 64 x 64 tiles, 1024 threads
            const int tiles_y = (H + 63) / 64;
            hipLaunchKernelGGL((resample_fwd_tiled4<64, 16>), dim3((unsigned)((long)B * tiles_x * tiles_y)), dim3(1024), 0, s,
                               img, is, flow, out, C, Hi, Wi, H, W, tiles_x, tiles_y, (bilinear & 1) ? 1 : 0);
            return launch_status();
        }
        if (((bilinear >> 12) & 3) == 3 && C == 3) {   // profiling: every channel window resident, 96 x 64 tiles, one workgroup per CU
            const int tiles_y = (H + 95) / 96;
            hipLaunchKernelGGL((resample_fwd_tiled_all<96, TW, 16, 3>), dim3((unsigned)((long)B * tiles_x * tiles_y)), dim3(1024), 0, s,
                               img, is, flow, out, Hi, Wi, H, W, tiles_x, tiles_y, (bilinear & 1) ? 1 : 0);
            return launch_status();
        }
        if (((bilinear >> 12) & 15) == 0 && C == 3) {
            // C == 3 (FlowNet2's only use): the three channel windows resident at once in 32 x 64 tiles (73.7 KB: two workgroups per
            // CU), one barrier in the whole kernel, corner offsets and weights formed once per pixel -- 8 x 3 x 384 x 512: 19.3 us
            // against 20.9 for the per-channel kernel (smooth flow 18.1 / 19.1), same bits
            const int tiles_y = (H + 31) / 32;
            unsigned long long *dbgbuf = nullptr;
#ifdef FN2_DEBUG_BUILD
            if (bilinear & 0x10000) dbgbuf = static_cast<unsigned long long *>(corr_f16x2_get_debug_buffer());   // profiling: timeline stamps
#endif
            // (512 persistent workgroups taking one tile and then half of one of the remaining 256 -- one balanced round instead of one
            // and a half -- measured slower: 18.6 us against 17.0, the half tile costs 4.8 us of fixed latencies)
            hipLaunchKernelGGL((resample_fwd_tiled_all<32, TW, 16, 3, 8>), dim3((unsigned)((long)B * tiles_x * tiles_y)), dim3(1024), 0, s,
                               img, is, flow, out, Hi, Wi, H, W, tiles_x, tiles_y, (bilinear & 1) ? 1 : 0, dbgbuf);
            return launch_status();
        }
        const int th = (bilinear >> 12) & 3 ? ((bilinear >> 12) & 3) == 1 ? 48 : 32 : tile_height(B, H, tiles_x);
        if (th == 48) FN2_RF(48); else FN2_RF(32);
#undef FN2_RF
        return launch_status();
    }
    if (W % 4 == 0 && aligned(flow, 16) && aligned(out, 16)) {
        const long ng = npix / 4;
        hipLaunchKernelGGL(resample_fwd_kernel<4>, dim3(stream_grid(ng)), dim3(256), 0, s, img, is, flow, out, C, Hi, Wi,
                           H, W, ng, (bilinear & 1) ? 1 : 0);
    } else {
        hipLaunchKernelGGL(resample_fwd_kernel<1>, dim3(stream_grid(npix)), dim3(256), 0, s, img, is, flow, out, C, Hi,
                           Wi, H, W, npix, (bilinear & 1) ? 1 : 0);
    }
    return launch_status();
}

extern "C" int fn2_resample2d_forward(const float *img, const int64_t *img_strides, const float *flow, float *out,
                                      int B, int C, int Hi, int Wi, int H, int W,
                                      int kernel_size, int bilinear, void *stream)
{
    return resample2d_forward_impl(img, img_strides, flow, out, B, C, Hi, Wi, H, W, kernel_size, bilinear != 0 ? 1 : 0, stream);
}

#ifdef FN2_DEBUG_BUILD
extern "C" int fn2_debug_resample2d_forward(const float *img, const int64_t *img_strides, const float *flow, float *out,
                                            int B, int C, int Hi, int Wi, int H, int W,
                                            int kernel_size, int bilinear, int flags, void *stream)
{
    return resample2d_forward_impl(img, img_strides, flow, out, B, C, Hi, Wi, H, W, kernel_size,
                                   (bilinear != 0 ? 1 : 0) | (flags & ~0xff), stream);
}
#endif

static int resample2d_backward_impl(const float *img, const int64_t *img_strides, const float *flow,
                                    const float *grad_out, float *grad_img, float *grad_flow,
                                    int B, int C, int Hi, int Wi, int H, int W,
                                    int kernel_size, int bilinear, void *stream)
{
    using namespace fn2;
    // both reference backward kernels ignore the bilinear flag; bit 8 of it selects the
    // untiled scatter kernel (profiling / A-B only)
    if (B < 0 || C < 0 || Hi < 1 || Wi < 1 || H < 0 || W < 0) return FN2_EINVAL;
    if (kernel_size < 1) return FN2_EINVAL;
    if ((long)B * H * W == 0) return FN2_OK;
    if (!img || !flow || !grad_out || !grad_img || !grad_flow) return FN2_EINVAL;
    if (!aligned(img, 4) || !aligned(flow, 4) || !aligned(grad_out, 4) || !aligned(grad_img, 4) || !aligned(grad_flow, 4))
        return FN2_EALIGN;
    ImgStrides is;
    if (img_strides) { is.b = img_strides[0]; is.c = img_strides[1]; is.h = img_strides[2]; is.w = img_strides[3]; }
    else { is.b = (long)C * Hi * Wi; is.c = (long)Hi * Wi; is.h = Wi; is.w = 1; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long npix = (long)B * H * W;
    if (kernel_size != 1) {
        hipLaunchKernelGGL(resample_bwd_ks_kernel, dim3(stream_grid(npix)), dim3(256), 0, s, img, is, flow, grad_out, grad_img,
                           grad_flow, C, Hi, Wi, H, W, npix, kernel_size);
        return launch_status();
    }
    const bool tiled_ok = (is.w == 1) && (is.h % 4 == 0) && (is.c % 4 == 0) && (is.b % 4 == 0) && aligned(img, 16) &&
                          (Hi == H) && (Wi == W) && (W % 4 == 0) && (H >= 16) && (W >= 32) && !(bilinear & 0x100);
    if (tiled_ok) {
        constexpr int TW = 64;
        const int abl = ((bilinear >> 9) & 7) | ((bilinear & 0x10000) ? 8 : 0);   // bits 9-11, 16 of `bilinear`: profiling switches, 0 from the bindings
        const int tiles_x = (W + TW - 1) / TW;
#define FN2_RB(TH, R, WPE, ACC)                                                                                        \
    do {                                                                                                               \
        const int tiles_y = (H + TH - 1) / TH;                                                                         \
        hipLaunchKernelGGL((resample_bwd_tiled<TH, TW, R, 1024, WPE, ACC>), dim3((unsigned)((long)B * tiles_x * tiles_y)), \
                           dim3(1024), 0, s, img, is, flow, grad_out, grad_img, grad_flow, C, Hi, Wi, H, W, tiles_x,   \
                           tiles_y, abl);                                                                              \
    } while (0)
        // bits 12-13: tile height (profiling: 1 = 48, 2 = 32, 3 = 64), 0 = automatic; bits 14-15: accumulation window
        // (profiling: 1 = fp64 cells 48 x 64 +- 12, 2 = fp64 cells 32 x 64 +- 16, 3 = fp64 cells 48 x 64 +- 16, one workgroup per CU)
        switch ((bilinear >> 12) & 15) {
        case 1: FN2_RB(48, 16, 8, 0); break;
        case 2: FN2_RB(32, 16, 8, 0); break;
        case 3: FN2_RB(64, 16, 8, 0); break;
        case 4: FN2_RB(48, 12, 8, 1); break;
        case 8: FN2_RB(32, 16, 8, 1); break;
        case 12: FN2_RB(48, 16, 4, 1); break;
        case 5: FN2_RB(48, 12, 8, 0); break;
        case 6: FN2_RB(96, 16, 4, 0); break;
        case 7: FN2_RB(96, 16, 4, 1); break;
        case 9: FN2_RB(48, 16, 4, 0); break;
        case 10: if (tile_height(B, H, tiles_x) == 48) FN2_RB(48, 16, 8, 0); else FN2_RB(32, 16, 8, 0); break;   // the round-3 choice
        // C == 3 (FlowNet2's only use): the three channels in one scatter, workgroup order alternating with i / 8 -- 8 x 3 x 384 x 512,
        // white-noise flow 51.0 us, smooth flow 37.7.  Other C: one channel at a time, 32-row tiles with fp64 cells (74 KB of LDS, two
        // workgroups per CU; selector 8): 54.2 / 38.6 us against 57.2-59.8 / 44.5 for the 48 x 64 fp32 tiles of round 3 (selector 10)
        default:
            if (C == 3) {
                C3xArgs a = c3x_args(img, is, flow, grad_img, (long)C * Hi * Wi, grad_flow, B, Hi, Wi, H, W, tiles_x, abl);
                a.gout = grad_out;
                hipLaunchKernelGGL((resample_bwd_c3x<32, TW, 16, 1024, 0, true>), dim3(c3x_grid(a)), dim3(1024), 0, s, a);
            } else FN2_RB(32, 16, 8, 1);
            break;
        }
#undef FN2_RB
    } else {
        hipLaunchKernelGGL(resample_bwd_kernel, dim3(stream_grid(npix)), dim3(256), 0, s, img, is, flow, grad_out,
                           grad_img, grad_flow, C, Hi, Wi, H, W, npix);
    }
    return launch_status();
}

extern "C" int fn2_resample2d_backward(const float *img, const int64_t *img_strides, const float *flow,
                                       const float *grad_out, float *grad_img, float *grad_flow,
                                       int B, int C, int Hi, int Wi, int H, int W,
                                       int kernel_size, int bilinear, void *stream)
{
    return resample2d_backward_impl(img, img_strides, flow, grad_out, grad_img, grad_flow, B, C, Hi, Wi, H, W, kernel_size,
                                    bilinear != 0 ? 1 : 0, stream);
}

#ifdef FN2_DEBUG_BUILD
extern "C" int fn2_debug_resample2d_backward(const float *img, const int64_t *img_strides, const float *flow,
                                             const float *grad_out, float *grad_img, float *grad_flow,
                                             int B, int C, int Hi, int Wi, int H, int W,
                                             int kernel_size, int bilinear, int flags, void *stream)
{
    return resample2d_backward_impl(img, img_strides, flow, grad_out, grad_img, grad_flow, B, C, Hi, Wi, H, W, kernel_size,
                                    (bilinear != 0 ? 1 : 0) | (flags & ~0xff), stream);
}
#endif

extern "C" int fn2_warp_diff_norm_cat(const float *pair, const float *flow, float *out, float div_flow,
                                      int B, int C, int H, int W, int bilinear, void *stream)
{
    using namespace fn2;
    bilinear = bilinear != 0 ? 1 : 0;
    if (B < 0 || C < 1 || H < 1 || W < 1 || !(div_flow == div_flow) || div_flow == 0.0f) return FN2_EINVAL;
    if ((long)B * H * W == 0) return FN2_OK;
    if (!pair || !flow || !out) return FN2_EINVAL;
    if (!aligned(pair, 4) || !aligned(flow, 4) || !aligned(out, 4)) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W, npix = (long)B * HW;
    const bool tiled_ok = (W % 4 == 0) && aligned(pair, 16) && (H >= 16) && (W >= 32) && !(bilinear & 0x100);
    if (tiled_ok) {
        ImgStrides is;
        is.b = 2 * C * HW; is.c = HW; is.h = W; is.w = 1;
        const float *img1 = pair + (long)C * HW;
        constexpr int TW = 64;
        const int tiles_x = (W + TW - 1) / TW;
#define FN2_WF(TH)                                                                                                      \
    do {                                                                                                                \
        const int tiles_y = (H + TH - 1) / TH;                                                                          \
        hipLaunchKernelGGL((resample_fwd_tiled<TH, TW, 16, 1>), dim3((unsigned)((long)B * tiles_x * tiles_y)),          \
                           dim3(1024), 0, s, img1, is, flow, out, C, H, W, H, W, tiles_x, tiles_y, (bilinear & 1) ? 1 : 0, \
                           pair, div_flow);                                                                             \
    } while (0)
        if (tile_height(B, H, tiles_x) == 48) FN2_WF(48); else FN2_WF(32);
#undef FN2_WF
    } else {
        hipLaunchKernelGGL(warp_diff_norm_cat_kernel, dim3(stream_grid(npix)), dim3(256), 0, s, pair, flow, out, C, H, W,
                           npix, (bilinear & 1) ? 1 : 0, div_flow, 0);
    }
    return launch_status();
}

// shapes resample_bwd_c3x takes (the tiled path's conditions, three channels)
static bool c3x_ok(fn2::ImgStrides is, const float *img, int C, int Hi, int Wi, int H, int W)
{
    return C == 3 && (is.w == 1) && (is.h % 4 == 0) && (is.c % 4 == 0) && (is.b % 4 == 0) && fn2::aligned(img, 16) && (Hi == H) &&
           (Wi == W) && (W % 4 == 0) && (H >= 16) && (W >= 32);
}

extern "C" int fn2_warp_diff_norm_cat_backward(const float *pair, const float *flow, const float *out_cat, const float *grad_cat,
                                               float *grad_pair, float *grad_flow, float div_flow, int B, int C, int H, int W,
                                               int bilinear, void *stream)
{
    using namespace fn2;
    (void)bilinear;   // both reference backward kernels ignore the flag (resample2d_kernel.cu:75-198)
    if (B < 0 || C < 1 || H < 1 || W < 1 || !(div_flow == div_flow) || div_flow == 0.0f) return FN2_EINVAL;
    if ((long)B * H * W == 0) return FN2_OK;
    if (!pair || !flow || !out_cat || !grad_cat || !grad_flow) return FN2_EINVAL;
    if (!aligned(pair, 4) || !aligned(flow, 4) || !aligned(out_cat, 4) || !aligned(grad_cat, 4) || !aligned(grad_flow, 4) ||
        (grad_pair && !aligned(grad_pair, 4)))
        return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W, npix = (long)B * HW;
    const float inv = 1.0f / div_flow;
    if (grad_pair) {   // the second image's gradient starts as its slice of the concat gradient (one strided copy), then the scatter adds
        hipError_t e = hipMemcpy2DAsync(grad_pair + (long)C * HW, 2 * C * HW * sizeof(float), grad_cat + (long)C * HW,
                                        (3 * C + 3) * HW * sizeof(float), C * HW * sizeof(float), B, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    ImgStrides is;
    is.b = 2L * C * HW; is.c = HW; is.h = W; is.w = 1;
    const float *img1 = pair + (long)C * HW;
    if (c3x_ok(is, img1, C, H, W, H, W)) {
        C3xArgs a = c3x_args(img1, is, flow, grad_pair ? grad_pair + 3 * HW : nullptr, 6 * HW, grad_flow, B, H, W, H, W, (W + 63) / 64, 0);
        a.gcat = grad_cat; a.pair = pair; a.outcat = out_cat; a.gpair0 = grad_pair; a.inv_div_flow = inv;
        if (grad_pair) hipLaunchKernelGGL((resample_bwd_c3x<32, 64, 16, 1024, 1, true>), dim3(c3x_grid(a)), dim3(1024), 0, s, a);
        else hipLaunchKernelGGL((resample_bwd_c3x<32, 64, 16, 1024, 1, false>), dim3(c3x_grid(a)), dim3(1024), 0, s, a);
        return launch_status();
    }
    hipLaunchKernelGGL(warp_diff_norm_cat_bwd_kernel, dim3(stream_grid(npix)), dim3(256), 0, s, pair, flow, out_cat, grad_cat, grad_pair,
                       grad_flow, C, H, W, npix, inv, 0, 1);
    return launch_status();
}

// models.py:157-161 / :170-174: ||pair[:, :C] - Resample2d(pair[:, C:], flow)||_2 with no concat -- row N2's forward kernel storing
// only the norm plane
extern "C" int fn2_warp_diff_norm(const float *pair, const float *flow, float *out_norm, int B, int C, int H, int W, int bilinear, void *stream)
{
    using namespace fn2;
    bilinear = bilinear != 0 ? 1 : 0;
    if (B < 0 || C < 1 || H < 1 || W < 1) return FN2_EINVAL;
    if ((long)B * H * W == 0) return FN2_OK;
    if (!pair || !flow || !out_norm) return FN2_EINVAL;
    if (!aligned(pair, 4) || !aligned(flow, 4) || !aligned(out_norm, 4)) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W, npix = (long)B * HW;
    if ((W % 4 == 0) && aligned(pair, 16) && (H >= 16) && (W >= 32)) {
        ImgStrides is;
        is.b = 2 * C * HW; is.c = HW; is.h = W; is.w = 1;
        const float *img1 = pair + (long)C * HW;
        const int tiles_x = (W + 63) / 64;
#define FN2_WN(TH)                                                                                                      \
    do {                                                                                                                \
        const int tiles_y = (H + TH - 1) / TH;                                                                          \
        hipLaunchKernelGGL((resample_fwd_tiled<TH, 64, 16, 2>), dim3((unsigned)((long)B * tiles_x * tiles_y)),          \
                           dim3(1024), 0, s, img1, is, flow, out_norm, C, H, W, H, W, tiles_x, tiles_y, bilinear, pair, 1.0f); \
    } while (0)
        if (tile_height(B, H, tiles_x) == 48) FN2_WN(48); else FN2_WN(32);
#undef FN2_WN
    } else {
        hipLaunchKernelGGL(warp_diff_norm_cat_kernel, dim3(stream_grid(npix)), dim3(256), 0, s, pair, flow, out_norm, C, H, W, npix,
                           bilinear, 1.0f, 1);
    }
    return launch_status();
}

// ... and its backward with respect to the flow (the pair is the network's input in FlowNet2; compose the unfused entry points when it
// needs a gradient): grad_flow = Resample2d's flow gradient for g_warped = -grad_norm * diff / (norm + 1e-9), the warp recomputed
extern "C" int fn2_warp_diff_norm_backward(const float *pair, const float *flow, const float *norm, const float *grad_norm,
                                           float *grad_flow, int B, int C, int H, int W, int bilinear, void *stream)
{
    using namespace fn2;
    bilinear = bilinear != 0 ? 1 : 0;
    if (B < 0 || C < 1 || H < 1 || W < 1) return FN2_EINVAL;
    if ((long)B * H * W == 0) return FN2_OK;
    if (!pair || !flow || !norm || !grad_norm || !grad_flow) return FN2_EINVAL;
    if (!aligned(pair, 4) || !aligned(flow, 4) || !aligned(norm, 4) || !aligned(grad_norm, 4) || !aligned(grad_flow, 4)) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W, npix = (long)B * HW;
    ImgStrides is;
    is.b = 2L * C * HW; is.c = HW; is.h = W; is.w = 1;
    const float *img1 = pair + (long)C * HW;
    if (c3x_ok(is, img1, C, H, W, H, W)) {
        C3xArgs a = c3x_args(img1, is, flow, nullptr, 0, grad_flow, B, H, W, H, W, (W + 63) / 64, 0);
        a.gcat = grad_norm; a.pair = pair; a.outcat = norm; a.bilinear = bilinear;
        hipLaunchKernelGGL((resample_bwd_c3x<32, 64, 16, 1024, 2, false>), dim3(c3x_grid(a)), dim3(1024), 0, s, a);
        return launch_status();
    }
    hipLaunchKernelGGL(warp_diff_norm_cat_bwd_kernel, dim3(stream_grid(npix)), dim3(256), 0, s, pair, flow, norm, grad_norm,
                       static_cast<float *>(nullptr), grad_flow, C, H, W, npix, 0.0f, 1, bilinear);
    return launch_status();
}
