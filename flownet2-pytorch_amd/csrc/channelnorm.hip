// channelnorm.hip -- ChannelNorm forward/backward for gfx950.
//
// Replaces reference kernels kernel_channelnorm_update_output (channelnorm_kernel.cu:18-60)
// and kernel_channelnorm_backward_input1 (:63-96).  Both are pure HBM streaming:
//   fwd  reads C planes, writes 1 plane   -> (C+1) * B*H*W * sizeof(T) bytes
//   bwd  reads C + 2 planes, writes C     -> (2C+2) * B*H*W * sizeof(T) bytes
// Layout: NCHW; one lane owns VEC consecutive pixels of one image and walks the channel
// planes (plane stride H*W), so every wave access is a fully coalesced 16 B/lane segment.
// Arithmetic follows the reference exactly: square in T, accumulate and sqrt in float (fwd);
// float product divided in double by (float(out) + 1e-9) (bwd).
#include "fn2_common.h"

namespace fn2 {

template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef float __attribute__((ext_vector_type(4))) type; static constexpr int N = 4; };
template <> struct VecOf<half_t> { typedef half_t __attribute__((ext_vector_type(8))) type; static constexpr int N = 8; };
template <> struct VecOf<double> { typedef double __attribute__((ext_vector_type(2))) type; static constexpr int N = 2; };

// ---------------------------------------------------------------- forward
// grid-stride over "pixel groups" of N pixels; HW % N == 0 guaranteed by the launcher.
template <typename T>
__global__ __launch_bounds__(256) void chnorm_fwd_vec(const T *__restrict__ in, T *__restrict__ out,
                                                      int C, long HW, long ngroups)
{
    typedef typename VecOf<T>::type V;
    constexpr int N = VecOf<T>::N;
    const long gpp = HW / N; // groups per plane
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < ngroups; g += (long)gridDim.x * blockDim.x) {
        const long b = g / gpp, p = (g - b * gpp) * N;
        const T *src = in + b * C * HW + p;
        float acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.0f;
        for (int c = 0; c < C; ++c) {
            const V v = *reinterpret_cast<const V *>(src + (long)c * HW);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const T sq = v[i] * v[i];     // square in T (:56)
                acc[i] = acc[i] + (float)sq;  // float accumulation (:51,:56)
            }
        }
        V o;
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = (T)__fsqrt_rn(acc[i]);
        store_out(reinterpret_cast<V *>(out + b * HW + p), o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void chnorm_fwd_scalar(const T *__restrict__ in, T *__restrict__ out,
                                                         int C, long HW, long npix)
{
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < npix; g += (long)gridDim.x * blockDim.x) {
        const long b = g / HW, p = g - b * HW;
        const T *src = in + b * C * HW + p;
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
            const T v = src[(long)c * HW];
            const T sq = v * v;
            acc = acc + (float)sq;
        }
        out[g] = (T)__fsqrt_rn(acc);
    }
}

// ---------------------------------------------------------------- backward
// chnorm_grad(go, x, o): fn2_common.h (shared with the fused warp backward of resample2d.hip)

// contiguous-pixel fast path: gout pixel stride 1 and row stride W (a channel slice of a
// contiguous NCHW tensor qualifies: only its batch stride differs).
template <typename T>
__global__ __launch_bounds__(256) void chnorm_bwd_vec(const T *__restrict__ in, const T *__restrict__ out,
                                                      const T *__restrict__ gout, long gs_b,
                                                      T *__restrict__ gin, int C, long HW, long ngroups)
{
    typedef typename VecOf<T>::type V;
    constexpr int N = VecOf<T>::N;
    const long gpp = HW / N;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < ngroups; g += (long)gridDim.x * blockDim.x) {
        const long b = g / gpp, p = (g - b * gpp) * N;
        const V o = *reinterpret_cast<const V *>(out + b * HW + p);
        const V go = *reinterpret_cast<const V *>(gout + b * gs_b + p);
        for (int c = 0; c < C; ++c) {
            const long off = (b * C + c) * HW + p;
            const V x = *reinterpret_cast<const V *>(in + off);
            V r;
#pragma unroll
            for (int i = 0; i < N; ++i) r[i] = (T)chnorm_grad((float)go[i], (float)x[i], (float)o[i]);
            store_out(reinterpret_cast<V *>(gin + off), r);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void chnorm_bwd_scalar(const T *__restrict__ in, const T *__restrict__ out,
                                                         const T *__restrict__ gout, long gs_b, long gs_h, long gs_w,
                                                         T *__restrict__ gin, int C, int H, int W, long npix)
{
    const long HW = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < npix; g += (long)gridDim.x * blockDim.x) {
        const long b = g / HW, p = g - b * HW;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        const float o = (float)out[g];
        const float go = (float)gout[b * gs_b + y * gs_h + x * gs_w];
        for (int c = 0; c < C; ++c) {
            const long off = (b * C + c) * HW + p;
            gin[off] = (T)chnorm_grad(go, (float)in[off], o);
        }
    }
}

static inline unsigned stream_grid(long nthreads)
{
    long blocks = (nthreads + 255) / 256;
    const long cap = 256L * 8; // 8 blocks per CU, grid-stride the rest
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

template <typename T>
static int chnorm_fwd_launch(const void *in_, void *out_, int B, int C, int H, int W, hipStream_t s)
{
    const T *in = static_cast<const T *>(in_);
    T *out = static_cast<T *>(out_);
    const long HW = (long)H * W, npix = (long)B * HW;
    constexpr int N = VecOf<T>::N;
    if (HW % N == 0 && aligned(in, sizeof(T) * N) && aligned(out, sizeof(T) * N)) {
        const long ng = npix / N;
        hipLaunchKernelGGL(chnorm_fwd_vec<T>, dim3(stream_grid(ng)), dim3(256), 0, s, in, out, C, HW, ng);
    } else {
        hipLaunchKernelGGL(chnorm_fwd_scalar<T>, dim3(stream_grid(npix)), dim3(256), 0, s, in, out, C, HW, npix);
    }
    return launch_status();
}

template <typename T>
static int chnorm_bwd_launch(const void *in_, const void *out_, const void *gout_, const int64_t *gs,
                             void *gin_, int B, int C, int H, int W, hipStream_t s)
{
    const T *in = static_cast<const T *>(in_);
    const T *out = static_cast<const T *>(out_);
    const T *gout = static_cast<const T *>(gout_);
    T *gin = static_cast<T *>(gin_);
    const long HW = (long)H * W, npix = (long)B * HW;
    const long gs_b = gs ? (long)gs[0] : HW, gs_h = gs ? (long)gs[2] : W, gs_w = gs ? (long)gs[3] : 1;
    constexpr int N = VecOf<T>::N;
    const bool dense_pix = (gs_w == 1 || W == 1) && (gs_h == W || H == 1);
    if (dense_pix && HW % N == 0 && gs_b % N == 0 && aligned(in, sizeof(T) * N) && aligned(out, sizeof(T) * N) &&
        aligned(gout, sizeof(T) * N) && aligned(gin, sizeof(T) * N)) {
        const long ng = npix / N;
        hipLaunchKernelGGL(chnorm_bwd_vec<T>, dim3(stream_grid(ng)), dim3(256), 0, s, in, out, gout, gs_b, gin, C, HW, ng);
    } else {
        hipLaunchKernelGGL(chnorm_bwd_scalar<T>, dim3(stream_grid(npix)), dim3(256), 0, s, in, out, gout, gs_b, gs_h,
                           gs_w, gin, C, H, W, npix);
    }
    return launch_status();
}

} // namespace fn2

extern "C" int fn2_channelnorm_forward(const void *in, void *out, int dtype, int B, int C, int H, int W, void *stream)
{
    using namespace fn2;
    if (B < 0 || C < 1 || H < 0 || W < 0) return FN2_EINVAL;
    if (!dtype_size(dtype)) return FN2_EDTYPE;
    if ((long)B * H * W == 0) return FN2_OK;
    if (!in || !out) return FN2_EINVAL;
    if (!aligned(in, dtype_size(dtype)) || !aligned(out, dtype_size(dtype))) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case FN2_F32: return chnorm_fwd_launch<float>(in, out, B, C, H, W, s);
    case FN2_F16: return chnorm_fwd_launch<half_t>(in, out, B, C, H, W, s);
    default: return chnorm_fwd_launch<double>(in, out, B, C, H, W, s);
    }
}

extern "C" int fn2_channelnorm_backward(const void *in, const void *out, const void *grad_out,
                                        const int64_t *gout_strides, void *grad_in, int dtype,
                                        int B, int C, int H, int W, void *stream)
{
    using namespace fn2;
    if (B < 0 || C < 1 || H < 0 || W < 0) return FN2_EINVAL;
    if (!dtype_size(dtype)) return FN2_EDTYPE;
    if ((long)B * H * W == 0) return FN2_OK;
    if (!in || !out || !grad_out || !grad_in) return FN2_EINVAL;
    const size_t es = dtype_size(dtype);
    if (!aligned(in, es) || !aligned(out, es) || !aligned(grad_out, es) || !aligned(grad_in, es)) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case FN2_F32: return chnorm_bwd_launch<float>(in, out, grad_out, gout_strides, grad_in, B, C, H, W, s);
    case FN2_F16: return chnorm_bwd_launch<half_t>(in, out, grad_out, gout_strides, grad_in, B, C, H, W, s);
    default: return chnorm_bwd_launch<double>(in, out, grad_out, gout_strides, grad_in, B, C, H, W, s);
    }
}
