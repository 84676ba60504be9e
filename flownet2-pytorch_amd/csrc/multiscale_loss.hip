// multiscale_loss.hip -- SURVEY.md 8f N3: FlowNet2's training loss (losses.py:52-86, MultiScale with the L1 norm) and
// the EPE metric next to it (losses.py:11-12) in one pass over the target flow.
//
// The reference evaluates, for the five predictions out_i (B x 2 x H/k_i x W/k_i, k_i = 4 << i):
//     t   = div_flow * target                                   (losses.py:74)
//     t_i = AvgPool2d(k_i, k_i)(t)                              (:69,:76)  -- five passes over the full-size target
//     L1_i  = mean |out_i - t_i|                                (:17, :78) over B*2*H_i*W_i elements
//     EPE_i = mean_{b,y,x} || t_i - out_i ||_2 over channels    (:12, :77)
//     loss = sum_i w_i L1_i,  epe = sum_i w_i EPE_i,  w_i = l_weight / 2^i     (:59)
// with ~35 small launches.  Here one workgroup owns one coarsest cell (k_max x k_max pixels of one batch item), reads it
// once, forms the finest pooled sums and derives every coarser level from the level below (sums of 2x2 sums: same value
// as the reference's k x k sum up to fp32 summation order), compares with the predictions, optionally writes
// d(loss)/d(out_i) = grad_scale * w_i / N_i * sign(out_i - t_i), and leaves per-workgroup partial sums that a second tiny
// kernel adds up in a fixed order (deterministic).
// norm = 'L2' (losses.py:64-67: the loss of each scale is L2() = mean over pixels of the channel 2-norm, :21-26 -- the same
// expression as EPE) differs only in the gradient: grad_scale * w_i / (N_i / 2) * (out_i - t_i) / ||out_i - t_i||_2, zero
// where the norm is zero (torch.norm's backward).
#include "fn2_common.h"

namespace fn2 {

constexpr int MS_MAX_SCALES = 6;
struct MsArgs {
    const float *out[MS_MAX_SCALES];
    float *grad[MS_MAX_SCALES];
    float gw[MS_MAX_SCALES];       // grad_scale * w_i / N_i  (norm 2: grad_scale * w_i / (N_i / 2))
    const float *target;
    float *partial;                // [nblocks][2 * ns]
    int B, H, W, s0, ns, kmax, bx, by, vec4, norm;
    float div_flow;
};

__global__ __launch_bounds__(256) void multiscale_l1_epe_kernel(MsArgs p)
{
    __shared__ float lv[2][2][256];      // [ping-pong][channel][cell]
    __shared__ float red[4][2 * MS_MAX_SCALES];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int cbx = t % p.bx; t /= p.bx;
    const int cby = t % p.by;
    const int b = t / p.by;
    const int Y0 = cby * p.kmax, X0 = cbx * p.kmax;
    const long HW = (long)p.H * p.W;
    float l1[MS_MAX_SCALES], ep[MS_MAX_SCALES];
#pragma unroll
    for (int i = 0; i < MS_MAX_SCALES; ++i) l1[i] = ep[i] = 0.0f;

    int n = p.kmax / p.s0;               // cells per block edge at the current level (<= 16)
    int k = p.s0;
    // level 0: a thread sums one s0 x s0 cell of each channel (row-major), after the div_flow scaling of every element
    if (tid < n * n) {
        const int cy = tid / n, cx = tid - cy * n;
        const int y0 = Y0 + cy * k, x0 = X0 + cx * k;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float sum = 0.0f;
            if (y0 + k <= p.H && x0 + k <= p.W) {
                const float *T = p.target + ((long)b * 2 + c) * HW;
                if (k == 4 && p.vec4) {          // one 16 B load per cell row (same summation order)
                    for (int yy = 0; yy < 4; ++yy) {
                        const float4 v = *reinterpret_cast<const float4 *>(T + (long)(y0 + yy) * p.W + x0);
                        sum = sum + p.div_flow * v.x; sum = sum + p.div_flow * v.y;
                        sum = sum + p.div_flow * v.z; sum = sum + p.div_flow * v.w;
                    }
                } else {
                    for (int yy = 0; yy < k; ++yy)
                        for (int xx = 0; xx < k; ++xx) sum = sum + p.div_flow * T[(long)(y0 + yy) * p.W + x0 + xx];
                }
            }
            lv[0][c][tid] = sum;
        }
    }
    __syncthreads();
    int cur = 0;
    for (int i = 0; i < p.ns; ++i) {
        const int Hi = p.H / k, Wi = p.W / k;
        if (tid < n * n) {
            const int cy = tid / n, cx = tid - cy * n;
            const int gy = Y0 / k + cy, gx = X0 / k + cx;
            if (gy < Hi && gx < Wi) {
                const float inv = (float)(k * k);
                const long o = ((long)b * 2 * Hi + gy) * Wi + gx, plane = (long)Hi * Wi;
                const float d0 = p.out[i][o] - lv[cur][0][tid] / inv;
                const float d1 = p.out[i][o + plane] - lv[cur][1][tid] / inv;
                l1[i] = fabsf(d0) + fabsf(d1);
                ep[i] = __fsqrt_rn(d0 * d0 + d1 * d1);
                if (p.grad[i]) {
                    if (p.norm == 2) {
                        const float g = ep[i] > 0.0f ? p.gw[i] / ep[i] : 0.0f;
                        p.grad[i][o] = g * d0;
                        p.grad[i][o + plane] = g * d1;
                    } else {
                        p.grad[i][o] = d0 > 0.0f ? p.gw[i] : (d0 < 0.0f ? -p.gw[i] : 0.0f);
                        p.grad[i][o + plane] = d1 > 0.0f ? p.gw[i] : (d1 < 0.0f ? -p.gw[i] : 0.0f);
                    }
                }
            }
        }
        if (i + 1 < p.ns) {              // next level: 2 x 2 sums of this one
            const int n2 = n / 2;
            if (tid < n2 * n2) {
                const int cy = tid / n2, cx = tid - cy * n2;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float *L = lv[cur][c];
                    lv[cur ^ 1][c][tid] = (L[(2 * cy) * n + 2 * cx] + L[(2 * cy) * n + 2 * cx + 1]) +
                                          (L[(2 * cy + 1) * n + 2 * cx] + L[(2 * cy + 1) * n + 2 * cx + 1]);
                }
            }
            __syncthreads();
            cur ^= 1; n = n2; k *= 2;
        }
    }
    // workgroup reduction of the 2*ns partial sums: wave shuffles, then 4 waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = 0; i < p.ns; ++i) {
        float a = l1[i], e = ep[i];
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); e += __shfl_down(e, off, 64); }
        if (lane == 0) { red[wave][i] = a; red[wave][p.ns + i] = e; }
    }
    __syncthreads();
    if (tid < 2 * p.ns) p.partial[(long)blockIdx.x * 2 * p.ns + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(256) void multiscale_reduce_kernel(const float *partial, float *sums, int nblocks, int nv)
{
    __shared__ float red[256];
    for (int v = 0; v < nv; ++v) {
        float a = 0.0f;
        for (int i = threadIdx.x; i < nblocks; i += 256) a += partial[(long)i * nv + v];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[v] = red[0];
        __syncthreads();
    }
}

} // namespace fn2

static int ms_geometry(int B, int H, int W, int start_scale, int num_scales, int *kmax, int *bx, int *by)
{
    if (B < 0 || H < 1 || W < 1 || num_scales < 1 || num_scales > fn2::MS_MAX_SCALES) return FN2_EINVAL;
    if (start_scale < 1 || (start_scale & (start_scale - 1))) return FN2_EINVAL;
    const int k = start_scale << (num_scales - 1);
    if (k / start_scale > 16 || start_scale > 16) return FN2_EUNSUPPORTED;   // <= 256 finest cells per workgroup
    *kmax = k; *bx = (W + k - 1) / k; *by = (H + k - 1) / k;
    return FN2_OK;
}

extern "C" size_t fn2_multiscale_workspace_bytes(int B, int H, int W, int start_scale, int num_scales)
{
    int kmax, bx, by;
    if (ms_geometry(B, H, W, start_scale, num_scales, &kmax, &bx, &by) != FN2_OK) return 0;
    return (size_t)B * bx * by * 2 * num_scales * sizeof(float);
}

extern "C" int fn2_multiscale_loss(const float *const *outputs, const float *target, float *sums, float *const *grads,
                                   const float *weights, float grad_scale, int norm, int B, int H, int W, int start_scale,
                                   int num_scales, float div_flow, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace fn2;
    MsArgs a;
    int rc = ms_geometry(B, H, W, start_scale, num_scales, &a.kmax, &a.bx, &a.by);
    if (rc != FN2_OK) return rc;
    if (!outputs || !target || !sums || !workspace || (norm != 1 && norm != 2)) return FN2_EINVAL;
    if (workspace_bytes < fn2_multiscale_workspace_bytes(B, H, W, start_scale, num_scales)) return FN2_EINVAL;
    if (!aligned(target, 4) || !aligned(sums, 4) || !aligned(workspace, 4)) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.target = target; a.partial = static_cast<float *>(workspace);
    a.B = B; a.H = H; a.W = W; a.s0 = start_scale; a.ns = num_scales; a.div_flow = div_flow; a.norm = norm;
    a.vec4 = (W % 4 == 0) && aligned(target, 16);
    for (int i = 0; i < MS_MAX_SCALES; ++i) { a.out[i] = nullptr; a.grad[i] = nullptr; a.gw[i] = 0.0f; }
    for (int i = 0; i < num_scales; ++i) {
        if (!outputs[i]) return FN2_EINVAL;
        a.out[i] = outputs[i];
        a.grad[i] = grads ? grads[i] : nullptr;
        const int k = start_scale << i;
        const double ni = (double)B * (norm == 2 ? 1 : 2) * (H / k) * (W / k);
        a.gw[i] = (grads && weights && ni > 0) ? (float)((double)grad_scale * (double)weights[i] / ni) : 0.0f;
    }
    const int nblocks = B * a.bx * a.by;
    if (nblocks == 0) {
        hipLaunchKernelGGL(multiscale_reduce_kernel, dim3(1), dim3(256), 0, s, a.partial, sums, 0, 2 * num_scales);
        return launch_status();
    }
    hipLaunchKernelGGL(multiscale_l1_epe_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, a);
    rc = launch_status();
    if (rc != FN2_OK) return rc;
    hipLaunchKernelGGL(multiscale_reduce_kernel, dim3(1), dim3(256), 0, s, a.partial, sums, nblocks, 2 * num_scales);
    return launch_status();
}

extern "C" int fn2_multiscale_l1_epe(const float *const *outputs, const float *target, float *sums, float *const *grads,
                                     const float *weights, float grad_scale, int B, int H, int W, int start_scale,
                                     int num_scales, float div_flow, void *workspace, size_t workspace_bytes, void *stream)
{
    return fn2_multiscale_loss(outputs, target, sums, grads, weights, grad_scale, 1, B, H, W, start_scale, num_scales, div_flow,
                               workspace, workspace_bytes, stream);
}
