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
// d(loss)/d(out_i) = grad_scale * w_i / N_i * sign(out_i - t_i), and leaves per-workgroup partial sums; the LAST workgroup to
// finish (a ticket counter in the workspace) adds them up in a fixed order -- deterministic, whichever workgroup that is -- and
// writes the 2*ns sums and the two weighted means (loss, epe).  Round 6: the predictions a thread compares against are loaded
// up front, together with the target rows (one memory round trip instead of one per level), and the separate reduce launch
// (9.1 us for 3 840 floats) is gone.
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
    float coef[2 * MS_MAX_SCALES]; // w_i / N_i (L1 sums) then w_i / (N_i / 2) (2-norm sums): loss / epe = sum_i coef * sums
    const float *target;
    float *partial;                // [nblocks][2 * ns]
    unsigned *ticket;              // zero before the launch; the last workgroup resets it
    float *sums;                   // 2 * ns, written by the last workgroup
    float *loss_epe;               // NULL or 2 floats: [loss (norm 1: sum_i coef_i sums_i; norm 2: = epe), epe]
    int B, H, W, s0, ns, kmax, bx, by, vec4, norm;
    float div_flow;
};

__global__ __launch_bounds__(256) void multiscale_l1_epe_kernel(MsArgs p)
{
    __shared__ float lv[2][2][256];      // [ping-pong][channel][cell]
    __shared__ float red[4][2 * MS_MAX_SCALES];
    __shared__ float fin[4][2 * MS_MAX_SCALES];
    __shared__ unsigned last;
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
    // the predictions this thread compares against at every level it takes part in: issued before the target rows, so that the
    // level loop below touches no global memory any more
    float o0[MS_MAX_SCALES], o1[MS_MAX_SCALES];
    {
        int nn = n, kk = k;
#pragma unroll
        for (int i = 0; i < MS_MAX_SCALES; ++i) {
            o0[i] = o1[i] = 0.0f;
            if (i < p.ns) {
                const int Hi = p.H / kk, Wi = p.W / kk;
                if (tid < nn * nn) {
                    const int cy = tid / nn, cx = tid - cy * nn;
                    const int gy = Y0 / kk + cy, gx = X0 / kk + cx;
                    if (gy < Hi && gx < Wi) {
                        const long o = ((long)b * 2 * Hi + gy) * Wi + gx;
                        o0[i] = p.out[i][o];
                        o1[i] = p.out[i][o + (long)Hi * Wi];
                    }
                }
                nn >>= 1; kk <<= 1;
            }
        }
    }
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
                    float4 v[4];
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy) v[yy] = *reinterpret_cast<const float4 *>(T + (long)(y0 + yy) * p.W + x0);
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy) {
                        sum = sum + p.div_flow * v[yy].x; sum = sum + p.div_flow * v[yy].y;
                        sum = sum + p.div_flow * v[yy].z; sum = sum + p.div_flow * v[yy].w;
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
#pragma unroll
    for (int i = 0; i < MS_MAX_SCALES; ++i) {
        if (i >= p.ns) break;
        const int Hi = p.H / k, Wi = p.W / k;
        if (tid < n * n) {
            const int cy = tid / n, cx = tid - cy * n;
            const int gy = Y0 / k + cy, gx = X0 / k + cx;
            if (gy < Hi && gx < Wi) {
                const float inv = (float)(k * k);
                const long o = ((long)b * 2 * Hi + gy) * Wi + gx, plane = (long)Hi * Wi;
                const float d0 = o0[i] - lv[cur][0][tid] / inv;
                const float d1 = o1[i] - lv[cur][1][tid] / inv;
                l1[i] = fabsf(d0) + fabsf(d1);
                ep[i] = __fsqrt_rn(d0 * d0 + d1 * d1);
                if (p.grad[i]) {
                    if (p.norm == 2) {
                        const float g = ep[i] > 0.0f ? p.gw[i] / ep[i] : 0.0f;
                        store_out(p.grad[i] + o, g * d0);
                        store_out(p.grad[i] + o + plane, g * d1);
                    } else {
                        store_out(p.grad[i] + o, d0 > 0.0f ? p.gw[i] : (d0 < 0.0f ? -p.gw[i] : 0.0f));
                        store_out(p.grad[i] + o + plane, d1 > 0.0f ? p.gw[i] : (d1 < 0.0f ? -p.gw[i] : 0.0f));
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
#pragma unroll
    for (int i = 0; i < MS_MAX_SCALES; ++i) {
        if (i >= p.ns) break;
        float a = l1[i], e = ep[i];
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); e += __shfl_down(e, off, 64); }
        if (lane == 0) { red[wave][i] = a; red[wave][p.ns + i] = e; }
    }
    __syncthreads();
    const int nv = 2 * p.ns;
    // partial sums written through to memory (the eight L2s are not coherent with each other), then one ticket per workgroup
    if (tid < nv) store_out(p.partial + (long)blockIdx.x * nv + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned tk = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (tk == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    // the last workgroup: every other workgroup's partial sums are in memory (release before its ticket, acquire after ours).
    // Fixed summation order whichever workgroup gets here: thread-strided rows, shuffle tree, four waves in order.
    __threadfence();
    const int nblocks = (int)gridDim.x;
    float acc[2 * MS_MAX_SCALES];
#pragma unroll
    for (int v = 0; v < 2 * MS_MAX_SCALES; ++v) acc[v] = 0.0f;
    for (int i = tid; i < nblocks; i += 256) {
#pragma unroll
        for (int v = 0; v < 2 * MS_MAX_SCALES; ++v)
            if (v < nv) acc[v] += __builtin_nontemporal_load(p.partial + (long)i * nv + v);
    }
#pragma unroll
    for (int v = 0; v < 2 * MS_MAX_SCALES; ++v) {
        float a = acc[v];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) fin[wave][v] = a;
    }
    __syncthreads();
    if (tid < nv) {
        const float sv = (fin[0][tid] + fin[1][tid]) + (fin[2][tid] + fin[3][tid]);
        store_out(p.sums + tid, sv);
        red[0][tid] = sv * p.coef[tid];
    }
    __syncthreads();
    if (tid == 0) {
        if (p.loss_epe) {
            float l = 0.0f, e = 0.0f;
            for (int i = 0; i < p.ns; ++i) { l += red[0][i]; e += red[0][p.ns + i]; }
            store_out(p.loss_epe, p.norm == 2 ? e : l);      // L2(): the expression of EPE (losses.py:21-26)
            store_out(p.loss_epe + 1, e);
        }
        __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // primed for the next launch
    }
}

// d loss / d out_i for an incoming gradient g (a device scalar): out[i] = unit[i] * g for all scales in ONE launch (the backward of
// the autograd node: five tensors, 1 MB in all)
struct MsScaleArgs {
    const float *in[MS_MAX_SCALES];
    float *out[MS_MAX_SCALES];
    long n[MS_MAX_SCALES];
    const float *scale;
    int ns;
};

__global__ __launch_bounds__(256) void multiscale_scale_kernel(MsScaleArgs p)
{
    const float g = *p.scale;
    const long stride = (long)gridDim.x * 256;
#pragma unroll
    for (int i = 0; i < MS_MAX_SCALES; ++i) {
        if (i >= p.ns) break;
        for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < p.n[i]; j += stride) store_out(p.out[i] + j, p.in[i][j] * g);
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

static size_t ms_partial_bytes(int B, int bx, int by, int num_scales)
{
    return ((size_t)B * bx * by * 2 * num_scales * sizeof(float) + 63) / 64 * 64;
}

extern "C" size_t fn2_multiscale_workspace_bytes(int B, int H, int W, int start_scale, int num_scales)
{
    int kmax, bx, by;
    if (ms_geometry(B, H, W, start_scale, num_scales, &kmax, &bx, &by) != FN2_OK) return 0;
    return ms_partial_bytes(B, bx, by, num_scales) + 64;       // partial sums + the ticket counter (its own 64-byte line)
}

static int ms_launch(const float *const *outputs, const float *target, float *sums, float *loss_epe, float *const *grads,
                     const float *weights, float grad_scale, int norm, int B, int H, int W, int start_scale, int num_scales,
                     float div_flow, void *workspace, size_t workspace_bytes, int primed, void *stream)
{
    using namespace fn2;
    MsArgs a;
    int rc = ms_geometry(B, H, W, start_scale, num_scales, &a.kmax, &a.bx, &a.by);
    if (rc != FN2_OK) return rc;
    if (!outputs || !target || !sums || !workspace || (norm != 1 && norm != 2)) return FN2_EINVAL;
    if (workspace_bytes < fn2_multiscale_workspace_bytes(B, H, W, start_scale, num_scales)) return FN2_EINVAL;
    if (!aligned(target, 4) || !aligned(sums, 4) || !aligned(workspace, 4) || (loss_epe && !aligned(loss_epe, 4))) return FN2_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.target = target; a.partial = static_cast<float *>(workspace);
    a.ticket = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + ms_partial_bytes(B, a.bx, a.by, num_scales));
    a.sums = sums; a.loss_epe = loss_epe;
    a.B = B; a.H = H; a.W = W; a.s0 = start_scale; a.ns = num_scales; a.div_flow = div_flow; a.norm = norm;
    a.vec4 = (W % 4 == 0) && aligned(target, 16);
    for (int i = 0; i < MS_MAX_SCALES; ++i) { a.out[i] = nullptr; a.grad[i] = nullptr; a.gw[i] = 0.0f; }
    for (int i = 0; i < 2 * MS_MAX_SCALES; ++i) a.coef[i] = 0.0f;
    for (int i = 0; i < num_scales; ++i) {
        const int k = start_scale << i;
        const double px = (double)B * (H / k) * (W / k);
        if (!outputs[i] && px > 0) return FN2_EINVAL;          // (a level without elements may come as a null pointer)
        a.out[i] = outputs[i];
        a.grad[i] = grads ? grads[i] : nullptr;
        const double ni = px * (norm == 2 ? 1 : 2);
        a.gw[i] = (grads && weights && ni > 0) ? (float)((double)grad_scale * (double)weights[i] / ni) : 0.0f;
        // the weighted means of losses.py:77-78: w_i / N_i as fp32 factors of the fp32 sums (N_i = elements for |.|, pixels for ||.||)
        a.coef[i] = (weights && px > 0) ? (float)((double)weights[i] / (2.0 * px)) : 0.0f;
        a.coef[num_scales + i] = (weights && px > 0) ? (float)((double)weights[i] / px) : 0.0f;
    }
    const int nblocks = B * a.bx * a.by;
    if (nblocks == 0) {   // empty batch: every sum is zero
        hipError_t e = hipMemsetAsync(sums, 0, 2 * num_scales * sizeof(float), s);
        if (e == hipSuccess && loss_epe) e = hipMemsetAsync(loss_epe, 0, 2 * sizeof(float), s);
        return e == hipSuccess ? FN2_OK : (int)e;
    }
    if (!primed && hipMemsetAsync(a.ticket, 0, sizeof(unsigned), s) != hipSuccess) return launch_status();
    hipLaunchKernelGGL(multiscale_l1_epe_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, a);
    return launch_status();
}

extern "C" int fn2_multiscale_loss(const float *const *outputs, const float *target, float *sums, float *const *grads,
                                   const float *weights, float grad_scale, int norm, int B, int H, int W, int start_scale,
                                   int num_scales, float div_flow, void *workspace, size_t workspace_bytes, void *stream)
{
    return ms_launch(outputs, target, sums, nullptr, grads, weights, grad_scale, norm, B, H, W, start_scale, num_scales, div_flow,
                     workspace, workspace_bytes, 0, stream);
}

extern "C" int fn2_multiscale_loss_fused(const float *const *outputs, const float *target, float *sums, float *loss_epe,
                                         float *const *grads, const float *weights, float grad_scale, int norm, int B, int H, int W,
                                         int start_scale, int num_scales, float div_flow, void *workspace, size_t workspace_bytes,
                                         int workspace_primed, void *stream)
{
    if (!loss_epe || !weights) return FN2_EINVAL;
    return ms_launch(outputs, target, sums, loss_epe, grads, weights, grad_scale, norm, B, H, W, start_scale, num_scales, div_flow,
                     workspace, workspace_bytes, workspace_primed, stream);
}

extern "C" int fn2_multiscale_scale_grads(const float *const *unit_grads, float *const *grads, const int64_t *numel, int num_scales,
                                          const float *scale, void *stream)
{
    using namespace fn2;
    if (!unit_grads || !grads || !numel || !scale || num_scales < 1 || num_scales > MS_MAX_SCALES) return FN2_EINVAL;
    MsScaleArgs a;
    long most = 0;
    for (int i = 0; i < MS_MAX_SCALES; ++i) { a.in[i] = nullptr; a.out[i] = nullptr; a.n[i] = 0; }
    for (int i = 0; i < num_scales; ++i) {
        if (numel[i] < 0 || (numel[i] > 0 && (!unit_grads[i] || !grads[i]))) return FN2_EINVAL;
        a.in[i] = unit_grads[i]; a.out[i] = grads[i]; a.n[i] = (long)numel[i];
        if (a.n[i] > most) most = a.n[i];
    }
    if (most == 0) return FN2_OK;
    a.scale = scale; a.ns = num_scales;
    const long blocks = (most + 255) / 256;
    hipLaunchKernelGGL(multiscale_scale_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return launch_status();
}

extern "C" int fn2_multiscale_l1_epe(const float *const *outputs, const float *target, float *sums, float *const *grads,
                                     const float *weights, float grad_scale, int B, int H, int W, int start_scale,
                                     int num_scales, float div_flow, void *workspace, size_t workspace_bytes, void *stream)
{
    return fn2_multiscale_loss(outputs, target, sums, grads, weights, grad_scale, 1, B, H, W, start_scale, num_scales, div_flow,
                               workspace, workspace_bytes, stream);
}
