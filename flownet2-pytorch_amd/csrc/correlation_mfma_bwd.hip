// correlation_mfma_bwd.hip -- correlation backward (both input gradients) on the gfx950 matrix cores.
//
// Replaces reference kernels correlation_backward_input1 / correlation_backward_input2
// (correlation_cuda_kernel.cu:150-241, :243-334; one launch per batch item each, :522-554) for
// the FlowNetC configuration family (kernel_size = 1, stride1 = 1, stride2 = 2, pad == md, fp32).
// For those parameters, with d = (tj-dr, ti-dr) in lattice units of 2 pixels:
//     gI1[n,c,p] = (1/C) * sum_d gO[n, tc(d),  p    ] * in2[n,c, p + 2d]
//     gI2[n,c,p] = (1/C) * sum_d gO[n, tc(d),  p - 2d] * in1[n,c, p - 2d]
// Both are the same banded contraction over the (2dr+1)^2 neighbours of the "centre" pixel p:
//     g[p, c] = sum_{d'} G[p, d'] * nbr[p + 2d', c]
// with  nbr = in2, G[p,d'] = gO[tc(d')][p]                 for gI1  (FLIP = 0)
//       nbr = in1, G[p,d'] = gO[tc(-d')][p + 2d']          for gI2  (FLIP = 1, d' = -d)
// so ONE kernel computes either gradient; FLIP only changes which gO row / column offset feeds
// each row of the G tile while it is staged.
//
// MFMA mapping (same parity lattice and 4x4 pixel blocks as the forward, correlation_mfma.hip):
// M = the 16 centre pixels of an A block, K = neighbour pixels (one 4x4 B block = 4 k-steps of 4),
// N = 16 channels:  acc[a, c] += G[a, b] * nbr[b, c]   with v_mfma_f32_16x16x4_f32 (exact fp32).
// The G operand of a (centre block, neighbour block) pair is the forward's output tile for that
// pair: it is staged in LDS in the forward epilogue's layout [ai][bi][ti][x] and read back with the
// inverse of the forward's accumulator->LDS scatter, once per neighbour row block u, into 48
// registers that are then reused for every channel tile.
//
// Work decomposition: one workgroup (8 waves) = (n, y-parity, row group of 4 lattice rows, x tile of
// 64 pixels, channel group of CG channels); it loops over the NV neighbour row blocks u (the sum
// over neighbours must stay inside one workgroup -- no atomics, deterministic) and, per u, over
// CG/16 channel tiles streamed through LDS (double buffered, [ch][xpar][row][col] like the
// forward's B tile but with channel stride 434 = 18 mod 32 so that the 16 channels x 2 k-slots of
// a 32-lane group hit 32 distinct banks).  Epilogue: accumulators -> LDS [ch][ai][x] -> each
// (channel, row) leaves as one coalesced 256 B store, scaled by 1/C.
//
// Algorithmic HBM bytes (both gradients): read gO twice + in1 + in2, write gI1 + gI2.
#include "corr_params.h"

namespace fn2 {
namespace mb {

constexpr int TILE_X = 64;
constexpr int DR_MAX = 10;
constexpr int B_COLS = TILE_X / 2 + 2 * DR_MAX;                          // 52 lattice columns incl. halo
constexpr int B_ROW = B_COLS, B_PAR = 4 * B_ROW, N_CH = 2 * B_PAR + 18;  // 52, 208, 434
constexpr int CK = 16;                                                    // channels per tile (MFMA N)
constexpr int N_FLOATS = CK * N_CH;                                       // 6944 per buffer
constexpr int G_RS = 66;                                                  // G tile x stride
constexpr int G_FLOATS = 16 * (2 * DR_MAX + 1) * G_RS;                    // 22176
constexpr int E_RS = 65;                                                  // epilogue x stride
static_assert(N_CH % 32 == 18, "16 channels x 2 k-slots must hit 32 distinct banks");

typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(2))) f2;

struct Args {
    const float *nbr[2];   // [0] = in2 (neighbours of gradInput1), [1] = in1 (neighbours of gradInput2)
    const float *gout;
    float *gin[2];         // [0] = gradInput1 (FLIP = 0), [1] = gradInput2 (FLIP = 1)
    int C, H, W;
    int dr, D;
    int NRG, NXT, NCG;
    int nflip;             // 2: both gradients in this launch (task id's top factor), 1: only `flip0`
    int flip0;
};

// Both gradients run in ONE launch (FLIP is the slowest-varying factor of the task id): at the FlowNetC shape a
// gradient has 384 tasks = 1.5 rounds of 256 single-workgroup CUs, the two together 768 = exactly 3 rounds.
// VAR: profiling switches (0 = the real kernel): 1 no MFMA, 2 no neighbour staging, 4 no G staging, 8 no stores
template <int NV, int NCT, int VAR = 0>
__global__ __launch_bounds__(512, 2) void corr_bwd_mfma_f32(Args p)
{
    constexpr int CG = NCT * CK;
    constexpr int E_FLOATS = CG * 4 * E_RS;
    constexpr int GS_FLOATS = (G_FLOATS > E_FLOATS ? G_FLOATS : E_FLOATS);
    __shared__ __attribute__((aligned(16))) float smem[GS_FLOATS + 2 * N_FLOATS];
    float *Gs = smem;
    float *Ns = smem + GS_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- task decode (channel group fastest: the groups sharing one G tile run together)
    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(t % p.NCG); t /= p.NCG;
    const int xt = (int)(t % p.NXT); t /= p.NXT;
    const int rg = (int)(t % p.NRG); t /= p.NRG;
    const int py = (int)(t & 1u); t >>= 1;
    const int nb = (int)gridDim.x / (p.nflip * 2 * p.NRG * p.NXT * p.NCG);   // batch size
    const int n = (int)(t % nb);
    const int FLIP = __builtin_amdgcn_readfirstlane(p.nflip == 2 ? (int)(t / nb) : p.flip0);

    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int X0 = xt * TILE_X;
    const int c_base = cg * CG;
    const float *nbr_n = p.nbr[FLIP] + ((long)n * p.C + c_base) * HW;
    const float *go_n = p.gout + (long)n * p.D * p.D * HW;

    // ---- neighbour-tile staging roles (as the forward's B tile): wave w stages row bi = w&3 of
    // channels 2k + (w>>2), k = 0..7; lane = lattice column jb (< 52), one float2 = (even, odd) x.
    const int s_bi = wave & 3;
    const int s_jb = lane;
    const int s_xb = X0 - 2 * p.dr + 2 * s_jb;
    const bool s_col_ok = (s_jb < B_COLS) && (s_xb >= 0) && (s_xb < p.W);
    const int n_dst = (wave >> 2) * N_CH + s_bi * B_ROW + s_jb;

    f2 rn[8];
    auto nbr_load = [&](int u, int ct) {
        if (VAR & 2) return;
        const int il = 4 * rg - p.dr + 4 * u + s_bi;     // neighbour lattice row
        const bool ok = s_col_ok && (il >= 0) && (il < HL);
        const float *src = ok ? nbr_n + (long)(ct * CK + (wave >> 2)) * HW + (long)(2 * il + py) * p.W + s_xb : nbr_n;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f2 v = *reinterpret_cast<const f2 *>(src + (ok ? (long)(2 * k) * HW : 0));
            rn[k] = ok ? v : (f2){0.0f, 0.0f};
        }
    };
    auto nbr_write = [&](int buf) {
        if (VAR & 2) return;
        float *N = Ns + buf * N_FLOATS;
        if (s_jb < B_COLS) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                N[n_dst + k * 2 * N_CH] = rn[k][0];
                N[n_dst + k * 2 * N_CH + B_PAR] = rn[k][1];
            }
        }
    };

    // ---- G tile staging: rows (plane = ai*4+bi, ti), 64 pixels each; wave w takes planes w, w+8;
    // lane = (row-in-pair hr, x pair hx): one float2 per lane, two rows per instruction.
    const int hx = lane & 31, hr = lane >> 5;
    auto g_stage = [&](int u) {
        if (VAR & 4) return;
        constexpr int NP = (2 * DR_MAX + 1 + 1) / 2;   // ti pairs
        const int HWi = p.H * p.W;                     // 32-bit offsets: D*D*H*W < 2^31 (checked by the launcher)
        // all 2 x NP loads are issued before the first LDS write: at this point of the u loop the 48 G-fragment
        // registers of the previous u are dead, so the registers are free and the load latency is paid once
        f2 rg_[2][NP];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pl = wave + 8 * h;
            const int ai = pl >> 2, bi = pl & 3;
            const int tj = 4 * u + bi - ai;          // displacement row index of this plane
            const int IL = 4 * rg + ai;              // centre lattice row
            const bool row_ok = (tj >= 0) && (tj < p.D) && (IL < HL);
            int base, step;                          // element offset of (ti = hr) and its increment per ti pair
            bool col_ok;
            if (!FLIP) {
                const int x = X0 + 2 * hx;
                col_ok = (x < p.W);
                base = (tj * p.D + hr) * HWi + (2 * IL + py) * p.W + x;
                step = 2 * HWi;
            } else {
                // G'[tj', ti'][p] = gO[(2dr - tj')*D + (2dr - ti')][p + 2d'],  d' = (tj'-dr, ti'-dr)
                const int ys = 2 * IL + py + 2 * (tj - p.dr);
                col_ok = (ys >= 0) && (ys < p.H);
                base = ((2 * p.dr - tj) * p.D + (2 * p.dr - hr)) * HWi + ys * p.W + X0 + 2 * hx + 2 * (hr - p.dr);
                step = -2 * HWi + 4;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int ti = 2 * i + hr;
                bool ok = row_ok && col_ok && (ti < p.D);
                if (FLIP) {
                    const int x = X0 + 2 * hx + 2 * (ti - p.dr);
                    ok = ok && (x >= 0) && (x < p.W);
                }
                f2 v = (f2){0.0f, 0.0f};
                if (ok) v = *reinterpret_cast<const f2 *>(go_n + (base + i * step));
                rg_[h][i] = v;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pl = wave + 8 * h;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int ti = 2 * i + hr;
                if (ti < p.D) *reinterpret_cast<f2 *>(Gs + (pl * p.D + ti) * G_RS + 2 * hx) = rg_[h][i];
            }
        }
    };

    // ---- MFMA roles
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) << 1;
    const int fi = lane & 15, fq = lane >> 4;
    // A operand (G): lane (i = centre pixel (ai, aj), q = bj); k-step (v, s = bi)
    const int g_ai = fi >> 2, g_aj = fi & 3;
    // B operand (neighbours): lane (q = bj, j = channel)
    const int n_frag = fi * N_CH + xpar * B_PAR + 4 * a0 + fq;   // + s*B_ROW + 4*(ab+v)

    f4 acc[2][NCT];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ab][ct] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

    // valid neighbour row blocks: rows 4rg - dr + 4u .. +3 must intersect [0, HL)
    int u_lo = 0, u_hi = NV - 1;
    while (u_lo < NV && (4 * rg - p.dr + 4 * u_lo + 3 < 0)) ++u_lo;
    while (u_hi >= 0 && (4 * rg - p.dr + 4 * u_hi >= HL)) --u_hi;

    if (u_lo <= u_hi) {
        int buf = 0;
        nbr_load(u_lo, 0);
        for (int u = u_lo; u <= u_hi; ++u) {
            g_stage(u);
            nbr_write(buf);
            __syncthreads();
            float gfr[2][NV][4];
#pragma unroll
            for (int ab = 0; ab < 2; ++ab)
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int ti = 4 * v + fq - g_aj;
                        bool ok = true;
                        if (v == 0) ok = (ti >= 0);
                        if (v >= NV - 2) ok = ok && (ti < p.D);
                        const int x = 2 * (4 * (a0 + ab) + g_aj) + xpar;
                        const float gv = Gs[((g_ai * 4 + s) * p.D + (ok ? ti : 0)) * G_RS + x];
                        gfr[ab][v][s] = ok ? gv : 0.0f;
                    }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                // prefetch the next neighbour chunk (next channel tile, or tile 0 of the next u)
                const bool last_ct = (ct == NCT - 1);
                const bool more = !last_ct || (u < u_hi);
                if (more) nbr_load(last_ct ? u + 1 : u, last_ct ? 0 : ct + 1);
                const float *N = Ns + buf * N_FLOATS;
                float nf[2][NV + 1];
#pragma unroll
                for (int j = 0; j < NV + 1; ++j) nf[0][j] = N[n_frag + 4 * j];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int cur = s & 1, nxt = cur ^ 1;
                    if (s + 1 < 4) {
#pragma unroll
                        for (int j = 0; j < NV + 1; ++j) nf[nxt][j] = N[n_frag + (s + 1) * B_ROW + 4 * j];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < NV; ++v)
#pragma unroll
                        for (int ab = 0; ab < 2; ++ab)
                            if (VAR & 1) asm volatile("" ::"v"(gfr[ab][v][s]), "v"(nf[cur][ab + v]));
                            else acc[ab][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(gfr[ab][v][s], nf[cur][ab + v], acc[ab][ct], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more) nbr_write(buf ^ 1);
                __syncthreads();
                buf ^= 1;
            }
        }
    }

    // ---- epilogue: acc[ab][ct][r] = g[centre pixel (ai = l>>4, aj = r) of block a0+ab][channel 16ct + (l&15)]
    {
        float *Es = smem;
        const int e_ai = fq, e_ch = fi;
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = 2 * (4 * (a0 + ab) + r) + xpar;
                    Es[((ct * CK + e_ch) * 4 + e_ai) * E_RS + x] = acc[ab][ct][r];
                }
        __syncthreads();
        const float fC = (float)p.C;
        const bool pow2 = (p.C & (p.C - 1)) == 0;
        const float rC = 1.0f / fC;
        const int xg = X0 + lane;
        float *gin_n = p.gin[FLIP] + ((long)n * p.C + c_base) * HW;
        for (int R = wave; R < CG * 4; R += 8) {
            const int ch = R >> 2, ai = R & 3;
            const int IL = 4 * rg + ai;
            if (IL >= HL) continue;
            if (xg < p.W) {
                float val = Es[R * E_RS + lane];
                val = pow2 ? val * rC : val / fC;    // sum / nelems (correlation_cuda_kernel.cu:238,:331)
                if (!(VAR & 8)) gin_n[(long)ch * HW + (long)(2 * IL + py) * p.W + xg] = val;
            }
        }
    }
}

template <int NV, int NCT>
static int launch(const Args &a, long ntasks, hipStream_t s)
{
    hipLaunchKernelGGL((corr_bwd_mfma_f32<NV, NCT>), dim3((unsigned)ntasks), dim3(512), 0, s, a);
    return launch_status();
}

template <int NCT>
static int launch_nv(int NV, const Args &a, long ntasks, hipStream_t s)
{
    switch (NV) {
    case 2: return launch<2, NCT>(a, ntasks, s);
    case 3: return launch<3, NCT>(a, ntasks, s);
    case 4: return launch<4, NCT>(a, ntasks, s);
    case 5: return launch<5, NCT>(a, ntasks, s);
    case 6: return launch<6, NCT>(a, ntasks, s);
    default: return FN2_EUNSUPPORTED;
    }
}

} // namespace mb

bool corr_bwd_mfma_f32_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    if (!corr_mfma_f32_applicable(dtype, C, H, W, pad, k, md, s1, s2)) return false;
    return C % 32 == 0;
}

// tune: 0 = shipped configuration (both gradients in one launch; 64- or 32-channel groups, see below);
//       1 = force 32-channel groups; 2 = force 64; 3 = one launch per gradient
int corr_backward_mfma_f32(const float *in1, const float *in2, const float *gout, float *g1, float *g2,
                           int B, int C, int H, int W, int md, int tune, hipStream_t s)
{
    if (!aligned(in1, 8) || !aligned(in2, 8) || !aligned(gout, 8)) return FN2_EALIGN;
    if ((long)(2 * (md / 2) + 1) * (2 * (md / 2) + 1) * H * W >= (1L << 31)) return FN2_EUNSUPPORTED;
    mb::Args a;
    a.gout = gout;
    a.C = C; a.H = H; a.W = W;
    a.dr = md / 2; a.D = 2 * a.dr + 1;
    const int NV = 1 + (a.dr + 1) / 2;
    a.NRG = (H / 2 + 3) / 4;
    a.NXT = (W + mb::TILE_X - 1) / mb::TILE_X;
    // 64-channel groups stage the G tile half as often as 32-channel groups, but the grid must also fill the
    // 256 CUs (one workgroup each) in whole rounds: pick the group size with the better last-round occupancy,
    // preferring 64 when they are close.  tune 3: one launch per gradient (A/B against the fused launch).
    const bool fused = (tune != 3);   // tune 10 + v: profiling variant v of the fused launch
    const long base = (long)B * 2 * a.NRG * a.NXT * (fused ? 2 : 1);
    auto round_eff = [](long t) { const long r = (t + 255) / 256; return r ? (double)t / (double)(r * 256) : 1.0; };
    bool g64 = (C % 64 == 0) && tune != 1;   // tune 2: 64 where possible, no occupancy heuristic
    if (g64 && (tune == 0 || tune == 3) && round_eff(base * (C / 32)) > 1.1 * round_eff(base * (C / 64))) g64 = false;
    a.NCG = C / (g64 ? 64 : 32);
    const long ntasks = base * a.NCG;
    if (ntasks == 0) return FN2_OK;
    a.nbr[0] = in2; a.gin[0] = g1;
    a.nbr[1] = in1; a.gin[1] = g2;
    if (tune >= 10 && tune < 26 && NV == 6 && g64) {   // profiling instantiations (FlowNetC radius only)
        a.nflip = 2; a.flip0 = 0;
        switch (tune - 10) {
#define FN2_BV(V) case V: hipLaunchKernelGGL((mb::corr_bwd_mfma_f32<6, 4, V>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
            FN2_BV(1) FN2_BV(2) FN2_BV(4) FN2_BV(6) FN2_BV(7) FN2_BV(8) FN2_BV(15)
#undef FN2_BV
        default: return FN2_EUNSUPPORTED;
        }
    }
    if (fused) {
        a.nflip = 2; a.flip0 = 0;
        return g64 ? mb::launch_nv<4>(NV, a, ntasks, s) : mb::launch_nv<2>(NV, a, ntasks, s);
    }
    a.nflip = 1;
    for (int f = 0; f < 2; ++f) {
        a.flip0 = f;
        const int rc = g64 ? mb::launch_nv<4>(NV, a, ntasks, s) : mb::launch_nv<2>(NV, a, ntasks, s);
        if (rc != FN2_OK) return rc;
    }
    return FN2_OK;
}

} // namespace fn2
