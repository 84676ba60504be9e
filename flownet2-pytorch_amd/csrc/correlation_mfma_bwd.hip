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
#include <type_traits>
#include "corr_params.h"
#include "bf16x3.h"

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

// ------------------------------------------------------------------------------------------------------------
// bf16x3 variant: the same contraction on v_mfma_f32_16x16x32_bf16 with every fp32 operand split exactly into three
// bf16 terms (bf16x3.h) -- fp32-class accuracy, and unlike the fp32 MFMA it runs concurrently with the vector-memory
// instructions that stage the next tiles (correlation_mfma.hip, finding 1).
//
// K = 32 per instruction = two adjacent neighbour blocks (v, v+1): k-slot t of lane group q is neighbour pixel
// (row q of the row block, lattice column 4(a+v)+t), t = 0..7, so
//   A operand (G):   lane (centre pixel i = (ai, aj), q) holds G[i][tj = 4u+q-ai][ti = 8 sl - aj + t], t = 0..7
//                    -- eight consecutive displacement columns, read from the G tile once per u and split in registers;
//   B operand (nbr): lane (channel j, q) holds nbr[row q][columns 4(a+2sl) .. +7][j]; the NV+1 blocks of a wave are read
//                    once per channel tile (16 B LDS reads), split in registers and shared by its two A blocks.
// Staging is LDS-DMA only (global_load_lds: no staging registers, no LDS write instructions): lanes whose source is
// outside the image read a 16 B block of zeros instead.
//   neighbour tile [ch][row][x] with both parities interleaved as in memory (strides 436 / 108 floats: 16 B aligned,
//                  109 bank quads per channel), one 16 B DMA piece per lane, 26 lanes per (channel, row);
//   G tile         [plane][ti][x], rows of 64 floats, plane stride 24*64 + 4 (one bank quad: the 16 planes a fragment
//                  read touches spread over the banks), one 16 B piece per lane, 4 ti rows per instruction.  The G tile
//                  of the NEXT u is staged in NCT parts, one per channel tile, while the fragments of the current u
//                  live in registers.  (A 4 B-per-lane DMA of the same tile measured 114 us per call.)
// Every phase issues its DMAs first, runs its MFMAs, then waits (vmcnt(0)) and passes ONE barrier.
// Preconditions beyond the fp32 kernel's: even radius (16 B aligned halo), W % 4 == 0, 16 B aligned inputs.
// Tile width TX: 64 (8 waves, one workgroup per CU) or 32 (4 waves, 80 640 B of LDS: TWO workgroups per CU).  A wave is
// stuck in vector-memory issue for as long as the CU's memory path needs to accept its DMAs (about as long as the
// wave's own MFMA phase) and the waves of one workgroup do that in lockstep; two independent workgroups per CU fill each
// other's gaps.  The price is the wider relative halo of the neighbour tile (72 columns for 32 instead of 104 for 64).
template <int TX> struct BT {
    static constexpr int NW = TX / 8;                    // waves: (x parity) x (pairs of A column blocks)
    static constexpr int RF = TX + 4 * DR_MAX;           // floats per neighbour row (both parities, halo): 104 / 72
    static constexpr int RPI = (TX == 64) ? 1 : 2;       // neighbour rows per DMA instruction (26 / 2 x 18 lanes)
    static constexpr int S_ROW = RF + (RPI == 1 ? 4 : 0);   // rows of one instruction are contiguous
    static constexpr int S_CH = 4 * S_ROW + 4;           // 436 / 292
    static constexpr int NB_FLOATS = CK * S_CH;
    static constexpr int PPR = TX / 4;                   // 16 B pieces per G row
    static constexpr int RG = 64 / PPR;                  // G rows (consecutive ti) per DMA instruction: 4 / 8
    static constexpr int NGRP = (2 * DR_MAX + 1 + RG - 1) / RG;   // 6 / 3 instructions per plane
    static constexpr int GP = (2 * DR_MAX + 1) * TX + 4; // G plane stride: 21 rows + one bank quad
    static constexpr int GB_FLOATS = 16 * GP;
    static constexpr int E_RS = TX + 1;
    static_assert(S_CH % 4 == 0 && (S_CH / 4) % 2 == 1, "16 B aligned, odd number of bank quads per channel");
    static_assert((16 * NGRP) % NW == 0, "G DMA instructions divide evenly over the waves");
};
__device__ __attribute__((aligned(16))) const float kZeroBlock[4] = {0.0f, 0.0f, 0.0f, 0.0f};

// VAR: profiling switches (0 = the real kernel): 1 no MFMA, 2 no neighbour DMA, 4 no G DMA, 8 no stores, 16 neighbour DMA
// before the G part, 128 s_memtime stamps of one phase dumped over gradInput1, 256 no s_setprio around the DMA issue,
// 1024 neighbour tile in 16 one-row DMAs per wave instead of 8 two-row ones (+10 us: the issue cost is per instruction)
template <int TX, int NV, int NCT, int VAR = 0>
__global__ __launch_bounds__(8 * TX, 2) void corr_bwd_mfma_bf16x3(Args p)
{
    typedef BT<TX> T;
    constexpr int NW = T::NW, S_ROW = T::S_ROW, S_CH = T::S_CH, NB_FLOATS = T::NB_FLOATS, GP = T::GP, GB_FLOATS = T::GB_FLOATS,
                  E_RS = T::E_RS;
    static_assert(NV % 2 == 0, "neighbour blocks are consumed in pairs");
    constexpr int NSL = NV / 2;                 // K = 32 slots per A block
    constexpr int CG = NCT * CK;
    constexpr int E_FLOATS = CG * 4 * E_RS;
    constexpr int GS_FLOATS = (GB_FLOATS > E_FLOATS ? GB_FLOATS : E_FLOATS);
    __shared__ __attribute__((aligned(16))) float smem[GS_FLOATS + 2 * NB_FLOATS];
    float *Gs = smem;
    float *Ns = smem + GS_FLOATS;
    typedef __attribute__((address_space(3))) void *lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // (Visiting the row groups from the middle outwards inside every XCD's share -- longest tasks first -- measured no
    // difference here; the forward kernel does gain from dispatching its all-padding tasks last.)
    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(t % p.NCG); t /= p.NCG;
    const int xt = (int)(t % p.NXT); t /= p.NXT;
    const int rg = (int)(t % p.NRG); t /= p.NRG;
    const int py = (int)(t & 1u); t >>= 1;
    const int nb = (int)gridDim.x / (p.nflip * 2 * p.NRG * p.NXT * p.NCG);
    const int n = (int)(t % nb);
    const int FLIP = __builtin_amdgcn_readfirstlane(p.nflip == 2 ? (int)(t / nb) : p.flip0);

    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int HWi = p.H * p.W;
    const int X0 = xt * TX;
    const int c_base = cg * CG;
    const float *nbr_n = (FLIP ? p.nbr[1] : p.nbr[0]) + ((long)n * p.C + c_base) * HW;
    const float *go_n = p.gout + (long)n * p.D * p.D * HW;
    const float *zeros = kZeroBlock;

    // ---- neighbour-tile DMA: 64 (channel, row) rows per tile, RPI rows per instruction, 8 instructions per wave:
    // instruction k of wave w covers rows (k*NW + w)*RPI .. of the (ch = row >> 2, bi = row & 3) list, i.e. always the
    // same bi0 = (w*RPI) & 3 and channels 2k + (w*RPI >> 2); lane = (row in instruction, 16 B piece)
    constexpr int PR = T::RF / 4;                                   // pieces per row
    constexpr int RPI = (VAR & 1024) ? 1 : T::RPI;                  // VAR 1024 (profiling): one row per instruction
    constexpr int NI = 64 / (NW * RPI), CSTEP = (NW * RPI) / 4 > 0 ? (NW * RPI) / 4 : 1;   // instructions per wave, channel step
    const int s_r = lane / PR, s_pc = lane - s_r * PR;
    const int s_bi = ((wave * RPI) & 3) + s_r, s_ch0 = (wave * RPI) >> 2;
    const int s_x = X0 - 2 * p.dr + 4 * s_pc;
    const bool s_col_ok = (s_x >= 0) && (s_x < p.W);
    auto nbr_dma = [&](int u, int ct, int buf) __attribute__((always_inline)) {
        if (VAR & 2) return;
        int HW_ = HWi;
        asm volatile("" : "+s"(HW_));    // as in g_dma
        const int il = 4 * rg - p.dr + 4 * u + s_bi;
        const bool ok = s_col_ok && (il >= 0) && (il < HL);
        const float *src = ok ? nbr_n + (long)(ct * CK + s_ch0) * HW_ + (2 * il + py) * p.W + s_x : zeros;
        const long step = ok ? CSTEP * (long)HW_ : 0;
        float *dst = Ns + buf * NB_FLOATS + s_ch0 * S_CH + ((wave * RPI) & 3) * S_ROW;
        if (lane < RPI * PR) {
#pragma unroll
            for (int k = 0; k < NI; ++k)
                __builtin_amdgcn_global_load_lds(src + k * step, (lds_ptr)(dst + k * CSTEP * S_CH), 16, 0, 0);
        }
    };

    // ---- G-tile DMA: rows (plane = ai*4+bi, ti) of TX pixels = TX/4 pieces of 16 B; one instruction = RG consecutive
    // ti of one plane (lane = (ti % RG, piece)), contiguous in LDS: Gs[plane*GP + ti*TX + x].  Wave w takes planes
    // w, w+NW, ..: 12 instructions per u, issued in NCT parts.  FLIP: the source row is shifted by 2(ti-dr) pixels; a piece that
    // straddles the image edge is fetched from the clamped position and the fragment read below compensates.
    // Per DMA the source is go_n + A(plane, u) + L[g]: A is wave-uniform (scalar arithmetic), L[g] the lane's part -- it does
    // not depend on u and is formed once per group g (-1: the lane's piece lies outside the image and reads zeros).  Keeping
    // the issue path to a handful of instructions matters: a phase issues 11 DMAs per wave, all waves at the same time.
    constexpr int GI = 12 / NCT;
    int g_lane[T::NGRP];
    {
        const int r = lane / T::PPR, x = X0 + 4 * (lane % T::PPR);
#pragma unroll
        for (int g = 0; g < T::NGRP; ++g) {
            const int ti = T::RG * g + r;
            if (!FLIP) g_lane[g] = (x < p.W && ti < p.D) ? ti * HWi + x : -1;
            else {
                // G'[tj', ti'][p] = gO[(2dr - tj')*D + (2dr - ti')][p + 2d'],  d' = (tj'-dr, ti'-dr): -ti' rows, x shifted
                const int xs = x + 2 * (ti - p.dr);
                g_lane[g] = (x < p.W && ti < p.D && xs >= -2 && xs <= p.W - 2) ? (p.D - 1 - ti) * HWi + min(max(xs, 0), p.W - 4) : -1;
            }
        }
    }
    const int g_rows_last = (2 * DR_MAX + 1) - T::RG * (T::NGRP - 1);   // ti rows of the last group that exist in the LDS tile
    auto g_dma = [&](int u, int pt) __attribute__((always_inline)) {
        if (VAR & 4) return;
        // opaque copies: the row addresses are cheap scalar arithmetic; left to the optimiser they are all hoisted out
        // of the u loop (hundreds of SGPRs, spilled)
        int D_ = p.D, HWi_ = HWi;
        asm volatile("" : "+s"(D_), "+s"(HWi_));
#pragma unroll
        for (int jj = 0; jj < GI; ++jj) {
            const int j = pt * GI + jj, hh = j / T::NGRP, g = j % T::NGRP;
            const int pl = wave + NW * hh;
            const int ai = pl >> 2, bi = pl & 3;
            const int tj = 4 * u + bi - ai;          // displacement row index of this plane
            const int IL = 4 * rg + ai;              // centre lattice row
            const int ys = 2 * IL + py + (FLIP ? 2 * (tj - p.dr) : 0);
            const bool row_ok = (tj >= 0) && (tj < D_) && (IL < HL) && (ys >= 0) && (ys < p.H);
            // !FLIP: plane tj*D (+ti), row ys;  FLIP: plane (2dr - tj)*D + (2dr - ti) = (D-1-tj)*D + (D-1-ti), row ys
            const int A = (FLIP ? (D_ - 1 - tj) : tj) * D_ * HWi_ + ys * p.W;
            const int L = g_lane[g];
            // one per-lane select, no wave-uniform branch: a negative offset marks "read zeros" (offsets are < 2^31)
            const int off = (A + L) | (L >> 31) | (row_ok ? 0 : (int)0x80000000);
            const float *src = off >= 0 ? go_n + off : zeros;
            if (g + 1 < T::NGRP || lane < g_rows_last * T::PPR)
                __builtin_amdgcn_global_load_lds(src, (lds_ptr)(Gs + pl * GP + g * 256), 16, 0, 0);   // RG rows = 256 floats
        }
    };
    // end of a phase: wait for this wave's DMAs and LDS reads, then one raw barrier.  (Leaving the G-tile DMAs in flight
    // across the barrier with a counted vmcnt measured slower: they delay the next phase's neighbour tile.)
    auto phase_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- MFMA roles
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) << 1;
    const int fi = lane & 15, fq = lane >> 4;
    const int g_ai = fi >> 2, g_aj = fi & 3;                   // A operand: centre pixel of the lane
    const int n_frag = fi * S_CH + fq * S_ROW + 8 * a0;        // B operand: 8 floats (4 columns x 2 parities) per block
    auto as_bf = [](const u4 &x) { return __builtin_bit_cast(bf16x8, x); };

    f4 acc[2][NCT];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ab][ct] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

    int u_lo = 0, u_hi = NV - 1;
    while (u_lo < NV && (4 * rg - p.dr + 4 * u_lo + 3 < 0)) ++u_lo;
    while (u_hi >= 0 && (4 * rg - p.dr + 4 * u_hi >= HL)) --u_hi;

    // VAR 128 (profiling): s_memtime stamps of one phase (second u, channel tile 1) of every wave of one workgroup,
    // dumped over the start of gradInput1 at the end
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i, bool on) __attribute__((always_inline)) { if ((VAR & 128) && on) ts[i] = __builtin_amdgcn_s_memtime(); };
    if (u_lo <= u_hi) {
        int buf = 0;
        nbr_dma(u_lo, 0, 0);
#pragma unroll
        for (int pt = 0; pt < NCT; ++pt) g_dma(u_lo, pt);
        phase_sync();
        for (int u = u_lo; u <= u_hi; ++u) {
            const bool tu = (u == u_lo + 1);
            const bool next_u = (u < u_hi);
            stamp(0, tu);
            // the neighbour tile of channel tile 1 (or of the next u) goes out before the fragment reads: the memory
            // path is the scarcest resource of this kernel and would otherwise idle during them
            if (NCT > 1 || next_u) nbr_dma(NCT > 1 ? u : u + 1, NCT > 1 ? 1 : 0, buf ^ 1);
            // G fragments of this u: 2 A blocks x NSL slots x 3 terms
            u4 ga[2][NSL][3];
#pragma unroll
            for (int ab = 0; ab < 2; ++ab)
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    float gv[8];
                    const int x = 2 * (4 * (a0 + ab) + g_aj) + xpar;
                    const float *Gp = Gs + (g_ai * 4 + fq) * GP;
                    if (!FLIP) {
#pragma unroll
                        for (int tt = 0; tt < 8; ++tt) {
                            const int ti = 8 * sl - g_aj + tt;
                            const bool ok = (ti >= 0) && (ti < p.D);
                            const float g = Gp[(ok ? ti : 0) * TX + x];
                            gv[tt] = ok ? g : 0.0f;
                        }
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 8; ++tt) {
                            const int ti = 8 * sl - g_aj + tt;
                            const int sh = 2 * (ti - p.dr);
                            const int P = X0 + (x & ~3) + sh;                 // source x of this pixel's 16 B piece
                            const int adj = P - min(max(P, 0), p.W - 4);       // 0, or -2 / +2 at the image edge
                            const bool ok = (ti >= 0) && (ti < p.D) && ((unsigned)(X0 + x + sh) < (unsigned)p.W);
                            const float g = Gp[(ok ? ti * TX + adj : 0) + x];
                            gv[tt] = ok ? g : 0.0f;
                        }
                    }
                    split3(gv, ga[ab][sl][0], ga[ab][sl][1], ga[ab][sl][2]);
                }
            stamp(1, tu);
            // every wave holds its fragments: the G tile may be overwritten with the next u's
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const bool last_ct = (ct == NCT - 1);
                const bool more = !last_ct || next_u;
                // DMAs of this phase: the neighbour tile of the next phase (already out for ct = 0) and part ct of the
                // next u's G tile
                stamp(2, tu && ct == 1);
                // (G part first and raised wave priority while issuing; VAR 16 / 256 switch them off -- differences are
                // within the run-to-run noise of a few us)
                if (!(VAR & 256)) __builtin_amdgcn_s_setprio(3);
                if (!(VAR & 16) && next_u) g_dma(u + 1, ct);
                if (ct > 0 && more) nbr_dma(last_ct ? u + 1 : u, last_ct ? 0 : ct + 1, buf ^ 1);
                if ((VAR & 16) && next_u) g_dma(u + 1, ct);
                if (!(VAR & 256)) __builtin_amdgcn_s_setprio(0);
                stamp(3, tu && ct == 1);
                const float *N = Ns + buf * NB_FLOATS + n_frag;
                // neighbour fragments: NV + 1 blocks x 4 columns of this parity, each split into three half operands;
                // a slot needs blocks 2sl .. 2sl+2, block 2sl+2 is kept for the next slot (sliding window keeps the
                // live set small: the G fragments already hold 72 registers)
                auto frag = [&](int b, u2 &h0, u2 &h1, u2 &h2) __attribute__((always_inline)) {
                    const f4 v0 = *reinterpret_cast<const f4 *>(N + 8 * b), v1 = *reinterpret_cast<const f4 *>(N + 8 * b + 4);
                    // the wave's x parity selects elements (par, par + 2) of each 16 B piece
                    const float r[4] = {xpar ? v0[1] : v0[0], xpar ? v0[3] : v0[2], xpar ? v1[1] : v1[0], xpar ? v1[3] : v1[2]};
                    split3_half(r, h0, h1, h2);
                };
                u2 hb[3][3];   // [term][block 2sl + 0..2]
                frag(0, hb[0][0], hb[1][0], hb[2][0]);
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    frag(2 * sl + 1, hb[0][1], hb[1][1], hb[2][1]);
                    frag(2 * sl + 2, hb[0][2], hb[1][2], hb[2][2]);
                    u4 nbv[2][3];
#pragma unroll
                    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
                        for (int T = 0; T < 3; ++T) {
                            const u2 lo = hb[T][ab], hi = hb[T][ab + 1];
                            nbv[ab][T] = (u4){lo[0], lo[1], hi[0], hi[1]};
                        }
                    // (G term, nbr term): 00 01 10 11 02 20
                    constexpr int PG[6] = {0, 0, 1, 1, 0, 2}, PN[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int ab = 0; ab < 2; ++ab) {
                            if (VAR & 1) asm volatile("" ::"v"(ga[ab][sl][PG[pr]]), "v"(nbv[ab][PN[pr]]));
                            else acc[ab][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(ga[ab][sl][PG[pr]]), as_bf(nbv[ab][PN[pr]]), acc[ab][ct], 0, 0, 0);
                        }
#pragma unroll
                    for (int T = 0; T < 3; ++T) hb[T][0] = hb[T][2];
                    __builtin_amdgcn_sched_barrier(0);
                }
                stamp(4, tu && ct == 1);
                phase_sync();
                stamp(6, tu && ct == 1);
                buf ^= 1;
            }
        }
    }

    // ---- epilogue (as corr_bwd_mfma_f32): acc[ab][ct][r] = g[centre pixel (ai = l>>4, aj = r) of block a0+ab][channel 16ct + (l&15)]
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
        float *gin_n = (FLIP ? p.gin[1] : p.gin[0]) + ((long)n * p.C + c_base) * HW;
        constexpr int RPE = 64 / TX;                      // (channel, row) rows per wave iteration
        const int xl = lane % TX, xg = X0 + xl;
        for (int R0 = wave * RPE; R0 < CG * 4; R0 += NW * RPE) {
            const int R = R0 + lane / TX;
            const int ch = R >> 2, ai = R & 3;
            const int IL = 4 * rg + ai;
            if (IL < HL && xg < p.W) {
                float val = Es[R * E_RS + xl];
                val = pow2 ? val * rC : val / fC;
                if (!(VAR & 8)) gin_n[(long)ch * HW + (long)(2 * IL + py) * p.W + xg] = val;
            }
        }
    }
    if ((VAR & 128) && blockIdx.x == 8 * 20 && lane == 0) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(p.gin[0]) + wave * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = ts[i];
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

// bf16x3 kernel preconditions beyond corr_bwd_mfma_f32_applicable: radius 10 (FlowNetC: max_displacement 20 or 21), C % 64,
// W % 4, 16 B aligned inputs
static bool bwd_bf16x3_ok(const float *in1, const float *in2, const float *gout, int C, int W, int dr)
{
    return dr == 10 && C % 64 == 0 && W % 4 == 0 && aligned(in1, 16) && aligned(in2, 16) && aligned(gout, 16);
}

// tune: 0 = shipped configuration: the bf16x3 kernel (32-px tiles, both gradients in one launch) where its preconditions
//           hold, otherwise the fp32 MFMA kernel as under 6;
//       6 = fp32 MFMA kernel, both gradients in one launch, 64- or 32-channel groups (see below); 1 = force 32-channel
//           groups; 2 = force 64; 3 = one launch per gradient; 10 + v = its profiling variant v;
//       4 = bf16x3 kernel or FN2_EUNSUPPORTED; 5 = bf16x3 with 64-px tiles; 40 + v = its profiling variant v
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
    a.nbr[0] = in2; a.gin[0] = g1;
    a.nbr[1] = in1; a.gin[1] = g2;
    const bool bf_ok = bwd_bf16x3_ok(in1, in2, gout, C, W, a.dr);
    if (tune == 0) tune = bf_ok ? 4 : 6;

    if (tune == 4 || tune == 5 || (tune >= 40 && tune < 1200)) {
        if (!bf_ok) return FN2_EUNSUPPORTED;
        const int TX = (tune == 5) ? 64 : 32;
        a.NXT = (W + TX - 1) / TX;
        a.NCG = C / 64;
        a.nflip = 2; a.flip0 = 0;
        const long ntasks = (long)B * 2 * a.NRG * a.NXT * 2 * a.NCG;
        if (ntasks == 0) return FN2_OK;
        if (tune == 5) { hipLaunchKernelGGL((mb::corr_bwd_mfma_bf16x3<64, 6, 4, 0>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status(); }
        switch (tune == 4 ? 0 : tune - 40) {
#define FN2_BV(V) case V: hipLaunchKernelGGL((mb::corr_bwd_mfma_bf16x3<32, 6, 4, V>), dim3((unsigned)ntasks), dim3(256), 0, s, a); return launch_status();
            FN2_BV(0) FN2_BV(1) FN2_BV(6) FN2_BV(7) FN2_BV(16) FN2_BV(128) FN2_BV(256) FN2_BV(1024)
#undef FN2_BV
        default: return FN2_EUNSUPPORTED;
        }
    }

    a.NXT = (W + mb::TILE_X - 1) / mb::TILE_X;
    // 64-channel groups stage the G tile half as often as 32-channel groups, but the grid must also fill the
    // 256 CUs (one workgroup each) in whole rounds: pick the group size with the better last-round occupancy,
    // preferring 64 when they are close.  tune 3: one launch per gradient (A/B against the fused launch).
    const bool fused = (tune != 3);
    const long base = (long)B * 2 * a.NRG * a.NXT * (fused ? 2 : 1);
    auto round_eff = [](long t) { const long r = (t + 255) / 256; return r ? (double)t / (double)(r * 256) : 1.0; };
    bool g64 = (C % 64 == 0) && tune != 1;   // tune 2: 64 where possible, no occupancy heuristic
    if (g64 && (tune == 6 || tune == 3) && round_eff(base * (C / 32)) > 1.1 * round_eff(base * (C / 64))) g64 = false;
    a.NCG = C / (g64 ? 64 : 32);
    const long ntasks = base * a.NCG;
    if (ntasks == 0) return FN2_OK;
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
