// correlation_mfma_f64.hip -- FlowNetC's cost volume and its two input gradients for DOUBLE tensors on the gfx950 fp64 matrix
// cores (v_mfma_f64_16x16x4_f64).  The reference dispatches double first-class (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// correlation_cuda_kernel.cu:386-415, :522-554); until round 6 double tensors took the one-thread-per-output kernel
// (correlation_direct.hip): 630 us forward / 9.2 ms backward at 8 x 256 x 48 x 64 -- 19x / 124x the fp32 kernels (VERDICT r5
// missing #4).  Configuration family: kernel_size 1, stride1 1, stride2 2, pad_size == max_displacement == 20 (FlowNetC.py:28).
//
// The tilings are those of round 1's fp32 matrix-core kernels (correlation_mfma.hip, correlation_mfma_bwd.hip: parity lattice,
// 4 x 4 pixel blocks, one v_mfma 16x16x4 chain per (A block, B block) pair, channels streamed through LDS), re-sized for 8-byte
// elements: 8 channels per forward chunk, a 32-pixel x tile in the backward.  Exact fp64 products and sums (an fma chain per
// output in channel order: the results differ from the reference's only in the order of the 4-channel partial sums).
//
//     forward   out[n, tj*21 + ti, y, x] = (1/C) * sum_c in1[n,c,y,x] * in2[n,c, y + 2(tj-10), x + 2(ti-10)]
//     backward  gI1[n,c,p] = (1/C) * sum_d gO[n, tc(d), p] * in2[n,c, p + 2d],   gI2[n,c,p] = (1/C) * sum_d gO[n, tc(d), p - 2d] * in1[n,c, p - 2d]
#include "corr_params.h"

namespace fn2 {
namespace md {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int DR = 10, D = 2 * DR + 1, NV = 6;   // displacement radius (lattice), planes per axis, B blocks an A block meets per axis
// D of v_mfma_f64_16x16x4_f64: lane l, register r holds D[row = 4 r + (l >> 4)][column = l & 15] -- NOT the fp32 16x16x4
// instruction's 4 (l >> 4) + r (measured: scripts/ubench/mfma_f64_layout.hip prints both indices for every lane and register).
// Rows = the M operand's index (lane & 15 of operand a).
__device__ __forceinline__ int d_row(int lane, int r) { return 4 * r + (lane >> 4); }

// ------------------------------------------------------------------------------------------------ forward
constexpr int TILE_X = 64;                                                  // image pixels per x tile (32 lattice columns per parity)
constexpr int A_ROW = 36, A_PAR = 4 * A_ROW, A_CH = 2 * A_PAR + 16;         // elements: [ch][xpar][row][col]; 304 = 16 mod 32: the two k slots of a
                                                                            // half-wave's 8-byte fragment read (2 x 32 lanes, 64 banks) fall on disjoint halves
constexpr int B_COLS = TILE_X / 2 + 2 * DR;                                 // 52
constexpr int B_ROW = B_COLS, B_PAR = 4 * B_ROW, B_CH = 2 * B_PAR + 16;     // 432 = 16 mod 32
constexpr int FCK = 8;                                                      // channels per chunk (2 k-steps of 4)
constexpr int F_BUF = FCK * (A_CH + B_CH);                                  // 5888 doubles = 47 KB; two buffers
static_assert(A_CH % 32 == 16 && B_CH % 32 == 16, "k-slot halves must be 16 eight-byte units apart");
constexpr int O_RS = 66;                                                    // epilogue x stride
constexpr int O_EL = 8 * D * O_RS + 64;                                     // 8 planes per pass (two passes) + one spare row
constexpr int F_LDS = (2 * F_BUF > O_EL ? 2 * F_BUF : O_EL);
static_assert(F_LDS * 8 <= 163840, "forward LDS budget");

struct FArgs {
    const double *in1, *in2;
    double *out;
    long out_bs;
    double slope;
    int C, H, W, NRG, NXT;
};

// One workgroup (8 waves) = one task (n, y parity, row group rg of 4 lattice rows, B row block u, x tile of 64 pixels): 16 A blocks
// (8 column blocks x 2 x parities) against the row of 13 B column blocks per parity they need; wave w: x parity w & 1, A column
// blocks 2 (w >> 1), 2 (w >> 1) + 1, 7 B fragments for 12 MFMA chains.
__global__ __launch_bounds__(512, 1) void corr_fwd_mfma_f64(FArgs p)
{
    __shared__ __attribute__((aligned(16))) double smem[F_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int u = (int)(t % NV); t /= NV;
    const int xt = (int)(t % p.NXT); t /= p.NXT;
    const int rg = (int)(t % p.NRG); t /= p.NRG;
    const int py = (int)(t & 1u);
    const int n = (int)(t >> 1);
    const int HL = p.H >> 1;
    const int ib0 = 4 * rg - DR + 4 * u;
    const bool all_pad = (ib0 + 3 < 0) || (ib0 >= HL);     // B rows entirely in the zero padding: the epilogue writes zeros
    const long HW = (long)p.H * p.W;
    const double *in1n = p.in1 + (long)n * p.C * HW;
    const double *in2n = p.in2 + (long)n * p.C * HW;
    const int X0 = xt * TILE_X;

    // ---- staging: 16 bytes (an even and an odd pixel) per lane; a wave instruction covers one B row (52 pairs) or two A rows (32 pairs)
    constexpr int KB = FCK / 2, KA = FCK / 4;
    const int sb_x = X0 - 2 * DR + 2 * lane;
    const bool sb_col_ok = (lane < B_COLS) && (sb_x >= 0) && (sb_x < p.W);
    const int sa_piece = lane & 31, sa_sub = lane >> 5;
    const int sa_x = X0 + 2 * sa_piece;
    const int a_ai = ((wave & 1) << 1) + sa_sub;
    const bool a_ok = (sa_x < p.W) && (4 * rg + a_ai < HL);
    const long a_off = (long)(2 * (4 * rg + a_ai) + py) * p.W + sa_x;
    const int b_il = ib0 + (wave & 3);
    const bool b_ok = sb_col_ok && b_il >= 0 && b_il < HL;
    d2 rb[KB], ra[KA];
    auto stage_load = [&](int c0) {
#pragma unroll
        for (int k = 0; k < KB; ++k) {    // B row r = 8k + w -> channel 2k + (w >> 2), row w & 3
            const int ch = c0 + 2 * k + (wave >> 2);
            rb[k] = b_ok ? *reinterpret_cast<const d2 *>(in2n + (long)ch * HW + (long)(2 * b_il + py) * p.W + sb_x) : (d2){0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < KA; ++k) {    // A row r = 16k + 2w + sub -> channel 4k + (w >> 1), row 2 (w & 1) + sub
            const int ch = c0 + 4 * k + (wave >> 1);
            ra[k] = a_ok ? *reinterpret_cast<const d2 *>(in1n + (long)ch * HW + a_off) : (d2){0.0, 0.0};
        }
    };
    auto stage_write = [&](int buf) {
        double *As = smem + buf * F_BUF, *Bs = As + FCK * A_CH;
        if (lane < B_COLS) {
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                double *d = Bs + (2 * k + (wave >> 2)) * B_CH + (wave & 3) * B_ROW + lane;
                d[0] = rb[k][0];
                d[B_PAR] = rb[k][1];
            }
        }
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            double *d = As + (4 * k + (wave >> 1)) * A_CH + a_ai * A_ROW + sa_piece;
            d[0] = ra[k][0];
            d[A_PAR] = ra[k][1];
        }
    };

    // ---- MFMA roles: operand a = A pixels (M), operand b = B pixels (N); lane (pixel fi = lane & 15, k slot fq = lane >> 4)
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) << 1;
    const int fi = lane & 15, fq = lane >> 4;
    const int a_frag = fq * A_CH + xpar * A_PAR + (fi >> 2) * A_ROW + 4 * a0 + (fi & 3);
    const int b_frag = fq * B_CH + xpar * B_PAR + (fi >> 2) * B_ROW + 4 * a0 + (fi & 3);
    d4 acc[2][NV];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[ab][v] = (d4){0.0, 0.0, 0.0, 0.0};
    auto mma_chunk = [&](int buf) {
        const double *As = smem + buf * F_BUF, *Bs = As + FCK * A_CH;
#pragma unroll
        for (int s = 0; s < FCK / 4; ++s) {
            double af[2], bf[NV + 1];
#pragma unroll
            for (int ab = 0; ab < 2; ++ab) af[ab] = As[a_frag + s * 4 * A_CH + 4 * ab];
#pragma unroll
            for (int j = 0; j < NV + 1; ++j) bf[j] = Bs[b_frag + s * 4 * B_CH + 4 * j];
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) acc[ab][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ab], bf[ab + v], acc[ab][v], 0, 0, 0);
        }
    };
    const int nchunks = all_pad ? 0 : p.C / FCK;
    if (nchunks > 0) {
        stage_load(0);
        stage_write(0);
        __syncthreads();
        for (int ck = 0; ck < nchunks; ++ck) {
            const int buf = ck & 1;
            if (ck + 1 < nchunks) stage_load((ck + 1) * FCK);
            mma_chunk(buf);
            if (ck + 1 < nchunks) stage_write(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: accumulators -> LDS [ai][bi][ti][x] (two passes of 8 planes) -> rows of 64 pixels, scaled by 1/C, LeakyReLU fused
    // acc[ab][v][r] = sum for A pixel (row d_row) of block a0 + ab and B pixel fi of block a0 + ab + v (column block offset v - 3 ...)
    {
        double *Os = smem;
        const int e_bi = fi >> 2, e_bj = fi & 3;
        const double fC = (double)p.C;
        const int hx = lane & 31, hr = lane >> 5;
        const int xg = X0 + 2 * hx;
        constexpr int DUMMY = 8 * D * O_RS;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int ab = 0; ab < 2; ++ab)
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int arow = d_row(lane, r), e_ai = arow >> 2, e_aj = arow & 3;
                        const bool mine = (e_ai >> 1) == pass;
                        const int ti = 4 * v + e_bj - e_aj;
                        const int x = 2 * (4 * (a0 + ab) + e_aj) + xpar;
                        const bool ok = mine && ti >= 0 && ti < D;
                        const int addr = ok ? (((e_ai & 1) * 4 + e_bi) * D + ti) * O_RS + x : DUMMY + lane;
                        if (mine) Os[addr] = acc[ab][v][r];
                    }
            __syncthreads();
            for (int pl = wave; pl < 8; pl += 8) {
                const int ai = 2 * pass + (pl >> 2), bi = pl & 3;
                const int tj = 4 * u + bi - ai, IL = 4 * rg + ai;
                if (tj < 0 || tj >= D || IL >= HL) continue;   // wave-uniform
                const int y = 2 * IL + py;
                double *orow = p.out + (long)n * p.out_bs + ((long)tj * D * p.H + y) * p.W + xg;
                const double *srow = Os + (long)pl * D * O_RS + 2 * hx;
                for (int ti0 = 0; ti0 < D; ti0 += 2) {
                    const int ti = ti0 + hr;
                    if (ti < D && xg < p.W) {
                        d2 val = *reinterpret_cast<const d2 *>(srow + ti * O_RS);
                        val[0] = val[0] / fC; val[1] = val[1] / fC;      // sum / nelems (correlation_cuda_kernel.cu:143)
                        if (p.slope != 1.0) { val[0] = val[0] > 0.0 ? val[0] : val[0] * p.slope; val[1] = val[1] > 0.0 ? val[1] : val[1] * p.slope; }
                        *reinterpret_cast<d2 *>(orow + (long)ti * HW) = val;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// g[p, c] = sum_{d'} G[p, d'] * nbr[p + 2d', c]:  nbr = in2, G[p,d'] = gO[tc(d')][p] (FLIP 0: gradInput1);
//                                                nbr = in1, G[p,d'] = gO[tc(-d')][p + 2d'] (FLIP 1: gradInput2).
// M = the 16 centre pixels of an A block (operand a: the G gather), K = the 16 pixels of a neighbour block (4 k-steps of 4 = its 4
// rows), N = 16 channels.  Task = (FLIP, n, y parity, row group of 4 centre rows, x tile of 32 pixels, group of 32 channels): 8 waves
// = x parity x centre column block (4 per parity in 32 pixels); the workgroup loops over the 6 neighbour row blocks u (no atomics,
// deterministic) and per u over the 2 channel tiles.
constexpr int BT_X = 32;                                                    // centre pixels per x tile (16 lattice columns per parity)
constexpr int N_COLS = BT_X / 2 + 2 * DR;                                   // 36 neighbour lattice columns incl. halo
constexpr int N_ROW = N_COLS, N_PAR = 4 * N_ROW, N_CH = 2 * N_PAR + 2;     // 290 = 2 mod 32: the 16 channels x 2 k slots of a half-wave read 32 distinct 8-byte banks
constexpr int BCK = 16;                                                     // channels per tile (MFMA N)
constexpr int N_EL = BCK * N_CH;                                            // 4640 doubles = 37 KB per buffer; two buffers (channel tile ct in buffer ct & 1)
// G tile [plane = 4 ai + bi][ti][x]: row stride 33, plane stride 697 = 1 mod 8 -- the gather of a half-wave (16 centre pixels x 2 k slots)
// then collides 2-way at most (4-way with even strides; no affine layout is conflict-free: a pixel's column and its displacement
// column move together)
constexpr int G_RS = BT_X + 1, G_PS = D * G_RS + 4;
constexpr int G_EL = 16 * G_PS;                                             // 11152 doubles = 89 KB
static_assert(G_PS % 8 == 1 && G_RS % 32 == 1, "gather bank pattern");
constexpr int BNCT = 4, BCG = BNCT * BCK;                                   // channel tiles / channels per task
constexpr int E_RS = BT_X + 1;
constexpr int E_EL = BCG * 4 * E_RS;
static_assert(BNCT % 2 == 0 && E_EL <= G_EL && (G_EL + 2 * N_EL) * 8 <= 163840, "backward LDS budget");

struct BArgs {
    const double *nbr[2];
    const double *gout;
    double *gin[2];
    int B, C, H, W, NRG, NXT, NCG;
};

__global__ __launch_bounds__(512, 1) void corr_bwd_mfma_f64(BArgs p)
{
    __shared__ __attribute__((aligned(16))) double smem[G_EL + 2 * N_EL];
    double *Gs = smem, *Ns = smem + G_EL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(t % p.NCG); t /= p.NCG;
    const int xt = (int)(t % p.NXT); t /= p.NXT;
    const int rg = (int)(t % p.NRG); t /= p.NRG;
    const int py = (int)(t & 1u); t >>= 1;
    const int n = (int)(t % p.B);
    const int FLIP = __builtin_amdgcn_readfirstlane((int)(t / p.B));
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int X0 = xt * BT_X;
    const int c_base = cg * BCG;
    const double *nbr_n = p.nbr[FLIP] + ((long)n * p.C + c_base) * HW;
    const double *go_n = p.gout + (long)n * D * D * HW;

    // ---- neighbour tile staging: wave w stages row bi = w & 3 of channels 2k + (w >> 2), k = 0..7; lane = lattice column (< 36)
    const int s_bi = wave & 3;
    const int s_xb = X0 - 2 * DR + 2 * lane;
    const bool s_col_ok = (lane < N_COLS) && (s_xb >= 0) && (s_xb < p.W);
    d2 rn[8];
    auto nbr_load = [&](int u, int ct) {
        const int il = 4 * rg - DR + 4 * u + s_bi;
        const bool ok = s_col_ok && il >= 0 && il < HL;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            rn[k] = ok ? *reinterpret_cast<const d2 *>(nbr_n + (long)(ct * BCK + 2 * k + (wave >> 2)) * HW + (long)(2 * il + py) * p.W + s_xb) : (d2){0.0, 0.0};
    };
    auto nbr_write = [&](int buf) {
        if (lane < N_COLS) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double *d = Ns + buf * N_EL + (2 * k + (wave >> 2)) * N_CH + s_bi * N_ROW + lane;
                d[0] = rn[k][0];
                d[N_PAR] = rn[k][1];
            }
        }
    };
    // ---- G tile staging: rows (plane = 4 ai + bi, ti) of 32 pixels; wave w takes planes w, w + 8; lane = (row-in-quad hr, x pair hx)
    const int hx = lane & 15, hr = lane >> 4;
    constexpr int NT = (D + 3) / 4;          // ti = hr + 4 i
    d2 rg_[2][NT];
    // the loads of a G tile -- all 2 x NT issued before the first LDS write: one load latency per u -- and its LDS writes
    auto g_load = [&](int u) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pl = wave + 8 * h, ai = pl >> 2, bi = pl & 3;
            const int tj = 4 * u + bi - ai, IL = 4 * rg + ai;
            const bool row_ok = tj >= 0 && tj < D && IL < HL;
            const int x = X0 + 2 * hx;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int ti = hr + 4 * i;
                d2 v = (d2){0.0, 0.0};
                if (!FLIP) {
                    if (row_ok && ti < D && x < p.W) v = *reinterpret_cast<const d2 *>(go_n + ((long)(tj * D + ti) * p.H + 2 * IL + py) * p.W + x);
                } else {     // G'[tj, ti][p] = gO[(20 - tj) * 21 + (20 - ti)][p + 2 d'],  d' = (tj - 10, ti - 10)
                    const int ys = 2 * IL + py + 2 * (tj - DR), xs = x + 2 * (ti - DR);
                    if (row_ok && ti < D && ys >= 0 && ys < p.H && xs >= 0 && xs < p.W)
                        v = *reinterpret_cast<const d2 *>(go_n + ((long)((2 * DR - tj) * D + (2 * DR - ti)) * p.H + ys) * p.W + xs);
                }
                rg_[h][i] = v;
            }
        }
    };
    auto g_write = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pl = wave + 8 * h;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int ti = hr + 4 * i;
                if (ti < D) { double *d = Gs + pl * G_PS + ti * G_RS + 2 * hx; d[0] = rg_[h][i][0]; d[1] = rg_[h][i][1]; }   // (rows are 8-byte aligned only)
            }
        }
    };

    // ---- MFMA roles: operand a (G): lane (centre pixel i = lane & 15 = (ai, aj), k slot q = lane >> 4 = bj); k-step = neighbour row s = bi
    //                  operand b (neighbours): lane (k slot q = bj, channel j = lane & 15)
    const int xpar = wave & 1, a0 = wave >> 1;          // centre column block (one per wave)
    const int fi = lane & 15, fq = lane >> 4;
    const int g_ai = fi >> 2, g_aj = fi & 3;
    const int n_frag = fi * N_CH + xpar * N_PAR + 4 * a0 + fq;   // + s * N_ROW + 4 v
    d4 acc[BNCT];
#pragma unroll
    for (int ct = 0; ct < BNCT; ++ct) acc[ct] = (d4){0.0, 0.0, 0.0, 0.0};
    int u_lo = 0, u_hi = NV - 1;
    while (u_lo < NV && (4 * rg - DR + 4 * u_lo + 3 < 0)) ++u_lo;
    while (u_hi >= 0 && (4 * rg - DR + 4 * u_hi >= HL)) --u_hi;
    // the G operand of (neighbour block v, neighbour row s): Gs[(g_ai * 4 + s) * G_PS + ti_v * G_RS + x], ti_v = 4 v + fq - g_aj; entries
    // outside the 21-wide band read column 0 and are replaced by zero.  Gathered per (channel tile, s) from LDS -- six 8-byte reads
    // in front of six 64-cycle MFMAs -- instead of once per u into 48 registers: those registers hold the NEXT u's G tile, whose
    // loads are then in flight during this u's MFMAs (both at once spill: 256 registers per lane at two waves per SIMD).
    int g_off[NV];
    bool g_ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int ti = 4 * v + fq - g_aj;
        g_ok[v] = ti >= 0 && ti < D;
        g_off[v] = g_ai * 4 * G_PS + (g_ok[v] ? ti : 0) * G_RS + 2 * (4 * a0 + g_aj) + xpar;
    }
    if (u_lo <= u_hi) { nbr_load(u_lo, 0); g_load(u_lo); }
    for (int u = u_lo; u <= u_hi; ++u) {
        g_write();
        nbr_write(0);
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < BNCT; ++ct) {
            // in flight during this tile's MFMAs: the next channel tile of u (and, once per u, the next u's G tile), then the next u's first tile
            if (ct + 1 < BNCT) nbr_load(u, ct + 1);
            else if (u < u_hi) nbr_load(u + 1, 0);
            if (ct == 0 && u < u_hi) g_load(u + 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double nf[NV], gf[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double gv = Gs[g_off[v] + s * G_PS];
                    gf[v] = g_ok[v] ? gv : 0.0;
                    nf[v] = Ns[(ct & 1) * N_EL + n_frag + s * N_ROW + 4 * v];
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(gf[v], nf[v], acc[ct], 0, 0, 0);
            }
            if (ct + 1 < BNCT) nbr_write((ct + 1) & 1);   // the other buffer: last read two tiles ago
            __syncthreads();                              // this tile has been read, the next one is complete
        }
    }
    // ---- epilogue: acc[ct][r] = g[centre pixel d_row(lane, r) of block a0][channel 16 ct + (lane & 15)] -> LDS [ch][ai][x] -> rows
    {
        double *Es = smem;
#pragma unroll
        for (int ct = 0; ct < BNCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int arow = d_row(lane, r), e_ai = arow >> 2, e_aj = arow & 3;
                Es[((ct * BCK + fi) * 4 + e_ai) * E_RS + 2 * (4 * a0 + e_aj) + xpar] = acc[ct][r];
            }
        __syncthreads();
        const double fC = (double)p.C;
        double *gin_n = p.gin[FLIP] + ((long)n * p.C + c_base) * HW;
        const int xl = lane & 31, rsub = lane >> 5;
        for (int R = 2 * wave + rsub; R < BCG * 4; R += 16) {
            const int ch = R >> 2, ai = R & 3, IL = 4 * rg + ai;
            if (IL >= HL || X0 + xl >= p.W) continue;
            gin_n[(long)ch * HW + (long)(2 * IL + py) * p.W + X0 + xl] = Es[R * E_RS + xl] / fC;   // sum / nelems (correlation_cuda_kernel.cu:238, :331)
        }
    }
}

} // namespace md

bool corr_mfma_f64_applicable(int dtype, int C, int H, int W, int pad, int k, int md_, int s1, int s2)
{
    if (dtype != FN2_F64) return false;
    if (k != 1 || s1 != 1 || s2 != 2 || pad != md_ || md_ != 2 * md::DR) return false;
    if (C % md::BCG != 0 || (H & 1) || (W & 1)) return false;
    if ((long)md::D * md::D * H * W >= 0x7fffffffL / 8) return false;
    return true;
}

int corr_forward_mfma_f64(const double *in1, const double *in2, double *out, long out_bs, double slope, int B, int C, int H, int W, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(out, 16) || (out_bs % 2) != 0) return FN2_EALIGN;
    md::FArgs a;
    a.in1 = in1; a.in2 = in2; a.out = out; a.out_bs = out_bs; a.slope = slope;
    a.C = C; a.H = H; a.W = W;
    a.NRG = (H / 2 + 3) / 4; a.NXT = (W + md::TILE_X - 1) / md::TILE_X;
    const long ntasks = (long)B * 2 * a.NRG * a.NXT * md::NV;
    if (ntasks == 0) return FN2_OK;
    if (ntasks > 0x3fffffffL) return FN2_EINVAL;
    hipLaunchKernelGGL(md::corr_fwd_mfma_f64, dim3((unsigned)ntasks), dim3(512), 0, s, a);
    return launch_status();
}

int corr_backward_mfma_f64(const double *in1, const double *in2, const double *gout, double *g1, double *g2, int B, int C, int H, int W, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(gout, 16) || !aligned(g1, 8) || !aligned(g2, 8)) return FN2_EALIGN;
    md::BArgs a;
    a.nbr[0] = in2; a.nbr[1] = in1; a.gout = gout; a.gin[0] = g1; a.gin[1] = g2;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.NRG = (H / 2 + 3) / 4; a.NXT = (W + md::BT_X - 1) / md::BT_X; a.NCG = C / md::BCG;
    const long ntasks = 2L * B * 2 * a.NRG * a.NXT * a.NCG;
    if (ntasks == 0) return FN2_OK;
    if (ntasks > 0x3fffffffL) return FN2_EINVAL;
    hipLaunchKernelGGL(md::corr_bwd_mfma_f64, dim3((unsigned)ntasks), dim3(512), 0, s, a);
    return launch_status();
}

} // namespace fn2
