// correlation_mfma.hip -- FlowNetC cost volume (correlation forward) on the gfx950 matrix cores.
//
// Replaces reference kernels channels_first + correlation_forward
// (correlation_cuda_kernel.cu:46-70, :73-147) for the configuration family FlowNetC uses
// (FlowNetC.py:28,31): kernel_size = 1, stride1 = 1, stride2 = 2, pad_size == max_displacement,
// fp32.  For those parameters
//     out[n, tj*D + ti, y, x] = (1/C) * sum_c in1[n,c,y,x] * in2[n,c, y + 2(tj-dr), x + 2(ti-dr)]
// with in2 read as 0 outside the image, dr = md/2, D = 2dr+1.
//
// Mapping to MFMA.  stride2 = 2 keeps pixel parity: pixel (y, x) only meets pixels of the same
// (y&1, x&1) class, so each class is a dense lattice (I, J) = (y>>1, x>>1) on which the
// displacement window is the contiguous (2dr+1)^2 box.  A 4x4 block of lattice pixels of in1
// ("A block", 16 pixels = the M rows of a 16x16 tile) against a 4x4 block of lattice pixels of
// in2 ("B block", the N columns) over K = channels is one v_mfma_f32_16x16x4_f32 chain: exact
// fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD.  An A block needs the B blocks covering
// [-dr, 3+dr]^2 around it: NV x NV blocks, NV = 1 + ceil(dr/2) (6 for dr = 10), of which the
// (2dr+1)^2/(4 NV)^2 = 76.6 % band is kept.
//
// Work decomposition.  One workgroup (8 waves) = one task (n, y-parity, row group rg of 4 lattice
// rows, B row block u, x tile of 64 pixels).  It owns 16 A blocks (8 column blocks x 2 x-parities)
// and the one row of 13 B column blocks per x-parity they need; wave w takes x-parity w&1 and A
// column blocks 2(w>>1), 2(w>>1)+1, whose 2 x NV B blocks overlap in all but one (7 B fragments
// for 12 MFMA chains).  Channels stream through LDS in chunks of 16, double-buffered:
//     global (NCHW, rows of W floats, coalesced 8 B/lane) -> registers -> LDS [ch][xpar][row][col]
// de-interleaving the x parity on the way, zero-filling everything outside the image.  The LDS
// strides (A: row 36, channel 304; B: row 52, channel 432 floats) make every ds_read_b32 of an
// MFMA operand conflict-free: the 16 pixels of a block hit 16 distinct banks and the second
// k-slot of the same 32-lane group is shifted by 16 banks.
// Epilogue: accumulators -> LDS as [A row][B row][ti][x] (x stride 65) -> each output row
// (n, channel, y, 64 consecutive x) leaves as one coalesced 256 B store.  The displacement-major
// NCHW output would otherwise be written as isolated 4-byte stores, as in the reference.
//
// HBM traffic: in1/in2 tiles are re-read by the NV row-block tasks and by neighbouring row
// groups, but those re-reads are served by L2 / Infinity Cache (both inputs total 50 MB at the
// FlowNetC shape); algorithmic bytes are 2 * B*C*H*W*4 read + B*D*D*H*W*4 written.
#include "corr_params.h"

namespace fn2 {

namespace mf {

constexpr int CK = 16;            // channels per LDS chunk
constexpr int TILE_X = 64;        // image pixels per x tile (32 lattice columns per parity)
constexpr int DR_MAX = 10;        // largest displacement radius (lattice units) this kernel handles
constexpr int NV_MAX = 6;         // 1 + ceil(DR_MAX / 2)
constexpr int A_ROW = 36, A_PAR = 4 * A_ROW, A_CH = 2 * A_PAR + 16;          // 144, 304
constexpr int B_COLS = TILE_X / 2 + 2 * DR_MAX;                               // 52
constexpr int B_ROW = B_COLS, B_PAR = 4 * B_ROW, B_CH = 2 * B_PAR + 16;      // 208, 432
constexpr int A_FLOATS = CK * A_CH, B_FLOATS = CK * B_CH;                    // 4864, 6912
constexpr int BUF_FLOATS = A_FLOATS + B_FLOATS;                              // 11776
constexpr int O_RS = 65;                                                      // epilogue x stride
constexpr int LDS_FLOATS = 2 * BUF_FLOATS;                                   // 23552 = 94208 B
static_assert(16 * (2 * DR_MAX + 1) * O_RS <= LDS_FLOATS, "epilogue staging must fit in the operand buffers");
static_assert(A_CH % 32 == 16 && B_CH % 32 == 16, "k-slot halves must be 16 banks apart");

typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(2))) f2;

struct Args {
    const float *in1, *in2;
    float *out;
    int C, H, W;     // H, W even
    int dr, D, NV;   // displacement radius (lattice), 2dr+1, B blocks per A block per axis
    int NRG, NXT;    // row groups per parity, x tiles
};

__global__ __launch_bounds__(512, 2) void corr_fwd_mfma_f32(Args p)
{
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- task decode (u fastest: the NV tasks sharing one A row group run back to back)
    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int u = (int)(t % p.NV); t /= p.NV;
    const int xt = (int)(t % p.NXT); t /= p.NXT;
    const int rg = (int)(t % p.NRG); t /= p.NRG;
    const int py = (int)(t & 1u);
    const int n = (int)(t >> 1);

    const int HL = p.H >> 1;                        // lattice rows per parity
    const int ib0 = 4 * rg - p.dr + 4 * u;          // first B lattice row of this task
    // B rows entirely in the zero padding: the products are exact zeros; skip the channel loop and
    // let the epilogue write them (the output is never pre-zeroed).
    const bool all_pad = (ib0 + 3 < 0) || (ib0 >= HL);

    const long HW = (long)p.H * p.W;
    const float *in1n = p.in1 + (long)n * p.C * HW;
    const float *in2n = p.in2 + (long)n * p.C * HW;
    const int X0 = xt * TILE_X;

    // ---- staging roles (fixed per thread for the whole task)
    // B: wave w stages B row bi = w&3 of channels c0 + 2k + (w>>2), k = 0..7; lane = lattice column jb.
    const int s_bi = wave & 3;
    const int s_jb = lane;                                       // < B_COLS active
    const int s_yb = 2 * (ib0 + s_bi) + py;                      // image row
    const int s_xb = X0 - 2 * p.dr + 2 * s_jb;                   // image x of the even element
    const bool b_ok = (s_jb < B_COLS) && (ib0 + s_bi >= 0) && (ib0 + s_bi < HL) && (s_xb >= 0) && (s_xb < p.W);
    const float *b_src = in2n + (long)(wave >> 2) * HW + (long)s_yb * p.W + s_xb;
    const int b_dst = (wave >> 2) * B_CH + s_bi * B_ROW + s_jb; // + k*2*B_CH, + B_PAR for the odd element
    // A: wave w stages A rows ai = 2(w&1) + (lane>>5) of channels c0 + 4k + (w>>1), k = 0..3; lane&31 = ja.
    const int s_ai = ((wave & 1) << 1) + (lane >> 5);
    const int s_ja = lane & 31;
    const int s_ya = 2 * (4 * rg + s_ai) + py;
    const int s_xa = X0 + 2 * s_ja;
    const bool a_ok = (4 * rg + s_ai < HL) && (s_xa < p.W);
    const float *a_src = in1n + (long)(wave >> 1) * HW + (long)s_ya * p.W + s_xa;
    const int a_dst = (wave >> 1) * A_CH + s_ai * A_ROW + s_ja;  // + k*4*A_CH, + A_PAR for the odd element

    f2 rb[8], ra[4];
    auto stage_load = [&](int c0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f2 v = {0.0f, 0.0f};
            if (b_ok) v = *reinterpret_cast<const f2 *>(b_src + (long)(c0 + 2 * k) * HW);
            rb[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f2 v = {0.0f, 0.0f};
            if (a_ok) v = *reinterpret_cast<const f2 *>(a_src + (long)(c0 + 4 * k) * HW);
            ra[k] = v;
        }
    };
    auto stage_write = [&](int buf) {
        float *As = smem + buf * BUF_FLOATS;
        float *Bs = As + A_FLOATS;
        if (s_jb < B_COLS) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                Bs[b_dst + k * 2 * B_CH] = rb[k][0];
                Bs[b_dst + k * 2 * B_CH + B_PAR] = rb[k][1];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            As[a_dst + k * 4 * A_CH] = ra[k][0];
            As[a_dst + k * 4 * A_CH + A_PAR] = ra[k][1];
        }
    };

    // ---- MFMA roles
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) << 1;                 // first of this wave's two A column blocks
    const int fi = lane & 15, fq = lane >> 4;        // pixel-in-block, k slot
    const int a_frag = fq * A_CH + xpar * A_PAR + (fi >> 2) * A_ROW + 4 * a0 + (fi & 3);
    const int b_frag = fq * B_CH + xpar * B_PAR + (fi >> 2) * B_ROW + 4 * a0 + (fi & 3);

    f4 acc[2][NV_MAX];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int v = 0; v < NV_MAX; ++v) acc[ab][v] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

    const int nchunks = all_pad ? 0 : p.C / CK;
    if (nchunks > 0) {
    stage_load(0);
    stage_write(0);
    __syncthreads();
    for (int ck = 0; ck < nchunks; ++ck) {
        const int buf = ck & 1;
        if (ck + 1 < nchunks) stage_load((ck + 1) * CK);
        const float *As = smem + buf * BUF_FLOATS;
        const float *Bs = As + A_FLOATS;
#pragma unroll
        for (int s = 0; s < CK / 4; ++s) {
            float af[2], bf[NV_MAX + 1];
#pragma unroll
            for (int ab = 0; ab < 2; ++ab) af[ab] = As[a_frag + s * 4 * A_CH + 4 * ab];
#pragma unroll
            for (int j = 0; j < NV_MAX + 1; ++j) bf[j] = Bs[b_frag + s * 4 * B_CH + 4 * j];
#pragma unroll
            for (int v = 0; v < NV_MAX; ++v) {
                if (v < p.NV) { // wave-uniform
#pragma unroll
                    for (int ab = 0; ab < 2; ++ab)
                        acc[ab][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ab], bf[ab + v], acc[ab][v], 0, 0, 0);
                }
            }
        }
        if (ck + 1 < nchunks) stage_write(buf ^ 1);
        __syncthreads();
    }
    }

    // ---- epilogue: accumulators -> LDS [ai][bi][ti][x] -> coalesced rows
    // D layout of v_mfma_f32_16x16x4_f32: lane l, register r holds D[row = 4*(l>>4) + r][col = l&15];
    // rows are A pixels (ai = row>>2 = l>>4, aj = row&3 = r), columns B pixels (bi = fi>>2, bj = fi&3).
    {
        float *Os = smem;
        const int e_ai = fq, e_bi = fi >> 2, e_bj = fi & 3;
        const int plane = (e_ai * 4 + e_bi) * p.D;
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int v = 0; v < NV_MAX; ++v) {
                if (v < p.NV) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ti = 4 * v + e_bj - r;
                        if (ti >= 0 && ti < p.D) {
                            const int x = 2 * (4 * (a0 + ab) + r) + xpar;
                            Os[(plane + ti) * O_RS + x] = acc[ab][v][r];
                        }
                    }
                }
            }
        __syncthreads();
        const float fC = (float)p.C;
        const int nrows = 16 * p.D;
        const int xg = X0 + lane;
        for (int R = wave; R < nrows; R += 8) {
            const int ti = R % p.D;
            const int pl = R / p.D;
            const int ai = pl >> 2, bi = pl & 3;
            const int tj = 4 * u + bi - ai;
            const int IL = 4 * rg + ai;
            if (tj < 0 || tj >= p.D || IL >= HL) continue; // wave-uniform
            const int y = 2 * IL + py;
            if (xg < p.W) {
                const float val = Os[R * O_RS + lane];
                p.out[(((long)n * p.D * p.D + (long)tj * p.D + ti) * p.H + y) * p.W + xg] = val / fC;
            }
        }
    }
}

} // namespace mf

// Preconditions of the MFMA forward path.
bool corr_mfma_f32_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    if (dtype != FN2_F32) return false;
    if (k != 1 || s1 != 1 || s2 != 2 || pad != md) return false;
    if (md / 2 > mf::DR_MAX || md < 2) return false;
    if (C % mf::CK != 0 || (H & 1) || (W & 1)) return false;
    return true;
}

int corr_forward_mfma_f32(const float *in1, const float *in2, float *out, int B, int C, int H, int W, int md,
                          hipStream_t s)
{
    if (!aligned(in1, 8) || !aligned(in2, 8)) return FN2_EALIGN;
    mf::Args a;
    a.in1 = in1; a.in2 = in2; a.out = out;
    a.C = C; a.H = H; a.W = W;
    a.dr = md / 2; a.D = 2 * a.dr + 1; a.NV = 1 + (a.dr + 1) / 2;
    const int HL = H / 2;
    a.NRG = (HL + 3) / 4;
    a.NXT = (W + mf::TILE_X - 1) / mf::TILE_X;
    const long ntasks = (long)B * 2 * a.NRG * a.NXT * a.NV;
    if (ntasks == 0) return FN2_OK;
    hipLaunchKernelGGL(mf::corr_fwd_mfma_f32, dim3((unsigned)ntasks), dim3(512), 0, s, a);
    return launch_status();
}

} // namespace fn2
