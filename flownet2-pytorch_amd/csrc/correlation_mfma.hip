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
// Epilogue: accumulators -> LDS as [A row][B row][ti][x] (x stride 66) -> each output row
// (n, channel, y, 64 consecutive x) leaves as one coalesced 256 B store.  The displacement-major
// NCHW output would otherwise be written as isolated 4-byte stores, as in the reference.
//
// HBM traffic: in1/in2 tiles are re-read by the NV row-block tasks and by neighbouring row
// groups, but those re-reads are served by L2 / Infinity Cache (both inputs total 50 MB at the
// FlowNetC shape); algorithmic bytes are 2 * B*C*H*W*4 read + B*D*D*H*W*4 written.
#include <type_traits>

#include "corr_params.h"
#include "bf16x3.h"

namespace fn2 {

namespace mf {

constexpr int TILE_X = 64;        // image pixels per x tile (32 lattice columns per parity)
constexpr int DR_MAX = 10;        // largest displacement radius (lattice units) this kernel handles
constexpr int NV_MAX = 6;         // 1 + ceil(DR_MAX / 2)
constexpr int A_ROW = 36, A_PAR = 4 * A_ROW, A_CH = 2 * A_PAR + 16;          // 144, 304
constexpr int B_COLS = TILE_X / 2 + 2 * DR_MAX;                               // 52
constexpr int B_ROW = B_COLS, B_PAR = 4 * B_ROW, B_CH = 2 * B_PAR + 16;      // 208, 432
constexpr int O_RS = 66;                                                      // epilogue x stride (8 B aligned rows, <= 2-way write conflicts)
static_assert(A_CH % 32 == 16 && B_CH % 32 == 16, "k-slot halves must be 16 banks apart");

typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(2))) f2;

struct Args {
    const float *in1, *in2;
    float *out;
    long out_bs;     // elements between batch items of `out` (D*D*H*W unless out is a channel slice of a larger buffer)
    float slope;     // LeakyReLU negative slope fused into the epilogue (1 = none)
    int C, H, W;     // H, W even
    int dr, D, NV;   // displacement radius (lattice), 2dr+1, B blocks per A block per axis
    int NRG, NXT;    // row groups per parity, x tiles
};

// Tunables of one instantiation.
//   CK    channels per LDS chunk (multiple of 4)
//   NBUF  1: one operand buffer, two barriers per chunk; 2: double-buffered, one barrier per chunk
//   EP    epilogue passes (1: all 16 (ai,bi) planes at once = 87 KB of LDS; 2: 8 planes per pass)
//   WPS   waves per SIMD the register budget is sized for (= workgroups per CU * 2)
//   VAR   ablation switches for profiling only (0 = the real kernel): 1 no MFMA, 2 no staging, 4 no stores,
//         8 no LDS fragment reads, 16 every chunk re-reads chunk 0 (L2-hot inputs),
//         32 no global loads, 64 no LDS staging writes
template <int CK, int NBUF, int EP>
struct Cfg {
    static constexpr int A_FLOATS = CK * A_CH, B_FLOATS = CK * B_CH, BUF_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int O_FLOATS = (16 / EP) * (2 * DR_MAX + 1) * O_RS + 64;   // + one spare row
    static constexpr int LDS_FLOATS = (NBUF * BUF_FLOATS > O_FLOATS) ? NBUF * BUF_FLOATS : O_FLOATS;
};

// ---- epilogue shared by the forward kernels: accumulators -> LDS [ai][bi][ti][x] -> coalesced rows
// D layout of v_mfma_f32_16x16x4_f32: lane l, register r holds D[row = 4*(l>>4) + r][col = l&15];
// rows are A pixels (ai = row>>2 = l>>4, aj = row&3 = r), columns B pixels (bi = fi>>2, bj = fi&3).
// The caller guarantees that every wave has finished reading the operand buffers that alias `smem`.
// NAB A blocks per wave, NW = 16 / NAB waves per workgroup (wave = (x parity, group of NAB A column blocks))
template <int NV, int EP, int VAR, int NAB = 2>
__device__ __forceinline__ void epilogue(float *smem, f4 (&acc)[NAB][NV], const Args &p, int lane, int wave, int n,
                                         int py, int rg, int u, int X0, int HL, long HW)
{
    constexpr int NW = 16 / NAB;
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) * NAB;
    const int fi = lane & 15, fq = lane >> 4;
    {
        float *Os = smem;
        constexpr int AI_PER_PASS = 4 / EP;
        constexpr int DUMMY = (16 / EP) * (2 * DR_MAX + 1) * O_RS;   // one spare row: sink for out-of-band entries
        const int e_ai = fq, e_bi = fi >> 2, e_bj = fi & 3;
        // acc / nelems as the reference forms it (correlation_cuda_kernel.cu:143); a power-of-two
        // channel count makes the reciprocal multiply exact, otherwise divide.
        const float fC = (float)p.C;
        const bool pow2 = (p.C & (p.C - 1)) == 0;
        const float rC = 1.0f / fC;
        const int hx = lane & 31, hr = lane >> 5;   // write-out: lane = (row-in-pair, x pair)
        const int xg = X0 + 2 * hx;
#pragma unroll
        for (int pass = 0; pass < EP; ++pass) {
            const bool mine = (EP == 1) || ((e_ai / AI_PER_PASS) == pass);
            const int plane = ((e_ai % AI_PER_PASS) * 4 + e_bi) * p.D;
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ti = 4 * v + e_bj - r;
                        const int x = 2 * (4 * (a0 + ab) + r) + xpar;
                        bool ok = mine;
                        if (v == 0) ok = ok && (ti >= 0);            // only the first and the last two block columns
                        if (v >= NV - 2) ok = ok && (ti < p.D);      // can fall outside the displacement band
                        const int addr = ok ? (plane + ti) * O_RS + x : DUMMY + lane;
                        if (mine) Os[addr] = acc[ab][v][r];
                    }
            __syncthreads();
            // each wave writes whole planes: rows (plane, ti) for ti = 0..D-1, two rows per instruction,
            // 8 B per lane -> one 256 B contiguous segment per row
            for (int pl = wave; pl < 4 * AI_PER_PASS; pl += NW) {
                const int ai = pass * AI_PER_PASS + (pl >> 2), bi = pl & 3;
                const int tj = 4 * u + bi - ai;
                const int IL = 4 * rg + ai;
                if (tj < 0 || tj >= p.D || IL >= HL) continue; // wave-uniform
                const int y = 2 * IL + py;
                float *orow = p.out + (long)n * p.out_bs + ((long)tj * p.D * p.H + y) * p.W + xg;
                const float *srow = Os + (long)pl * p.D * O_RS + 2 * hx;
                for (int ti0 = 0; ti0 < p.D; ti0 += 2) {
                    const int ti = ti0 + hr;
                    if (ti < p.D && xg < p.W && !(VAR & 4)) {
                        f2 val = *reinterpret_cast<const f2 *>(srow + ti * O_RS);
                        if (pow2) { val[0] *= rC; val[1] *= rC; }
                        else { val[0] /= fC; val[1] /= fC; }
                        if (p.slope != 1.0f) {   // fused LeakyReLU (FlowNetC.py:87), wave-uniform branch
                            val[0] = val[0] > 0.0f ? val[0] : val[0] * p.slope;
                            val[1] = val[1] > 0.0f ? val[1] : val[1] * p.slope;
                        }
                        *reinterpret_cast<f2 *>(orow + (long)ti * HW) = val;   // (non-temporal stores: no measurable difference)
                    }
                }
            }
            if (pass + 1 < EP) __syncthreads();
        }
    }
}

template <int NV, int CK, int NBUF, int EP, int WPS, int VAR, bool X4>
__global__ __launch_bounds__(512, WPS) void corr_fwd_mfma_f32(Args p)
{
    typedef Cfg<CK, NBUF, EP> G;
    static_assert(CK % 8 == 0 && (CK * 4) % 8 == 0, "staging assigns whole rows to waves");
    __shared__ __attribute__((aligned(16))) float smem[G::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- task decode (u fastest: the NV tasks sharing one A row group run back to back)
    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int u = (int)(t % NV); t /= NV;
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

    // ---- staging roles (fixed per thread for the whole task).  A chunk has CK*4 B rows (one per
    // (channel, bi)) of 104 pixels (52 lattice columns x 2 parities) and CK*4 A rows of 64 pixels.
    // X4 (dr even, W % 4 == 0): 16 B per lane -- a wave instruction covers 2 B rows (26 pieces each) or
    // 4 A rows (16 pieces each); a piece (e,o,e,o) is de-interleaved into two ds_write_b64.
    // otherwise: 8 B per lane, one B row (52 pairs) or two A rows (32 pairs) per wave instruction.
    constexpr int KB = X4 ? CK / 4 : CK / 2;   // B load instructions per wave per chunk
    constexpr int KA = X4 ? CK / 8 : CK / 4;   // A load instructions per wave per chunk
    static_assert(!X4 || CK % 8 == 0, "X4 staging assigns 8 rows per wave");
    typedef typename std::conditional<X4, f4, f2>::type ld_t;
    // B
    const int sb_piece = X4 ? (lane & 31) : lane;                    // piece / pair index within the row
    const int sb_sub = X4 ? (lane >> 5) : 0;                         // row within the instruction
    const int sb_x = X0 - 2 * p.dr + (X4 ? 4 : 2) * sb_piece;        // image x of the first element
    const bool sb_col_ok = (sb_piece < (X4 ? B_COLS / 2 : B_COLS)) && (sb_x >= 0) && (sb_x < p.W);
    // A
    const int sa_piece = X4 ? (lane & 15) : (lane & 31);
    const int sa_sub = X4 ? (lane >> 4) : (lane >> 5);
    const int sa_x = X0 + (X4 ? 4 : 2) * sa_piece;
    const bool sa_col_ok = (sa_x < p.W);
    // row -> (channel within chunk, lattice row) for instruction k:
    //   X4:  B row r = 8w + 2k + sub  -> ch = 2w + (k>>1), bi = 2(k&1) + sub ;  A row r = 8w + 4k + sub -> ch = 2w + k, ai = sub
    //   else B row r = 8k + w        -> ch = 2k + (w>>2), bi = w&3          ;  A row r = 16k + 2w + sub -> ch = 4k + (w>>1), ai = 2(w&1)+sub
    auto b_ch = [&](int k) { return X4 ? 2 * wave + (k >> 1) : 2 * k + (wave >> 2); };
    auto b_bi = [&](int k) { return X4 ? 2 * (k & 1) + sb_sub : (wave & 3); };
    auto a_ch = [&](int k) { return X4 ? 2 * wave + k : 4 * k + (wave >> 1); };
    const int a_ai = X4 ? sa_sub : ((wave & 1) << 1) + sa_sub;
    const bool a_ok = sa_col_ok && (4 * rg + a_ai < HL);
    const long a_off = (long)(2 * (4 * rg + a_ai) + py) * p.W + sa_x;

    ld_t rb[KB], ra[KA];
    auto stage_load = [&](int c0) {
        if (VAR & 2) return;
        if (VAR & 16) c0 = 0;   // profiling: every chunk re-reads chunk 0 (always L2-resident)
        if (VAR & 32) {         // profiling: no global loads (the LDS writes still happen)
#pragma unroll
            for (int k = 0; k < KB; ++k) rb[k] = (ld_t)(1.0f);
#pragma unroll
            for (int k = 0; k < KA; ++k) ra[k] = (ld_t)(3.0f);
            return;
        }
        // Every lane loads (out-of-image lanes from a valid dummy address); the zeroing select and the parity
        // de-interleave happen in stage_write, i.e. in program order AFTER the MFMA phase and behind its
        // sched_barriers, so nothing forces a wait for these loads before the MFMAs have been issued.
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int il = ib0 + b_bi(k);
            const bool ok = sb_col_ok && (il >= 0) && (il < HL);
            rb[k] = *reinterpret_cast<const ld_t *>(in2n + (ok ? (long)(c0 + b_ch(k)) * HW + (long)(2 * il + py) * p.W + sb_x : 0));
        }
#pragma unroll
        for (int k = 0; k < KA; ++k)
            ra[k] = *reinterpret_cast<const ld_t *>(in1n + (a_ok ? (long)(c0 + a_ch(k)) * HW + a_off : 0));
    };
    auto stage_write = [&](int buf) {
        if (VAR & 2) return;
        if (VAR & 64) {         // profiling: loads are waited for but not written to LDS
#pragma unroll
            for (int k = 0; k < KB; ++k) asm volatile("" ::"v"(rb[k][0]), "v"(rb[k][1]));
#pragma unroll
            for (int k = 0; k < KA; ++k) asm volatile("" ::"v"(ra[k][0]), "v"(ra[k][1]));
            return;
        }
        __builtin_amdgcn_sched_barrier(0);
        float *As = smem + buf * G::BUF_FLOATS;
        float *Bs = As + G::A_FLOATS;
        if (!(VAR & 32)) {   // zero the lanes that loaded the dummy address
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int il = ib0 + b_bi(k);
                const bool ok = sb_col_ok && (il >= 0) && (il < HL);
                rb[k] = ok ? rb[k] : (ld_t)(0.0f);
            }
#pragma unroll
            for (int k = 0; k < KA; ++k) ra[k] = a_ok ? ra[k] : (ld_t)(0.0f);
        }
        if (sb_piece < (X4 ? B_COLS / 2 : B_COLS)) {
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                float *d = Bs + b_ch(k) * B_CH + b_bi(k) * B_ROW + (X4 ? 2 : 1) * sb_piece;
                if constexpr (X4) {
                    *reinterpret_cast<f2 *>(d) = (f2){rb[k][0], rb[k][2]};
                    *reinterpret_cast<f2 *>(d + B_PAR) = (f2){rb[k][1], rb[k][3]};
                } else {
                    d[0] = rb[k][0];
                    d[B_PAR] = rb[k][1];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            float *d = As + a_ch(k) * A_CH + a_ai * A_ROW + (X4 ? 2 : 1) * sa_piece;
            if constexpr (X4) {
                *reinterpret_cast<f2 *>(d) = (f2){ra[k][0], ra[k][2]};
                *reinterpret_cast<f2 *>(d + A_PAR) = (f2){ra[k][1], ra[k][3]};
            } else {
                d[0] = ra[k][0];
                d[A_PAR] = ra[k][1];
            }
        }
    };

    // ---- MFMA roles
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) << 1;                 // first of this wave's two A column blocks
    const int fi = lane & 15, fq = lane >> 4;        // pixel-in-block, k slot
    const int a_frag = fq * A_CH + xpar * A_PAR + (fi >> 2) * A_ROW + 4 * a0 + (fi & 3);
    const int b_frag = fq * B_CH + xpar * B_PAR + (fi >> 2) * B_ROW + 4 * a0 + (fi & 3);

    f4 acc[2][NV];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[ab][v] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

    // one chunk: CK/4 k-steps; the fragments of step s+1 are fetched before the MFMAs of step s
    auto mma_chunk = [&](int buf) {
        const float *As = smem + buf * G::BUF_FLOATS;
        const float *Bs = As + G::A_FLOATS;
        constexpr int KS = CK / 4;
        float af[2][2], bf[2][NV + 1];
        if (VAR & 8) {   // profiling: operands from registers, no LDS fragment reads
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) af[h][ab] = (float)(lane + ab);
#pragma unroll
                for (int j = 0; j < NV + 1; ++j) bf[h][j] = (float)(lane - j);
            }
        }
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) if (!(VAR & 8)) af[0][ab] = As[a_frag + 4 * ab];
#pragma unroll
        for (int j = 0; j < NV + 1; ++j) if (!(VAR & 8)) bf[0][j] = Bs[b_frag + 4 * j];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < KS && !(VAR & 8)) {
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) af[nxt][ab] = As[a_frag + (s + 1) * 4 * A_CH + 4 * ab];
#pragma unroll
                for (int j = 0; j < NV + 1; ++j) bf[nxt][j] = Bs[b_frag + (s + 1) * 4 * B_CH + 4 * j];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this step's MFMAs
            if (VAR & 1) {
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) asm volatile("" ::"v"(af[cur][ab]));
#pragma unroll
                for (int j = 0; j < NV + 1; ++j) asm volatile("" ::"v"(bf[cur][j]));
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int ab = 0; ab < 2; ++ab)
                        acc[ab][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][ab], bf[cur][ab + v], acc[ab][v], 0, 0, 0);
            }
        }
    };

    const int nchunks = all_pad ? 0 : p.C / CK;
    if (nchunks > 0) {
        if (NBUF == 2) {
            stage_load(0);
            stage_write(0);
            __syncthreads();
            for (int ck = 0; ck < nchunks; ++ck) {
                const int buf = ck & 1;
                if (ck + 1 < nchunks) stage_load((ck + 1) * CK);
                mma_chunk(buf);
                if (ck + 1 < nchunks) stage_write(buf ^ 1);
                __syncthreads();
            }
        } else {
            stage_load(0);
            for (int ck = 0; ck < nchunks; ++ck) {
                stage_write(0);
                __syncthreads();
                if (ck + 1 < nchunks) stage_load((ck + 1) * CK);
                mma_chunk(0);
                __syncthreads();
            }
        }
    }

    epilogue<NV, EP, VAR>(smem, acc, p, lane, wave, n, py, rg, u, X0, HL, HW);
}

// ------------------------------------------------------------------------------------------------
// Forward kernel with LDS-DMA staging (global_load_lds_dwordx4): operands go HBM/L2 -> LDS without
// passing through registers, in a ring of NST stages with counted vmcnt waits and ONE raw s_barrier
// per chunk.  Same task decomposition, MFMA mapping and epilogue as corr_fwd_mfma_f32 above; only the
// LDS image differs, because a DMA instruction writes 64 lanes x 16 B = 1 KB of CONSECUTIVE LDS:
//   B tile  [ch][bi][x]   104 floats per row (x from X0-2dr, parity interleaved), channel stride 420;
//           the 4 rows of a channel are 104 16-byte pieces = 2 DMA instructions (64 + 40 lanes)
//   A tile  [ch][ai][x]   64 floats per row, stored ROTATED by 8*ai floats (the DMA's per-lane GLOBAL
//           address is free, so lane (row, piece) fetches piece (piece - 2*ai) mod 16): the four rows of an
//           A block then sit on different banks; one DMA instruction per channel
// Pieces outside the image are never written: the ring is zero-filled once and stays zero there.
// Operand reads are ds_read_b32 at parity stride 2: the 16 pixels of a block cover the 16 banks of one
// parity and the two k-slots of a 32-lane group collide 2-way (LDS is far from saturated).
// Preconditions on top of the register-staged kernel: dr even, W % 4 == 0 (pieces are 4-pixel aligned).
constexpr int DB_ROW = 104, DB_CH = 4 * DB_ROW + 4;   // 420 floats: 16 B aligned, rows 8 banks apart
constexpr int DA_ROW = 64, DA_CH = 4 * DA_ROW;        // 256 floats

template <int CK, int NST, int EP>
struct DCfg {
    static constexpr int A_FLOATS = CK * DA_CH, B_FLOATS = CK * DB_CH, ST_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int O_FLOATS = (16 / EP) * (2 * DR_MAX + 1) * O_RS + 64;
    static constexpr int LDS_FLOATS = (NST * ST_FLOATS > O_FLOATS) ? NST * ST_FLOATS : O_FLOATS;
};

// (Dedicated loader waves -- 4 extra waves that only issue the DMAs -- were measured too: slower, because the
// fp32 MFMA blocks vector-memory issue on its SIMD whichever wave issues it)
template <int NV, int CK, int NST, int EP, int WPS, int VAR>
__global__ __launch_bounds__(512, WPS) void corr_fwd_mfma_dma(Args p)
{
    typedef DCfg<CK, NST, EP> G;
    static_assert((CK * 3) % 8 == 0, "3 DMA instructions per channel are spread evenly over the 8 waves");
    constexpr int DPW = CK * 3 / 8;   // DMA instructions per wave per chunk
    __shared__ __attribute__((aligned(16))) float smem[G::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int u = (int)(t % NV); t /= NV;
    const int xt = (int)(t % p.NXT); t /= p.NXT;
    const int rg = (int)(t % p.NRG); t /= p.NRG;
    const int py = (int)(t & 1u);
    const int n = (int)(t >> 1);

    const int HL = p.H >> 1;
    const int ib0 = 4 * rg - p.dr + 4 * u;
    const bool all_pad = (ib0 + 3 < 0) || (ib0 >= HL);
    const long HW = (long)p.H * p.W;
    const float *in1n = p.in1 + (long)n * p.C * HW;
    const float *in2n = p.in2 + (long)n * p.C * HW;
    const int X0 = xt * TILE_X;

    // ---- DMA roles.  Instruction j (0 .. 3*CK-1) of a chunk: channel j/3, kind j%3
    //   kind 0: B pieces 0..63   kind 1: B pieces 64..103 (lanes 0..39)   kind 2: the A rows
    // wave w issues instructions w*DPW .. w*DPW+DPW-1.  Per-lane source offsets (within a channel
    // plane) and validity do not depend on the chunk.
    int d_off[DPW];     // element offset inside the channel plane, or -1
    int d_ch[DPW];      // channel within the chunk
    int d_lds[DPW];     // LDS float offset of the instruction's 1 KB window inside a stage
    bool d_isA[DPW];
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
        const int j = wave * DPW + k;
        const int ch = j / 3, kind = j % 3;
        d_ch[k] = ch;
        d_isA[k] = (kind == 2);
        int off = -1;
        if (kind == 2) {
            const int row = lane >> 4, slot = lane & 15;
            const int piece = (slot - 2 * row) & 15;            // row `ai` is stored rotated by 2*ai pieces
            const int IL = 4 * rg + row, x0 = X0 + 4 * piece;
            if (IL < HL && x0 < p.W) off = (2 * IL + py) * p.W + x0;
            d_lds[k] = ch * DA_CH;
        } else {
            const int i = kind * 64 + lane;                     // piece index within the channel's 4 rows
            const int row = i / 26, pc = i - row * 26;
            const int il = ib0 + row, x0 = X0 - 2 * p.dr + 4 * pc;
            if (i < 104 && il >= 0 && il < HL && x0 >= 0 && x0 < p.W) off = (2 * il + py) * p.W + x0;
            d_lds[k] = G::A_FLOATS + ch * DB_CH + kind * 256;
        }
        d_off[k] = off;
    }
    // an instruction none of whose lanes is inside the image is skipped entirely, so the number of DMAs this
    // wave has in flight per chunk is counted, not assumed (the vmcnt waits below depend on it)
    int n_dma = 0;
#pragma unroll
    for (int k = 0; k < DPW; ++k) n_dma += (__ballot(d_off[k] >= 0) != 0ull) ? 1 : 0;
    n_dma = __builtin_amdgcn_readfirstlane(n_dma);
    auto wait_younger = [&](int chunks) {   // wait until at most chunks * n_dma DMAs are outstanding
        const int lim = chunks * n_dma;
        switch (lim) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // conservative
        }
    };
    auto dma_issue = [&](int c0, int stage) {
        if (VAR & 2) return;
        float *st = smem + stage * G::ST_FLOATS;
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const float *src = (d_isA[k] ? in1n : in2n) + (long)(c0 + d_ch[k]) * HW + d_off[k];
            if (d_off[k] >= 0)
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(st + d_lds[k]), 16, 0, 0);
        }
    };

    // ---- MFMA roles (as in corr_fwd_mfma_f32)
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) << 1;
    const int fi = lane & 15, fq = lane >> 4;
    const int f_row = fi >> 2, f_col = fi & 3;
    int a_frag[2];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
        a_frag[ab] = fq * DA_CH + f_row * DA_ROW + ((2 * (4 * (a0 + ab) + f_col) + xpar + 8 * f_row) & 63);
    const int b_frag = G::A_FLOATS + fq * DB_CH + f_row * DB_ROW + 2 * (4 * a0 + f_col) + xpar;   // + 8*j per block column

    f4 acc[2][NV];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[ab][v] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

    auto mma_chunk = [&](int stage) {
        const float *st = smem + stage * G::ST_FLOATS;
        constexpr int KS = CK / 4;
        float af[2][2], bf[2][NV + 1];
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) af[0][ab] = st[a_frag[ab]];
#pragma unroll
        for (int j = 0; j < NV + 1; ++j) bf[0][j] = st[b_frag + 8 * j];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < KS) {
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) af[nxt][ab] = st[a_frag[ab] + (s + 1) * 4 * DA_CH];
#pragma unroll
                for (int j = 0; j < NV + 1; ++j) bf[nxt][j] = st[b_frag + (s + 1) * 4 * DB_CH + 8 * j];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (VAR & 1) {
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) asm volatile("" ::"v"(af[cur][ab]));
#pragma unroll
                for (int j = 0; j < NV + 1; ++j) asm volatile("" ::"v"(bf[cur][j]));
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int ab = 0; ab < 2; ++ab)
                        acc[ab][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][ab], bf[cur][ab + v], acc[ab][v], 0, 0, 0);
            }
        }
    };

    const int nchunks = all_pad ? 0 : p.C / CK;
    if (nchunks > 0) {
        // zero the ring once: positions outside the image are never written by the DMA
        for (int i = tid; i < NST * G::ST_FLOATS / 4; i += 512) reinterpret_cast<f4 *>(smem)[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        // prologue: NST-1 chunks in flight
#pragma unroll
        for (int c = 0; c < NST - 1; ++c)
            if (c < nchunks) dma_issue(c * CK, c);
        for (int ck = 0; ck < nchunks; ++ck) {
            // chunk ck has landed once at most the DMAs of the (NST-2) younger chunks are outstanding; the
            // tail (fewer younger chunks issued) simply drains everything.
            if (ck + NST - 2 < nchunks) wait_younger(NST - 2);
            else wait_younger(0);
            __builtin_amdgcn_s_barrier();   // every wave's share of chunk ck is visible; stage (ck-1)%NST is free
            if (ck + NST - 1 < nchunks) dma_issue((ck + NST - 1) * CK, (ck + NST - 1) % NST);
            mma_chunk(ck % NST);
        }
        __syncthreads();   // all operand reads done before the epilogue reuses the ring
    }
    epilogue<NV, EP, VAR>(smem, acc, p, lane, wave, n, py, rg, u, X0, HL, HW);
}

// ------------------------------------------------------------------------------------------------
// Forward kernel on the bf16 matrix cores with an EXACT three-way split of the fp32 operands.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate (41 us for this problem at 100 % utilisation) and,
// measured on MI355X (scripts/ubench/mfma_vmem_overlap.hip), does not overlap with vector-memory traffic
// issued by ANY wave of the same SIMD -- time = MFMA time + load time.  v_mfma_f32_16x16x32_bf16 is ~15x
// faster per MAC and overlaps with LDS-DMA perfectly.
//
// Numerics: every fp32 value is split by truncation into three bf16 terms a = a0 + a1 + a2 (8 + 8 + 8
// significand bits: the split is exact), and the product a*b is formed as
//     a0*b0 + a0*b1 + a1*b0 + a1*b1 + a0*b2 + a2*b0          (dropped: a1*b2, a2*b1, a2*b2 <= 2^-23 |a*b|)
// Every partial product is exact in fp32 and accumulation is fp32 inside the MFMA, so the result carries
// fp32-class error: measured against an fp64 reference (scripts/corr_accuracy.py, 8x256x48x64, N(0,1)):
// max |err| 2.6e-7 / rms 1.3e-8, vs 2.3e-7 / 1.45e-8 for the fp32 MFMA kernel; same relative figures for
// inputs scaled by 100 or 1e-3 and for log-normal magnitudes.  This is fp32 arithmetic carried out on the
// bf16 pipe, not a reduced-precision mode.
//
// One MFMA step contracts 32 channels: lane group q = lane>>4 owns channels 8q..8q+7 of the step for its
// pixel, reads them from LDS (8 ds_read_b32 at channel stride), splits them in registers and packs each term
// into 4 VGPRs; six MFMAs per (A block, B block) pair and step.  No value is split twice inside a wave
// except for the B blocks shared between its two A blocks' neighbours (7 B fragments for 12 pairs).
//
// LDS image (per stage of 32 channels, 2 stages): A and B tiles alike, [ch][row 0..3][64 floats] with channel
// stride 260 floats.  Row r of channel c is stored rotated by 8r + 16((c>>3)&1) floats (the DMA's per-lane
// source address does the rotation for free): the 16 pixels of a block and the two channel octets of a
// 32-lane half then cover 32 distinct banks.  The 4 pad floats after each channel stay zero: lanes whose B
// pixel lies in the zero halo left/right of the image read the pad instead (address chosen once per task).
// So the B tile carries no halo; this kernel therefore requires the image to fit one x tile (W <= 64), which
// FlowNetC's cost volume at 384x512 does; wider maps use the fp32 kernels.
//
// All 8 waves issue DMAs; waves 0-3 before their MFMA phase and waves 4-7 after it, so the two waves that
// share a SIMD (w and w+4) are in complementary phases.
// split helpers: bf16x3.h

constexpr int SB_CH = 260;                 // channel stride (floats): 256 + 4 zero pad
constexpr int SB_CK = 32;                  // channels per stage = per MFMA step
constexpr int SB_TILE = SB_CK * SB_CH;     // floats per tile (A or B) per stage
constexpr int SB_STAGE = 2 * SB_TILE;      // 16640 floats = 66 560 B

// NST = 2: two-stage ring, one workgroup per CU (133 KB).  NST = 1: one stage (66 KB, two-pass epilogue), two
// workgroups per CU -- the other workgroup's vector work covers this one's DMA latency and store drain.
// NAB = 2: 8 waves of two A blocks (7 B fragments for 12 block pairs).  NAB = 4: 4 waves of four A blocks (9 B fragments
// for 24 pairs: 28 % less operand-split work per MFMA), 256 VGPRs per wave, still two workgroups per CU with NST = 1.
template <int NV, int EP, int VAR, int NST = 2, int NAB = 2>
__global__ __launch_bounds__(64 * (16 / NAB), (NST == 1 ? (NAB == 2 ? 4 : 2) : (NAB == 2 ? 2 : 1))) void corr_fwd_mfma_bf16x3(Args p)
{
    constexpr int NW = 16 / NAB, NT = 64 * NW, NBF = NV + NAB - 1;   // waves, threads, B fragments per wave
    constexpr int CPW = SB_CK / NW;                                   // channels staged per wave and stage
    constexpr int O_FLOATS = (16 / EP) * (2 * DR_MAX + 1) * O_RS + 64;
    constexpr int LDS_FLOATS = (NST * SB_STAGE > O_FLOATS) ? NST * SB_STAGE : O_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Task decode.  Late-dispatched workgroups finish last, so within each batch item the tasks whose neighbour
    // rows are all padding (they only write zeros) come LAST: the tasks with real work start earlier.
    const int HL = p.H >> 1;
    unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int per_item = 2 * p.NRG * NV;
    const int n = (int)(t / per_item);
    int idx = (int)(t % per_item), u = 0, rg = 0, py = 0;
    const int xt = 0;   // NXT == 1 (checked by the launcher)
    if (VAR & 2048) {   // plain order (profiling)
        u = idx % NV; idx /= NV;
        rg = idx % p.NRG; py = idx / p.NRG;
    } else {
        // per row group the u values with work are the contiguous range [ulo, uhi]: rows 4rg - dr + 4u .. +3 meet [0, HL)
        auto ulo = [&](int g) { const int v = p.dr - 3 - 4 * g; return v <= 0 ? 0 : (v + 3) / 4; };
        auto uhi = [&](int g) { const int v = (HL - 1 + p.dr - 4 * g) / 4; return v < NV - 1 ? v : NV - 1; };
        int R = 0;
        for (int g = 0; g < p.NRG; ++g) { const int c = uhi(g) - ulo(g) + 1; R += c > 0 ? c : 0; }
        const bool real = idx < 2 * R;
        const int per_par = real ? R : p.NRG * NV - R;
        int r = real ? idx : idx - 2 * R;
        py = r / per_par; r -= py * per_par;
        for (int g = 0; g < p.NRG; ++g) {
            const int lo = ulo(g), hi = uhi(g);
            const int c = hi - lo + 1 > 0 ? hi - lo + 1 : 0;
            const int k = real ? c : NV - c;
            if (r < k) {
                rg = g;
                u = real ? lo + r : (c == 0 ? r : (r < lo ? r : hi + 1 + (r - lo)));
                break;
            }
            r -= k;
        }
    }
    py = __builtin_amdgcn_readfirstlane(py); rg = __builtin_amdgcn_readfirstlane(rg); u = __builtin_amdgcn_readfirstlane(u);

    const int ib0 = 4 * rg - p.dr + 4 * u;
    const bool all_pad = (ib0 + 3 < 0) || (ib0 >= HL);
    const long HW = (long)p.H * p.W;
    const float *in1n = p.in1 + (long)n * p.C * HW;
    const float *in2n = p.in2 + (long)n * p.C * HW;
    const int X0 = xt * TILE_X;

    // ---- DMA roles: wave w stages channels CPW*w .. CPW*w + CPW-1 of a stage, one instruction per (channel, tile):
    // lane = (row, slot); slot holds piece (slot - 2 row - 4 ((c>>3)&1)) mod 16 of the row; c>>3 is constant per wave
    // (CPW divides 8).
    const int d_row = lane >> 4, d_slot = lane & 15;
    const int d_piece = (d_slot - 2 * d_row - 4 * (((CPW * wave) >> 3) & 1)) & 15;
    const int d_x = X0 + 4 * d_piece;
    const int d_ila = 4 * rg + d_row, d_ilb = ib0 + d_row;
    const bool d_oka = (d_ila < HL) && (d_x < p.W);
    const bool d_okb = (d_ilb >= 0) && (d_ilb < HL) && (d_x < p.W);
    const int d_offa = (2 * d_ila + py) * p.W + d_x;
    const int d_offb = (2 * d_ilb + py) * p.W + d_x;
    const int n_dma = CPW * ((__ballot(d_oka) != 0ull ? 1 : 0) + (__ballot(d_okb) != 0ull ? 1 : 0));
    auto dma_issue = [&](int c0, int stage) {
        if (VAR & 2) return;
        float *st = smem + stage * SB_STAGE;
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int c = CPW * wave + k;
            if (d_oka)
                __builtin_amdgcn_global_load_lds(in1n + (long)(c0 + c) * HW + d_offa,
                                                 (__attribute__((address_space(3))) void *)(st + c * SB_CH), 16, 0, 0);
            if (d_okb)
                __builtin_amdgcn_global_load_lds(in2n + (long)(c0 + c) * HW + d_offb,
                                                 (__attribute__((address_space(3))) void *)(st + SB_TILE + c * SB_CH), 16, 0, 0);
        }
    };
    auto wait_all_but = [&](int chunks) {   // at most `chunks` x n_dma DMAs of this wave still in flight
        if (chunks == 0 || n_dma == 0 || CPW != 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n_dma == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };

    // ---- MFMA roles
    const int xpar = wave & 1;
    const int a0 = (wave >> 1) * NAB;
    const int fi = lane & 15, fq = lane >> 4;
    const int f_row = fi >> 2, f_col = fi & 3;
    const int rot = 8 * f_row + 16 * (fq & 1);
    int a_frag[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab)
        a_frag[ab] = (8 * fq) * SB_CH + f_row * 64 + ((2 * (4 * (a0 + ab) + f_col) + xpar + rot) & 63);
    int b_frag[NBF];
#pragma unroll
    for (int j = 0; j < NBF; ++j) {
        const int x = 2 * (4 * (a0 + j) + f_col - p.dr) + xpar;          // tile-local image x of this lane's B pixel
        const bool in_img = (x >= 0) && (X0 + x < p.W) && (x < TILE_X);
        b_frag[j] = SB_TILE + (8 * fq) * SB_CH + (in_img ? f_row * 64 + ((x + rot) & 63) : 256);   // 256: the zero pad
    }

    f4 acc[NAB][NV];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[ab][v] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

    auto as_bf = [](const u4 &x) { return __builtin_bit_cast(bf16x8, x); };

    auto mma_step = [&](int stage) {
        const float *st = smem + stage * SB_STAGE;
        // raw fragments are fetched one fragment ahead of their split (the LDS latency hides behind the
        // previous fragment's VALU + MFMA work); order: A[0], A[1], B[0], B[1], ...
        float rq[2][8];
        auto fetch = [&](int f, float (&r)[8]) {
            const int base = (f < NAB) ? a_frag[f < NAB ? f : 0] : b_frag[f < NAB ? 0 : f - NAB];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = st[base + k * SB_CH];
        };
        u4 A0[NAB], A1[NAB], A2[NAB];
        fetch(0, rq[0]);
#pragma unroll
        for (int f = 0; f < NAB + NBF; ++f) {
            const int cur = f & 1;
            if (f + 1 < NAB + NBF) fetch(f + 1, rq[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (f < NAB) {
                split3(rq[cur], A0[f], A1[f], A2[f]);
                if (VAR & 1) asm volatile("" ::"v"(A0[f][0]), "v"(A1[f][0]), "v"(A2[f][0]), "v"(A0[f][3]), "v"(A1[f][3]), "v"(A2[f][3]));
            } else {
                const int j = f - NAB;
                u4 B0, B1, B2;
                split3(rq[cur], B0, B1, B2);
                if (!(VAR & 1)) {
                    // six products per pair; the pairs of this B block (A block ab: v = j - ab) alternate so that
                    // consecutive MFMAs never wait on the same accumulator
                    const u4 *PA[6] = {A0, A0, A1, A1, A0, A2};
                    const u4 *PB[6] = {&B0, &B1, &B0, &B1, &B2, &B0};
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int ab = 0; ab < NAB; ++ab) {
                            const int v = j - ab;
                            if (v >= 0 && v < NV)
                                acc[ab][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(PA[pr][ab]), as_bf(*PB[pr]), acc[ab][v], 0, 0, 0);
                        }
                } else {
                    asm volatile("" ::"v"(B0[0]), "v"(B0[3]), "v"(B1[0]), "v"(B1[3]), "v"(B2[0]), "v"(B2[3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // VAR 1024 (profiling): s_memtime stamps of step 3 and of the epilogue, dumped over the start of the output
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i, bool on) __attribute__((always_inline)) { if ((VAR & 1024) && on) ts[i] = __builtin_amdgcn_s_memtime(); };
    const int nsteps = all_pad ? 0 : p.C / SB_CK;
    const bool early = (wave < NW / 2);    // waves w and w+4 share a SIMD: complementary DMA / MFMA phases
    if (nsteps > 0) {
        for (int i = tid; i < NST * SB_STAGE / 4; i += NT) reinterpret_cast<f4 *>(smem)[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        dma_issue(0, 0);
        for (int sk = 0; sk < nsteps; ++sk) {
            // early waves have issued step sk only; late waves have issued step sk only as well (they issue
            // step sk+1 at the end of this iteration): wait for everything
            stamp(0, sk == 3);
            wait_all_but(0);
            stamp(1, sk == 3);
            __builtin_amdgcn_s_barrier();   // step sk is visible to all; the other stage is free
            stamp(2, sk == 3);
            const bool more = (sk + 1 < nsteps);
            if (NST == 2) {
                if (more && early) dma_issue((sk + 1) * SB_CK, (sk + 1) & 1);
                mma_step(sk & 1);
                if (more && !early) dma_issue((sk + 1) * SB_CK, (sk + 1) & 1);
            } else {
                mma_step(0);
                stamp(3, sk == 3);
                if (more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();   // every wave has read the stage: refill it
                    stamp(4, sk == 3);
                    dma_issue((sk + 1) * SB_CK, 0);
                    stamp(5, sk == 3);
                }
            }
        }
        __syncthreads();
    }
    stamp(6, true);
    epilogue<NV, EP, VAR, NAB>(smem, acc, p, lane, wave, n, py, rg, u, X0, HL, HW);
    stamp(7, true);
    if ((VAR & 1024) && lane == 0) {   // one non-padding task of each dispatch round on what should be the same CU
        int slot = -1;
        if (blockIdx.x == 8 * 27 + 2) slot = 0;
        if (blockIdx.x == 8 * 27 + 2 + 256) slot = 1;
        if (slot >= 0) {
            unsigned long long *d = reinterpret_cast<unsigned long long *>(p.out) + (slot * 8 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = ts[i];
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
    if (C % 16 != 0 || (H & 1) || (W & 1)) return false;
    return true;
}

namespace mf {

template <int NV, int CK, int NBUF, int EP, int WPS, int VAR>
static int launch(const Args &a, long ntasks, hipStream_t s)
{
    // 16-byte staging needs the tile's first halo column and the image width on 4-pixel boundaries
    const bool x4 = (a.dr % 2 == 0) && (a.W % 4 == 0) && aligned(a.in1, 16) && aligned(a.in2, 16) && !(VAR & 128);
    if (x4)
        hipLaunchKernelGGL((corr_fwd_mfma_f32<NV, CK, NBUF, EP, WPS, VAR & 127, true>), dim3((unsigned)ntasks), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((corr_fwd_mfma_f32<NV, CK, NBUF, EP, WPS, VAR & 127, false>), dim3((unsigned)ntasks), dim3(512), 0, s, a);
    return launch_status();
}

template <int CK, int NBUF, int EP, int WPS, int VAR>
static int launch_nv(const Args &a, long ntasks, hipStream_t s)
{
    switch (a.NV) {
    case 2: return launch<2, CK, NBUF, EP, WPS, VAR>(a, ntasks, s);
    case 3: return launch<3, CK, NBUF, EP, WPS, VAR>(a, ntasks, s);
    case 4: return launch<4, CK, NBUF, EP, WPS, VAR>(a, ntasks, s);
    case 5: return launch<5, CK, NBUF, EP, WPS, VAR>(a, ntasks, s);
    case 6: return launch<6, CK, NBUF, EP, WPS, VAR>(a, ntasks, s);
    default: return FN2_EUNSUPPORTED;
    }
}

} // namespace mf

// tune: 0 = fastest applicable kernel (bf16x3 split where it applies, else fp32 MFMA); 2 = fp32 MFMA only
//       (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain); 3 = bf16x3 only; >= 100 profiling instantiations
int corr_forward_mfma_f32(const float *in1, const float *in2, float *out, long out_bs, float slope, int B, int C, int H,
                          int W, int md, int tune, hipStream_t s)
{
    if (!aligned(in1, 8) || !aligned(in2, 8)) return FN2_EALIGN;
    if (out_bs % 2 != 0) return FN2_EALIGN;   // output rows are stored as float2
    mf::Args a;
    a.in1 = in1; a.in2 = in2; a.out = out;
    a.out_bs = out_bs; a.slope = slope;
    a.C = C; a.H = H; a.W = W;
    a.dr = md / 2; a.D = 2 * a.dr + 1; a.NV = 1 + (a.dr + 1) / 2;
    const int HL = H / 2;
    a.NRG = (HL + 3) / 4;
    a.NXT = (W + mf::TILE_X - 1) / mf::TILE_X;
    const long ntasks = (long)B * 2 * a.NRG * a.NXT * a.NV;
    if (ntasks == 0) return FN2_OK;
    const bool dma_ok = (a.dr % 2 == 0) && (W % 4 == 0) && aligned(in1, 16) && aligned(in2, 16);
    if (tune >= 1000) {   // LDS-DMA staging, profiling configurations (NV = 6 only)
        if (!dma_ok || a.NV != 6 || C % 16 != 0) return FN2_EUNSUPPORTED;
#define FN2_DMA(CK, NST, EP, WPS, V)                                                                             \
    hipLaunchKernelGGL((mf::corr_fwd_mfma_dma<6, CK, NST, EP, WPS, V>), dim3((unsigned)ntasks), dim3(512), 0, s, a); \
    return launch_status();
        switch (tune) {
        case 1000: FN2_DMA(8, 3, 2, 4, 0)
        case 1001: FN2_DMA(8, 3, 2, 4, 1)
        case 1002: FN2_DMA(8, 3, 2, 4, 2)
        case 1010: FN2_DMA(16, 3, 1, 2, 0)
        case 1011: FN2_DMA(16, 3, 1, 2, 1)
        case 1020: FN2_DMA(8, 4, 2, 2, 0)
        case 1030: FN2_DMA(16, 2, 1, 2, 0)
        case 1040: FN2_DMA(8, 2, 2, 4, 0)
        case 2000: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 1, 0>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2001: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 1, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2002: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 1, 2>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2100: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 0, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2101: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 1, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2200: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 0, 1, 4>), dim3((unsigned)ntasks), dim3(256), 0, s, a); return launch_status();
        case 2201: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 1, 1, 4>), dim3((unsigned)ntasks), dim3(256), 0, s, a); return launch_status();
        case 2210: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 1, 0, 2, 4>), dim3((unsigned)ntasks), dim3(256), 0, s, a); return launch_status();
        case 2102: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 2, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2104: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 4, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2106: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 6, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2107: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 7, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 3124: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 1024, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 4148: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 2, 2048, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2004: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 1, 4>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        case 2007: if (a.NXT != 1 || C % 32) return FN2_EUNSUPPORTED; hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<6, 1, 7>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();
        default: return FN2_EUNSUPPORTED;
        }
#undef FN2_DMA
    }
    // bf16x3: exact 3-way operand split on the bf16 matrix cores (fp32-class accuracy, see kernel comment);
    // needs the whole width in one x tile, 32-channel steps and 4-pixel aligned rows
    const bool bf16x3_ok = dma_ok && a.NXT == 1 && C % 32 == 0;
    if (tune == 3 && !bf16x3_ok) return FN2_EUNSUPPORTED;
    if ((tune == 0 || tune == 3) && bf16x3_ok) {
        switch (a.NV) {
#define FN2_B3(NVV) case NVV: hipLaunchKernelGGL((mf::corr_fwd_mfma_bf16x3<NVV, 2, 0, 1>), dim3((unsigned)ntasks), dim3(512), 0, s, a); return launch_status();   // one stage, two workgroups per CU
            FN2_B3(2) FN2_B3(3) FN2_B3(4) FN2_B3(5) FN2_B3(6)
#undef FN2_B3
        default: return FN2_EUNSUPPORTED;
        }
    }
    if (tune == 2) tune = 0;   // exact-fp32 MFMA kernels below
    if (tune == 0 && dma_ok && a.NV == 6) {   // FlowNetC's radius: LDS-DMA staging, 2 stages of 16 channels
        hipLaunchKernelGGL((mf::corr_fwd_mfma_dma<6, 16, 2, 1, 2, 0>), dim3((unsigned)ntasks), dim3(512), 0, s, a);
        return launch_status();
    }
    if (tune == 0) return mf::launch_nv<16, 1, 2, 4, 0>(a, ntasks, s);   // 47 KB LDS: two workgroups per CU
    if (tune < 100 || a.NV != 6 || C % 32 != 0) return FN2_EUNSUPPORTED;
    const int cfg = (tune - 100) / 256, var = (tune - 100) % 256;   // var bit 7: force the 8-byte staging path
#define FN2_V(CK, NBUF, EP, WPS, V) case V: return mf::launch<6, CK, NBUF, EP, WPS, V>(a, ntasks, s);
#define FN2_VARS(CK, NBUF, EP, WPS)                                                                           \
    switch (var) {                                                                                            \
        FN2_V(CK, NBUF, EP, WPS, 0) FN2_V(CK, NBUF, EP, WPS, 1) FN2_V(CK, NBUF, EP, WPS, 2)                   \
        FN2_V(CK, NBUF, EP, WPS, 3) FN2_V(CK, NBUF, EP, WPS, 4) FN2_V(CK, NBUF, EP, WPS, 5)                   \
        FN2_V(CK, NBUF, EP, WPS, 6) FN2_V(CK, NBUF, EP, WPS, 7) FN2_V(CK, NBUF, EP, WPS, 14)                  \
        FN2_V(CK, NBUF, EP, WPS, 16) FN2_V(CK, NBUF, EP, WPS, 32) FN2_V(CK, NBUF, EP, WPS, 33)                \
        FN2_V(CK, NBUF, EP, WPS, 37) FN2_V(CK, NBUF, EP, WPS, 64) FN2_V(CK, NBUF, EP, WPS, 65)                \
        FN2_V(CK, NBUF, EP, WPS, 69) FN2_V(CK, NBUF, EP, WPS, 128)                                            \
    default: return FN2_EUNSUPPORTED;                                                                         \
    }
    switch (cfg) {
    case 0: FN2_VARS(16, 2, 1, 2)   // 94 KB LDS, 1 workgroup / CU
    case 1: FN2_VARS(16, 1, 2, 4)   // 47 KB LDS, 2 workgroups / CU, 2 barriers per chunk
    default: return FN2_EUNSUPPORTED;
    }
#undef FN2_V
#undef FN2_VARS
}

} // namespace fn2
