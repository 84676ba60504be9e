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
#include "corr_params.h"

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
    int C, H, W;     // H, W even
    int dr, D, NV;   // displacement radius (lattice), 2dr+1, B blocks per A block per axis
    int NRG, NXT;    // row groups per parity, x tiles
};

// Tunables of one instantiation.
//   CK    channels per LDS chunk (multiple of 4)
//   NBUF  1: one operand buffer, two barriers per chunk; 2: double-buffered, one barrier per chunk
//   EP    epilogue passes (1: all 16 (ai,bi) planes at once = 87 KB of LDS; 2: 8 planes per pass)
//   WPS   waves per SIMD the register budget is sized for (= workgroups per CU * 2)
//   VAR   ablation switches for profiling only (0 = the real kernel): 1 no MFMA, 2 no staging, 4 no stores
template <int CK, int NBUF, int EP>
struct Cfg {
    static constexpr int A_FLOATS = CK * A_CH, B_FLOATS = CK * B_CH, BUF_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int O_FLOATS = (16 / EP) * (2 * DR_MAX + 1) * O_RS + 64;   // + one spare row
    static constexpr int LDS_FLOATS = (NBUF * BUF_FLOATS > O_FLOATS) ? NBUF * BUF_FLOATS : O_FLOATS;
};

template <int NV, int CK, int NBUF, int EP, int WPS, int VAR>
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
    // (channel, bi)) and CK*4 A rows; wave w takes B rows w, w+8, ... (lane = lattice column jb) and
    // A row pairs 2w, 2w+1, 2w+16, ... (lane>>5 picks the row, lane&31 = ja).
    constexpr int KB = CK / 2;   // B rows per wave per chunk: row index = 8k + w -> channel 2k + (w>>2), bi = w&3
    constexpr int KA = CK / 4;   // A row pairs per wave per chunk: row = 16k + 2w + (lane>>5) -> channel 4k + (w>>1)
    const int s_bi = wave & 3;
    const int s_jb = lane;                                       // < B_COLS active
    const int s_yb = 2 * (ib0 + s_bi) + py;                      // image row
    const int s_xb = X0 - 2 * p.dr + 2 * s_jb;                   // image x of the even element
    const bool b_ok = (s_jb < B_COLS) && (ib0 + s_bi >= 0) && (ib0 + s_bi < HL) && (s_xb >= 0) && (s_xb < p.W);
    const float *b_src = b_ok ? in2n + (long)(wave >> 2) * HW + (long)s_yb * p.W + s_xb : in2n;
    const int b_dst = (wave >> 2) * B_CH + s_bi * B_ROW + s_jb; // + k*2*B_CH, + B_PAR for the odd element
    const int s_ai = ((wave & 1) << 1) + (lane >> 5);
    const int s_ja = lane & 31;
    const int s_ya = 2 * (4 * rg + s_ai) + py;
    const int s_xa = X0 + 2 * s_ja;
    const bool a_ok = (4 * rg + s_ai < HL) && (s_xa < p.W);
    const float *a_src = a_ok ? in1n + (long)(wave >> 1) * HW + (long)s_ya * p.W + s_xa : in1n;
    const int a_dst = (wave >> 1) * A_CH + s_ai * A_ROW + s_ja;  // + k*4*A_CH, + A_PAR for the odd element
    const float b_keep = b_ok ? 1.0f : 0.0f, a_keep = a_ok ? 1.0f : 0.0f;
    (void)b_keep; (void)a_keep;

    f2 rb[KB], ra[KA];
    auto stage_load = [&](int c0) {
        if (VAR & 2) return;
        // out-of-image lanes read a valid dummy address and are zeroed by the select below: no
        // divergent branch around the loads
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            f2 v = *reinterpret_cast<const f2 *>(b_src + (b_ok ? (long)(c0 + 2 * k) * HW : 0));
            rb[k] = b_ok ? v : (f2){0.0f, 0.0f};
        }
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            f2 v = *reinterpret_cast<const f2 *>(a_src + (a_ok ? (long)(c0 + 4 * k) * HW : 0));
            ra[k] = a_ok ? v : (f2){0.0f, 0.0f};
        }
    };
    auto stage_write = [&](int buf) {
        if (VAR & 2) return;
        float *As = smem + buf * G::BUF_FLOATS;
        float *Bs = As + G::A_FLOATS;
        if (s_jb < B_COLS) {
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                Bs[b_dst + k * 2 * B_CH] = rb[k][0];
                Bs[b_dst + k * 2 * B_CH + B_PAR] = rb[k][1];
            }
        }
#pragma unroll
        for (int k = 0; k < KA; ++k) {
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
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) af[0][ab] = As[a_frag + 4 * ab];
#pragma unroll
        for (int j = 0; j < NV + 1; ++j) bf[0][j] = Bs[b_frag + 4 * j];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < KS) {
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

    // ---- epilogue: accumulators -> LDS [ai][bi][ti][x] -> coalesced rows
    // D layout of v_mfma_f32_16x16x4_f32: lane l, register r holds D[row = 4*(l>>4) + r][col = l&15];
    // rows are A pixels (ai = row>>2 = l>>4, aj = row&3 = r), columns B pixels (bi = fi>>2, bj = fi&3).
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
            for (int ab = 0; ab < 2; ++ab)
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
                        if (EP == 1 || mine) Os[addr] = acc[ab][v][r];
                    }
            __syncthreads();
            // each wave writes whole planes: rows (plane, ti) for ti = 0..D-1, two rows per instruction,
            // 8 B per lane -> one 256 B contiguous segment per row
            for (int pl = wave; pl < 4 * AI_PER_PASS; pl += 8) {
                const int ai = pass * AI_PER_PASS + (pl >> 2), bi = pl & 3;
                const int tj = 4 * u + bi - ai;
                const int IL = 4 * rg + ai;
                if (tj < 0 || tj >= p.D || IL >= HL) continue; // wave-uniform
                const int y = 2 * IL + py;
                float *orow = p.out + (((long)n * p.D * p.D + (long)tj * p.D) * p.H + y) * p.W + xg;
                const float *srow = Os + (long)pl * p.D * O_RS + 2 * hx;
                for (int ti0 = 0; ti0 < p.D; ti0 += 2) {
                    const int ti = ti0 + hr;
                    if (ti < p.D && xg < p.W && !(VAR & 4)) {
                        f2 val = *reinterpret_cast<const f2 *>(srow + ti * O_RS);
                        if (pow2) { val[0] *= rC; val[1] *= rC; }
                        else { val[0] /= fC; val[1] /= fC; }
                        *reinterpret_cast<f2 *>(orow + (long)ti * HW) = val;
                    }
                }
            }
            if (pass + 1 < EP) __syncthreads();
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
    hipLaunchKernelGGL((corr_fwd_mfma_f32<NV, CK, NBUF, EP, WPS, VAR>), dim3((unsigned)ntasks), dim3(512), 0, s, a);
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

// tune: 0 = shipped configuration; 100 + 8*cfg + var = profiling instantiations (NV = 6 only)
int corr_forward_mfma_f32(const float *in1, const float *in2, float *out, int B, int C, int H, int W, int md,
                          int tune, hipStream_t s)
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
    if (tune == 0) return mf::launch_nv<16, 1, 2, 4, 0>(a, ntasks, s);   // 47 KB LDS: two workgroups per CU
    if (tune < 100 || a.NV != 6 || C % 32 != 0) return FN2_EUNSUPPORTED;
    const int cfg = (tune - 100) / 8, var = (tune - 100) % 8;
#define FN2_VARS(CK, NBUF, EP, WPS)                                                   \
    switch (var) {                                                                    \
    case 0: return mf::launch<6, CK, NBUF, EP, WPS, 0>(a, ntasks, s);                 \
    case 1: return mf::launch<6, CK, NBUF, EP, WPS, 1>(a, ntasks, s);                 \
    case 2: return mf::launch<6, CK, NBUF, EP, WPS, 2>(a, ntasks, s);                 \
    case 3: return mf::launch<6, CK, NBUF, EP, WPS, 3>(a, ntasks, s);                 \
    case 4: return mf::launch<6, CK, NBUF, EP, WPS, 4>(a, ntasks, s);                 \
    default: return FN2_EUNSUPPORTED;                                                 \
    }
    switch (cfg) {
    case 0: FN2_VARS(16, 2, 1, 2)   // 94 KB LDS, 1 workgroup / CU
    case 1: FN2_VARS(16, 1, 2, 4)   // 47 KB LDS, 2 workgroups / CU, 2 barriers per chunk
    case 2: FN2_VARS(8, 2, 2, 4)    // 47 KB LDS, 2 workgroups / CU, 1 barrier per 8-channel chunk
    case 3: FN2_VARS(32, 1, 1, 2)   // 94 KB LDS, 1 workgroup / CU, 32-channel chunks
    default: return FN2_EUNSUPPORTED;
    }
#undef FN2_VARS
}

} // namespace fn2
