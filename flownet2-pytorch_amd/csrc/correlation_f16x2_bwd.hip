// correlation_f16x2_bwd.hip -- correlation backward (both input gradients) on the gfx950 f16 matrix cores, with the
// operand handling of correlation_f16x2.hip: every fp32 value is split once per use into two f16 terms
// (x = h + l, h = RNE_f16(x), l = RNE_f16(x - h)) and a product is ah*bh + ah*bl + al*bh in fp32 accumulators.
//
// Replaces reference kernels correlation_backward_input1 / correlation_backward_input2
// (correlation_cuda_kernel.cu:150-241, :243-334; one launch per batch item each, :522-554) for FlowNetC's
// configuration (kernel_size 1, stride1 1, stride2 2, pad == max_displacement == 20, fp32, maps up to 64 wide):
//     gI1[n,c,p] = (1/C) * sum_d gO[n, tc(d), p     ] * in2[n,c, p + 2d]
//     gI2[n,c,p] = (1/C) * sum_d gO[n, tc(d), p - 2d] * in1[n,c, p - 2d]          d in [-10,10]^2 (lattice units of 2 px)
// Both are a banded contraction over the neighbours q of a "centre" pixel p on its parity lattice:
//     g[c, p] = sum_q  X[c, q] * G[q, p]        X = in2, G[q,p] = gO[q - p][p]   (FLIP 0: the gO pixel is the centre)
//                                               X = in1, G[q,p] = gO[p - q][q]   (FLIP 1: the gO pixel is the neighbour)
// i.e. a matrix product with M = 16 channels, K = neighbour pixels (two 4x4 blocks = 32 per MFMA), N = the 16 centre
// pixels of a 4x4 block.
//
// Task = (FLIP, n, y parity, row group rg of 4 centre lattice rows, group of 64 channels); the workgroup loops over the
// 6 neighbour row blocks u itself (rows 4rg - 10 + 4u .. +3): the sum over neighbours stays in registers, no atomics,
// deterministic.  Per u:
//   G image   the 16 (centre row ai, neighbour row bi) combinations x 21 displacement columns x 64 pixels of gO that the
//             pair (rg, u) touches -- the forward's output tile -- as raw fp32 rows [ai][ti][bi][x] in LDS, copied by LDS-DMA
//             (buffer_load_dwordx4 ... lds: the four rows bi of one (ai, ti) = 1 KB per instruction; no VGPRs, no VALU, no
//             ds_write; rows outside the image / the displacement range arrive as zeros from the buffer range check); the
//             ti and ai strides are padded so that the gather below is conflict-free (GL<FLIP>).  The G operand of (centre
//             block a, neighbour block pair j) is a GATHER from it: lane (pixel, k group) picks 8 values with
//             ds_read_b32 / b64 (one address register + immediates; slots outside the 21-wide band are read anyway and
//             replaced by zero) and splits them into the hi and lo f16 fragments in registers (an image element is gathered
//             ~1.1 times per u, so splitting here costs what splitting while staging would -- but the image does not pass
//             through the staging waves' registers and the LDS store path); gathered once per u, reused for the 4 channel
//             tiles.
//   X tile    4 neighbour rows x 64 pixels x 32 channels per chunk, split once while staging, in the forward kernel's LDS image
//             (8-byte chunks of 4 lattice columns, [term][parity][channel][...]) with the 16-byte units of a channel ordered
//             (block pair, row pair, block): the X operand of (channel tile, block pair) is two plain ds_read_b128 (hi, lo).
// 12 waves per workgroup, 3 per SIMD (168-register budget: 158 used, no scratch), specialised: waves 0-3 stage (the G DMA,
// buffer loads of X whose range check returns zeros outside the image, X split, LDS writes), waves 4-11 gather and run the
// MFMAs: matrix wave w takes x parity w&1 and the centre column blocks of role w>>1 ({0,3},{1,2},{4,7},{5,6}: 6 (block, pair)
// products each) x 4 channel tiles.  MFMA and VALU instructions of one SIMD do not overlap (scripts/ubench/mfma_valu_overlap),
// so a u costs the SIMD 144 MFMAs + ~650 VALU instructions: the kernel runs within ~25 % of that bound.
// Two barriers per u: [gather the 6 G operands | write both X chunks] [72 MFMAs | DMA of G(u+1)].
// Epilogue: accumulators -> LDS [channel][row][x] (16-byte slots rotated; over the X buffers, so that the G image of the
// workgroup's next task is already being filled) -> rows of 256 B, scaled by 1/C; outputs that came out non-finite (an
// operand did not fit an f16) are recomputed in plain fp32.
#include <type_traits>

#include "corr_params.h"
#include "f16x2_split.h"

namespace fn2 {
namespace hb {
using f16s::exp_stat;
using f16s::scale_exp;
using f16s::split2;
using f16s::to_sgpr;
using f16s::wave_sum;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define FN2_LDS(T) __attribute__((address_space(3))) T

// Cache policy of the gradient row stores: sc1 (agent scope) -- written through the XCD's L2 as they are issued.  The eight L2s
// are not coherent with each other, so everything a kernel wrote has to be in memory when it ends; left dirty (scope 0) the 50 MB
// of gradients are flushed in one burst at the end of the kernel, which the NEXT kernel of the stream waits for: inside bench.py's
// step sc1 measured 75.4 us against 77.0 for this kernel and 3 us less for the kernel after it (scripts/gpu_bench_ab.sh; the
// forward kernel's rows have had it since round 2: temporal stores cost it 3 % of the step).  The same hint on the gO DMA (GNT)
// costs 6 us: those rows ARE re-read, by the other channel groups and the other gradient.
#ifdef FN2_ABL_BWDTEMPORAL   // timing ablation
constexpr int BWD_STORE_AUX = 0;
#else
constexpr int BWD_STORE_AUX = 2;
#endif
#ifdef FN2_ABL_GNT
constexpr int G_LOAD_AUX = 2;
#else
constexpr int G_LOAD_AUX = 0;
#endif
constexpr int DR = 10, D = 21, NU = 6;
constexpr int CG = 64, NCT = CG / 16;             // channels per task, channel tiles of 16
constexpr int CK = 32;                            // channels per X chunk (2 tiles)
// X chunk image (bytes), as in correlation_f16x2.hip
constexpr int CHS = 288, PARS = CK * CHS, XTERM = 2 * PARS, XBUF = 2 * XTERM;   // 9216, 18432, 36864
// G image (bytes): [ai][ti][bi][x], fp32, x in natural pixel order: the four neighbour rows bi of one (ai, ti) are one
// contiguous KB -- one 16-byte-per-lane DMA instruction -- and the strides of ti and ai are free (the DMA's LDS base only needs
// 4-byte alignment: scripts/ubench/dma_align_probe.hip).  They are chosen so that the 32 lanes of a gather instruction -- 16
// centre pixels (ai, aj) x the two neighbour column blocks of a pair (k groups g = 0, 1: blk = g & 1) -- hit 32 distinct banks:
//   FLIP 0 (a lane's aj moves the displacement row down by one and the pixel by two): 8-byte gathers (pixel pair, the
//           wave's parity is picked in registers), units of 8 B mod 32: aj -> -3, ai -> +4, blk -> +16
//   FLIP 1 (aj moves the displacement row up by one, the pixel is the k slot's): 4-byte gathers, banks mod 32: aj -> +1, ai -> +8,
//           blk -> +4
template <int FLIP> struct GL {
    static constexpr int BI = 256;
    static constexpr int TI = FLIP ? 1024 + 4 : 1024 + 32;
    static constexpr int AI = FLIP ? D * TI + 76 : D * TI + 128;      // 21664 / 22304
    static constexpr int IMG = 3 * AI + D * TI;                         // 86580 / 89088
};
static_assert(GL<0>::TI % 256 == 32 && GL<0>::AI % 256 == 32 && GL<1>::TI % 128 == 4 && GL<1>::AI % 128 == 32, "gather bank pattern");
constexpr int GIMG = GL<0>::IMG;
static_assert(GL<1>::IMG <= GIMG && GIMG % 16 == 0, "G image");
constexpr int X_OFS = GIMG, ZERO_OFS = X_OFS + 2 * XBUF, LDS_BYTES = ZERO_OFS + 64;                          // 89088, 162816, 162880
constexpr int E_BYTES = CG * 4 * 64 * 4;          // epilogue image [64 channels][4 rows][64 x] floats, aliases the X buffers
static_assert(E_BYTES <= 2 * XBUF && LDS_BYTES <= 163840, "LDS budget");

struct Args {
    const float *nbr[2];   // [0] = in2 (neighbours for gradInput1), [1] = in1 (for gradInput2)
    const float *gout;
    float *gin[2];         // [0] = gradInput1, [1] = gradInput2
    int B, C, H, W;        // H even, W % 8 == 0, W <= 64, C % 64 == 0
    int NRG, NCGR;         // row groups per parity, channel groups
    int nflip, flip0;      // 2: both gradients in this launch; 1: only flip0
    float fC, rC;          // (float)C and 1 / C: kernel arguments so that they are SGPRs (no float SALU on gfx950)
    unsigned long long *dbg;   // profiling variant 64 only: s_memtime stamps (fn2_debug_set_buffer)
};

// centre column blocks of a wave role and the block pairs (2j, 2j+1) they meet
__host__ __device__ constexpr int a_blk(int role, int ab) { return role == 0 ? (ab ? 3 : 0) : role == 1 ? (ab ? 2 : 1) : role == 2 ? (ab ? 7 : 4) : (ab ? 6 : 5); }
__host__ __device__ constexpr bool meets(int a, int j) { return 2 * j + 1 >= a - 3 && 2 * j <= a + 3; }   // j in 0..3
__host__ __device__ constexpr int frag_idx(int role, int ab, int j)   // index among the role's (block, pair) products, or -1
{
    int idx = 0;
    for (int b = 0; b < 2; ++b)
        for (int jj = 0; jj < 4; ++jj) {
            if (b == ab && jj == j) return meets(a_blk(role, ab), j) ? idx : -1;
            if (meets(a_blk(role, b), jj)) ++idx;
        }
    return -1;
}
__host__ __device__ constexpr int frag_sub(int role, int ab, int j)   // index among the products of ONE centre block, or -1
{
    int idx = 0;
    for (int jj = 0; jj < 4; ++jj) {
        if (jj == j) return meets(a_blk(role, ab), j) ? idx : -1;
        if (meets(a_blk(role, ab), jj)) ++idx;
    }
    return -1;
}
constexpr int NF = 6;   // (centre block, block pair) products of a wave
static_assert(frag_idx(0, 1, 3) == 5 && frag_idx(1, 1, 3) == -1 && frag_idx(1, 1, 2) == 5 && frag_idx(2, 1, 3) == 5 && frag_idx(3, 1, 3) == 5,
              "6 products per role");

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

// one gradient element as a plain fp32 fma chain (cold path: outputs whose matrix-core result is non-finite)
__device__ __forceinline__ float exact_grad(const Args &p, int flip, int n, int c, int y, int x)
{
    const long HW = (long)p.H * p.W;
    const float *X = p.nbr[flip] + ((long)n * p.C + c) * HW;
    const float *g = p.gout + (long)n * D * D * HW;
    float s = 0.0f;
    for (int tj = 0; tj < D; ++tj)
        for (int ti = 0; ti < D; ++ti) {
            const int sgn = flip ? -1 : 1;
            const int yq = y + sgn * 2 * (tj - DR), xq = x + sgn * 2 * (ti - DR);   // the neighbour pixel
            if (yq < 0 || yq >= p.H || xq < 0 || xq >= p.W) continue;
            const long gp = flip ? (long)yq * p.W + xq : (long)y * p.W + x;          // the gO pixel
            s = fmaf(g[(long)(tj * D + ti) * HW + gp], X[(long)yq * p.W + xq], s);
        }
    return s;
}

constexpr int NSW = 4, NWAVES = NSW + 8;   // staging waves, waves per workgroup (3 per SIMD: 168 VGPRs each)
constexpr int XK = 32 / (2 * NSW);          // X items per chunk (32 channels) and staging lane
struct XSet { u4 v[XK][2]; };      // one X chunk of one lane: [slot][half] x 16 B (8 pixels)

// The lane id, formed where it is needed (two instructions) instead of an opaque copy of a variable that lives -- and may be
// spilled -- across the task loop (correlation_f16x2_bwd_wide.hip)
__device__ __forceinline__ int lane_now()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// VAR: profiling switches (0 = the real kernel): 1 no MFMA, 2 no global loads, 4 no stores, 8 no gathers / operand reads,
//      16 no split / LDS staging writes
template <int VAR>
__global__ __launch_bounds__(NWAVES * 64, 3) void corr_bwd_f16x2(Args p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    // the scale exponents (f16x2_split.h) of the task about to start: [0] = kx + kg (undone in the epilogue), [1] = kg (the matrix
    // waves scale the G operand).  Written by staging wave 0 before the last barrier of the previous task / barrier (A), read by
    // the matrix waves after it.
    __shared__ int scl_k[2];
    // ... and, since round 4, ONE EXPONENT PER CHANNEL for the X operand: scl_sx[task parity][channel of the task].  X is the A matrix of
    // the products (rows = channels): a per-row scale leaves through the rows of D, i.e. per gradient channel, exactly.  With one
    // exponent per task, channels 1000 x smaller than the task's typical magnitude kept 13 bits (their residual term went
    // f16-subnormal): tests/test_gpu_parity.py::test_correlation_backward_per_channel_error.  scl_k[0] is unused now.
    // The float scales 2^kx, ordered for the staging lanes: position 8 (c & 7) + (c >> 3), so that the eight channels a lane stages
    // (c = s_ch + 8 i) are 32 contiguous bytes -- two 16-byte reads at the top of a phase instead of a read and a wait per item;
    // the epilogue takes a channel's exponent out of the same word
    __shared__ __attribute__((aligned(16))) float scl_sx[2][CG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < NSW;
    const int w8 = is_stage ? wave : wave - NSW;   // staging wave 0 .. NSW-1 / matrix wave 0 .. 7
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int per_fn = 2 * p.NRG * p.NCGR;                  // tasks per (flip, batch item)
    const int ntasks = p.nflip * p.B * per_fn;
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int lgC = pow2 ? 31 - __builtin_clz((unsigned)p.C) : 0;
    if (tid < 16) reinterpret_cast<unsigned *>(smem + ZERO_OFS)[tid] = 0u;   // the zero words of the gathers (8-byte reads: FLIP 0)
    // VAR 64 (profiling): s_memtime stamps of wave 0 (staging) and the first matrix wave during the workgroup's first task
    unsigned long long ts[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ts[i] = 0;
    auto stamp = [&](int i) __attribute__((always_inline)) { if (VAR & 64) ts[i] = __builtin_amdgcn_s_memtime(); };
    auto dump = [&]() __attribute__((always_inline)) {
        if ((VAR & 64) && p.dbg && lane == 0 && (wave == 0 || wave == NSW)) {
            unsigned long long *d = p.dbg + (blockIdx.x * 2 + (wave ? 1 : 0)) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = ts[i];
        }
    };
    stamp(0);

    struct Task { int flip, n, py, rg, cg; };
    auto get_task = [&](int t) -> Task {
        Task k;
        // Task order: channel group fastest, then row group, then the GRADIENT, then (y parity, batch item): the 48 tasks that read
        // the gO planes of one (n, py) -- both gradients, every row group and channel group -- are consecutive, i.e. run on ONE XCD
        // (xcd_remap keeps consecutive tasks together) within two of its three rounds, and share its L2.  With the gradient as the
        // outermost index (rounds 2-3) the two gradients of an item ran 384 tasks apart and each fetched gO from the fabric again.
        k.cg = t % p.NCGR; t /= p.NCGR;
        k.rg = t % p.NRG; t /= p.NRG;
#ifdef FN2_ABL_FLIPOUTER   // A/B: the round-3 order
        k.py = t & 1; t >>= 1;
        k.n = t % p.B;
        k.flip = p.nflip == 2 ? t / p.B : p.flip0;
#else
        k.flip = p.nflip == 2 ? t % 2 : p.flip0;
        if (p.nflip == 2) t >>= 1;
        k.py = t & 1; t >>= 1;
        k.n = t;
#endif
        k.cg = __builtin_amdgcn_readfirstlane(k.cg); k.rg = __builtin_amdgcn_readfirstlane(k.rg);
        k.py = __builtin_amdgcn_readfirstlane(k.py); k.n = __builtin_amdgcn_readfirstlane(k.n);
        k.flip = __builtin_amdgcn_readfirstlane(k.flip);
        return k;
    };

    // ---- write-out of the epilogue image (all waves): 256 rows (channel, centre row) of 64 floats, 4 rows per instruction
    float *Es = reinterpret_cast<float *>(smem + X_OFS);
    // Laid out for few VALU instructions (the epilogue is issue-bound like the rest): a lane group g = lane >> 4 owns centre
    // row ai = g, wave w owns channels w, w + NWAVES, ... (scalars): the LDS offset is a lane part + 1 KB per channel, the global
    // row a buffer store with one lane offset and a scalar channel offset.
    // ksum = kx + kg: the matrix-core sums carry 2^ksum (the operand scales of the task); removed with the 1/C, exactly
    auto store_rows = [&](const Task &tk, int kg, int par) {   // kg: the task's G exponent; par: which half of scl_sx holds its X exponents
        int ln = lane_now();   // keeps the row geometry from being hoisted out of the task loop (and spilled)
        const int g = ln >> 4, xg = 4 * (ln & 15);
        constexpr int NRI = (CG + NWAVES - 1) / NWAVES;      // channels per wave (the last one partial)
        const int y = 2 * (4 * tk.rg + g) + tk.py;
        const bool lane_ok = 4 * tk.rg + g < HL && xg < p.W;
        const unsigned vo = lane_ok ? (unsigned)((y * p.W + xg) * 4) : 0x80000000u;   // out-of-range lanes store nothing
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.gin[tk.flip] + (long)tk.n * p.C * HW, 0, (unsigned)(p.C * HW * 4), 0x00020000);
        auto chan = [&](int i) { return wave + NWAVES * i; };
        auto read_row = [&](int c) {   // Es[c][ai = g][x], 16-byte slots rotated by 8 ai + 32 ((c >> 2) & 1)
            return *reinterpret_cast<const f4 *>(Es + (c * 4 + g) * 64 + ((xg + 8 * g + 32 * ((c >> 2) & 1)) & 63));
        };
        f4 vals[NRI];
#pragma unroll
        for (int i = 0; i < NRI; ++i) vals[i] = read_row(chan(i) & (CG - 1));
        // Scaling: v_ldexp_f32 by a scalar exponent -- 2^-ksum and, for a power-of-two C, the 1/C in one exact step.  A general C
        // is copied from the kernel arguments (SGPR) once per call, before the first store.  (As a VGPR value across the
        // task loop it gets spilled, and a scratch reload waits for vmcnt(0) -- for the acknowledgement of every row store
        // before it.  Copied by an asm statement BETWEEN the stores, the copy can land in a data register of the 16-byte store
        // just issued: the hardware needs a wait state there that the compiler does not insert for inline assembly.)
        float f = 1.0f;
        if (!pow2) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "s"(p.fC));
        const int kx_ex = -lgC;   // sums of the fp32 fallback (the matrix-core sums of channel c also carry 2^(kx[c] + kg))
        auto ksum_of = [&](int c) {
            const int cc = c & (CG - 1);
            const int bits = to_sgpr(__builtin_bit_cast(int, scl_sx[par][8 * (cc & 7) + (cc >> 3)]));
            return (bits >> 23) - 127 + kg;
        };
        auto scaled = [&](f4 val, int kx) {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = __builtin_ldexpf(val[e], kx);
            if (!pow2) { val[0] /= f; val[1] /= f; val[2] /= f; val[3] /= f; }
            return val;
        };
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < NRI; ++i) {
            const int c = chan(i);
            if (c >= CG) continue;                                    // uniform
            // inf / nan: an operand did not fit an f16 (class mask: sNaN, qNaN, -inf, +inf)
#ifdef FN2_ABL_GPRESPLIT
            if (false &&
#else
            if ((VAR & 31) == 0 && lane_ok &&
#endif
                (__builtin_amdgcn_classf(vals[i][0], 0x207) | __builtin_amdgcn_classf(vals[i][1], 0x207) |
                 __builtin_amdgcn_classf(vals[i][2], 0x207) | __builtin_amdgcn_classf(vals[i][3], 0x207)))
                bad |= 1u << i;
            if (!(VAR & 4))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, scaled(vals[i], -ksum_of(c) - lgC)), rso, (int)vo, (int)((tk.cg * CG + c) * HW * 4), BWD_STORE_AUX);
        }
        // Non-finite values (an operand beyond the f16 range): a second pass recomputes exactly those outputs with an fp32 fma
        // chain and stores the row again.  Kept out of the loop above: inlined there, its live state pushes the row values
        // into scratch.
        if (bad) {
#pragma unroll 1
            for (int i = 0; i < NRI; ++i) {
                if (!(bad >> i & 1)) continue;
                const int c = chan(i);
                const int ksum = ksum_of(c);
                f4 val = read_row(c);
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    const float cur = e == 0 ? val[0] : e == 1 ? val[1] : e == 2 ? val[2] : val[3];
                    const bool nonfin = (__builtin_bit_cast(unsigned, cur) & 0x7f800000u) == 0x7f800000u;
                    // (the finite entries take the matrix-core scale here: the whole row is finished as fallback values)
                    const float ex = nonfin ? exact_grad(p, tk.flip, tk.n, tk.cg * CG + c, y, xg + e) : __builtin_ldexpf(cur, -ksum);
                    val[0] = e == 0 ? ex : val[0]; val[1] = e == 1 ? ex : val[1];
                    val[2] = e == 2 ? ex : val[2]; val[3] = e == 3 ? ex : val[3];
                }
                *reinterpret_cast<f4 *>(p.gin[tk.flip] + (((long)tk.n * p.C + tk.cg * CG + c) * p.H + y) * p.W + xg) = scaled(val, kx_ex);
            }
        }
    };

    if (is_stage) {
        // ================= staging waves =================
        // G rows: staging wave w copies the 21 rows of planes 2w and 2w + 1 (plane = 4 ai + bi), one 256-byte row per DMA
        // instruction (lane = pixel x); LDS address and gO offset of a row are scalars.
        // X items (as the forward's tiles): slot k covers channels 16k .. 16k+15 of the chunk
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_row = (lane >> 2) & 3;
        const int s_ch = 2 * w8 + (lane >> 5);            // + 2 NSW k: slot k of a chunk
        const int s_x = 8 * s_piece;
        // within a channel: 16-byte unit 4 j + 2 gg + blk (block pair j, rows 2 gg .. 2 gg + 1 of block 2 j + blk): the four k groups
        // g = 2 gg + blk of an X operand are consecutive units
        const int w_ofs = s_ch * CHS + (s_piece >> 1) * 64 + (s_piece & 1) * 16 + (s_row >> 1) * 32 + (s_row & 1) * 8;
        const unsigned xbytes = (unsigned)(p.C * HW * 4), gbytes = (unsigned)(D * D * HW * 4);

        // G(u) of task tk -> LDS: staging wave w copies centre row ai = w, one DMA instruction per displacement column ti (lane =
        // (neighbour row bi, 16-byte piece)).  FLIP 0: tj = 4u + bi - ai, gO row = centre row ai.  FLIP 1: tj = 20 - 4u - bi + ai,
        // gO row = neighbour row bi.  Rows that do not exist get an out-of-range lane offset: the DMA writes zeros.
        auto g_dma = [&](const Task &tk, int u) {
            if (VAR & 2) return;
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gout + (long)tk.n * D * D * HW), 0, gbytes, 0x00020000);
            int ln = lane_now();
            const int bi = ln >> 4, pc = ln & 15;
#pragma unroll
            for (int k = 0; k < 4 / NSW; ++k) {
                const int ai = (4 / NSW) * w8 + k;
                const int tj = tk.flip ? 20 - 4 * u - bi + ai : 4 * u + bi - ai;
                const int il = tk.flip ? 4 * tk.rg - DR + 4 * u + bi : 4 * tk.rg + ai;
                const bool ok = tj >= 0 && tj < D && il >= 0 && il < HL && 4 * pc < p.W;
                const unsigned vo = ok ? (unsigned)(((tj * D * p.H + 2 * il + tk.py) * p.W + 4 * pc) * 4) : 0x80000000u;
                const int l0 = tk.flip ? ai * GL<1>::AI : ai * GL<0>::AI, lt = tk.flip ? GL<1>::TI : GL<0>::TI;
#pragma unroll
                for (int ti = 0; ti < D; ++ti)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (FN2_LDS(void) *)(smem + l0 + ti * lt), 16, (int)vo, ti * (int)(HW * 4), 0, G_LOAD_AUX);
            }
        };
        // X chunk (u, ch): neighbour rows 4rg - 10 + 4u .. +3, channels cg*64 + 32*ch .. +31; item k of a lane: channel 2 NSW k + s_ch
        auto x_issue1 = [&](XSet &L, const Task &tk, int u, int ch, int k) {
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.nbr[tk.flip] + (long)tk.n * p.C * HW), 0, xbytes, 0x00020000);
            const int il = 4 * tk.rg - DR + 4 * u + s_row;
            const bool ok = il >= 0 && il < HL && s_x < p.W;
            const unsigned vo = ok ? (unsigned)((s_ch * HW + (long)(2 * il + tk.py) * p.W + s_x) * 4) : 0x80000000u;
            const int soff = (int)((tk.cg * CG + ch * CK + 2 * NSW * k) * HW * 4);
            if (VAR & 2) { L.v[k][0] = (u4)(0x3c000000u + lane); L.v[k][1] = L.v[k][0]; return; }
            L.v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)vo, soff, 0);
            L.v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(vo + 16), soff, 0);
        };
        auto x_issue = [&](XSet &L, const Task &tk, int u, int ch) {
#pragma unroll
            for (int k = 0; k < XK; ++k) x_issue1(L, tk, u, ch, k);
        };
        // the current task's X scales: one per channel, i.e. per item of a lane (channel 32 ch + 8 k + s_ch): read from the LDS table
        // where they are used (as registers across the u loop they push the staging waves into scratch)
        auto x_write1 = [&](const XSet &L, char *buf, int k, float sc) {
            if (VAR & 16) { asm volatile("" ::"v"(L.v[k][0]), "v"(L.v[k][1])); return; }
#ifdef FN2_ABL_NOMUL
            const f4 x0 = __builtin_bit_cast(f4, L.v[k][0]), x1 = __builtin_bit_cast(f4, L.v[k][1]); (void)sc;
#else
            const f4 x0 = __builtin_bit_cast(f4, L.v[k][0]) * sc, x1 = __builtin_bit_cast(f4, L.v[k][1]) * sc;   // power of two: exact
#endif
            char *dst = buf + w_ofs + k * 2 * NSW * CHS;
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                unsigned h01, l01, h23, l23;
                split2(x0[par], x0[2 + par], h01, l01);
                split2(x1[par], x1[2 + par], h23, l23);
                *(FN2_LDS(u2) *)(dst + par * PARS) = (u2){h01, h23};
                *(FN2_LDS(u2) *)(dst + XTERM + par * PARS) = (u2){l01, l23};
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // the DMA's LDS writes are complete when its vector-memory counter has drained
        auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        // Operand sample of a task, straight from global memory, ONE 8-byte load per lane and operand = 128 values each (a load
        // costs the vector-memory path the same whatever its width; 16 bytes would cost the staging waves a scratch spill): X from
        // the neighbour rows 4rg - 2 .. 4rg + 1 (u = 2: always meets the image; lane -> channel and row, the column varies), G from
        // the gO image of the same u (the central displacement rows; lane -> (ai, bi), displacement column and pixel vary).
        // EVERY staging wave loads the same values and derives the same two exponents: no exchange, no barrier.  Requested at the
        // start of the previous task's last MFMA phase, evaluated while the matrix waves scatter that task's accumulators.
        constexpr int U0 = 2;
        struct Samp { u2 x, x2, g; };   // X: lane = channel of the task, 4 values of it (two rows, two places); G: as before
        auto sample_issue = [&](const Task &tk, Samp &S) {
            int ln = lane_now();
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.nbr[tk.flip] + (long)tk.n * p.C * HW), 0, xbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gout + (long)tk.n * D * D * HW), 0, gbytes, 0x00020000);
            const int ai = ln & 3, bi = (ln >> 2) & 3, q = ln >> 4;
            const int tj = tk.flip ? 20 - 4 * U0 - bi + ai : 4 * U0 + bi - ai;                 // as g_dma
            const int ilg = tk.flip ? 4 * tk.rg - DR + 4 * U0 + bi : 4 * tk.rg + ai;
            const int x = 2 * (((((5 * ln) >> 1) & 31) * (p.W >> 1)) >> 5);
            const int c = tk.cg * CG + ln, ilx = min(4 * tk.rg + (ln & 3), HL - 1);   // X sample rows: two of the centre rows, clamped into the
            // image -- every lane (= channel) gets four real values whatever the map's height (a lane without a sample would leave its
            // channel unscaled: scripts/soak_fuzz.py, H = 2)
            const int ti = (5 * q + (ln & 3) + bi) % D;
            const unsigned ox = (ilx >= 0 && ilx < HL) ? (unsigned)((c * HW + (long)(2 * ilx + tk.py) * p.W + x) * 4) : 0x80000000u;
            const int ilx2 = min(4 * tk.rg + ((ln + 2) & 3), HL - 1), xb = (x + (p.W >> 1)) >= p.W ? x + (p.W >> 1) - p.W : x + (p.W >> 1);
            const unsigned ox2 = (ilx2 >= 0 && ilx2 < HL) ? (unsigned)((c * HW + (long)(2 * ilx2 + tk.py) * p.W + xb) * 4) : 0x80000000u;
            const unsigned og = (ilg >= 0 && ilg < HL) ? (unsigned)((((tj * D + ti) * p.H + 2 * ilg + tk.py) * p.W + x) * 4) : 0x80000000u;
#ifdef FN2_ABL_NOSAMPLELOAD   // timing ablation
            S.x = (u2)0x3f000000u; S.x2 = S.x; S.g = (u2)0x3f000000u; (void)ox; (void)ox2; (void)og;
#else
            S.x = (VAR & 2) ? (u2)0x3f800000u : __builtin_amdgcn_raw_buffer_load_b64(rsx, (int)ox, 0, 0);
            S.x2 = (VAR & 2) ? (u2)0x3f800000u : __builtin_amdgcn_raw_buffer_load_b64(rsx, (int)ox2, 0, 0);
            S.g = (VAR & 2) ? (u2)0x3f800000u : __builtin_amdgcn_raw_buffer_load_b64(rsg, (int)og, 0, 0);
#endif
        };
        // kx: THIS LANE'S channel; kg: one exponent for the task, as before
        auto sample_scales = [&](const Samp &S, int &kx, int &kg) {
            // this lane's channel: the MEAN binary exponent of its non-zero samples lands at 2^T_GEO = 2^-1, the rule of
            // f16x2_split.h per channel (the mean, not the maximum: one outlier among the four must not push the channel's ordinary
            // values into the f16 subnormals; four samples put the mean within about a bit of the channel's)
            const unsigned t4 = exp_stat(S.x[0]) + exp_stat(S.x[1]) + exp_stat(S.x2[0]) + exp_stat(S.x2[1]);
            const unsigned sum = t4 & 0xffffu, cnt = t4 >> 16;
            const int em = cnt ? (int)((2u * sum + cnt) / (2u * cnt)) : 0;   // 1 .. 4 samples: a division by a small count
            const int k = f16s::T_GEO + 127 - em;
            kx = cnt == 0u ? 0 : (k < -126 ? -126 : k > 127 ? 127 : k);
            const unsigned tg = exp_stat(S.g[0]) + exp_stat(S.g[1]);
            kg = scale_exp(wave_sum(tg));
        };
        auto publish = [&](int par, int kx, int kg) {
            if (wave == 0) {
                int ln = lane_now();   // (the table address is formed here, not kept across the task loop: it would be spilled)
                scl_sx[par][8 * (ln & 7) + (ln >> 3)] = f16s::scale_from_exp(kx);
                if (ln == 0) scl_k[1] = kg;
            }
        };
        XSet XA0, XA1, XB0, XB1;
        int t = (int)xcd_remap(blockIdx.x, gridDim.x);
        Samp SM;
        int kx_n = 0, kg_n = 0;                                // the next task's scale exponents (kx: of this lane's sample channel)
        int it = 0;                                            // tasks done by this workgroup: parity selects the half of scl_sx
        if (t < ntasks) {
            const Task tk = get_task(t);
            sample_issue(tk, SM);
            x_issue(XA0, tk, 0, 0);
            x_issue(XA1, tk, 0, 1);
            g_dma(tk, 0);
            stamp(1);
            dma_wait();
            stamp(2);
            sample_scales(SM, kx_n, kg_n);
            publish(0, kx_n, kg_n);
        }
        __syncthreads();                                       // (A) G(0) complete, the first task's exponents published
        stamp(3);
        for (; t < ntasks; t += gridDim.x) {
            const Task tk = get_task(t);
            const bool has_next = t + (int)gridDim.x < ntasks;
            const Task tn = get_task(has_next ? t + (int)gridDim.x : t);
            const bool first = t < (int)gridDim.x;
            const int kg_cur = kg_n, par = it & 1;
            auto one_u = [&](int u, XSet &C0, XSet &C1, XSet &N0, XSet &N1) {
                // the scale of item k of chunk ch (channel s_ch + 8 (k + 4 ch)) is read from the table one item ahead of its use: all
                // eight of them in registers across the phase do not fit the staging waves' budget (168 with four X sets in flight)
                const float *sxp = &scl_sx[par][8 * s_ch];
                // phase 1 (the matrix waves gather the G operands of u): request the next X chunks, write both X chunks of u
                // (all loads first: interleaving them with the items of x_write measured 4 us slower)
                float sc = sxp[0];
                if (u + 1 < NU) { x_issue(N0, tk, u + 1, 0); x_issue(N1, tk, u + 1, 1); }
                else if (has_next) { x_issue(N0, tn, 0, 0); x_issue(N1, tn, 0, 1); }
#pragma unroll
                for (int k = 0; k < 2 * XK; ++k) {
                    const float nx = sxp[k + 1 < 2 * XK ? k + 1 : k];
                    if (k < XK) x_write1(C0, smem + X_OFS, k, sc);
                    else x_write1(C1, smem + X_OFS + XBUF, k - XK, sc);
                    sc = nx;
                }
                if (first && u < 2) stamp(4 + 4 * u);
                __syncthreads();                               // (B) the G image is free, the X chunks complete
                if (first && u < 2) stamp(5 + 4 * u);
                // phase 2 (all MFMAs of u): G(u+1), or G(0) of the next task, by DMA (and the next task's operand sample, ahead of it)
                if (u + 1 < NU) g_dma(tk, u + 1);
                else if (has_next) { sample_issue(tn, SM); g_dma(tn, 0); }
                dma_wait();
                if (first && u < 2) stamp(6 + 4 * u);
                __syncthreads();                               // (A') the X buffers are free, the next G image complete
                if (first && u < 2) stamp(7 + 4 * u);
            };
            for (int u = 0; u < NU; u += 2) {
                one_u(u, XA0, XA1, XB0, XB1);
                one_u(u + 1, XB0, XB1, XA0, XA1);
            }
            // while the matrix waves scatter their accumulators: the next task's scale exponents (every wave has read the current
            // ones: the matrix waves do at the top of the task)
            if (has_next) { sample_scales(SM, kx_n, kg_n); publish(par ^ 1, kx_n, kg_n); }
            if (first) stamp(12);
            __syncthreads();                                   // epilogue image (over the X buffers) complete
            if (first) stamp(13);
            store_rows(tk, kg_cur, par);
            __syncthreads();                                   // image read: the X buffers are free for the next task
            if (first) stamp(14);
            ++it;
        }
        stamp(15);
        dump();
        return;
    }

    // ================= matrix-core waves =================
    // No static priority for the matrix waves (the forward kernel has one): with it the staging wave of a SIMD only runs once
    // both matrix waves have finished gathering; without it the X stores overlap the gather (measured 1-1.5 us faster).
    if (VAR & 256) __builtin_amdgcn_s_setprio(2);      // 256 (profiling): with the priority
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);

    int itm = 0;   // tasks done by this workgroup (parity: which half of scl_sx holds the current task's X exponents)
    auto run_task = [&](const Task &tk, auto flipc, bool first) {
        constexpr int FLIP = decltype(flipc)::value;
        const int kg_cur = to_sgpr(scl_k[1]);                   // published before the barrier this wave just passed
        const f16s::scale2_t sc_g2 = f16s::scale2_from_exp(kg_cur);
        int ln = lane_now();
        const int f_i = ln & 15, f_g = ln >> 4;                 // pixel / channel index, k group
        const int f_ai = f_i >> 2, f_aj = f_i & 3;
        // X operand base: lane = (channel i, k group g): block 2j + (g&1), rows 2(g>>1), 2(g>>1)+1 -> the 16-byte unit 4j + g
        const int xb = xpar * PARS + f_i * CHS + f_g * 16;

        f4 acc[2][NCT];
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ab][ct] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        h8 gh[NF], gl[NF];   // the G operands of the wave's 6 (centre block, block pair) products

        // Gather of the G operands.  Slot s of k group g = neighbour (block m = 2j + blk, row bi = 2gg + (s>>2), column bj = s&3)
        // with blk = g&1, gg = g>>1, dm = m - a:
        //   FLIP 0: ti = 4 dm + bj - aj + 10, x = 2 (4a + aj) + par      FLIP 1: ti = 10 - 4 dm - bj + aj, x = 2 (4m + bj) + par
        // byte offset = ai AI + ti TI + bi BI + 4 x = lane part + slot part + (a, j) part; the lane part is recomputed in
        // every call from an opaque copy of the lane id (hoisted out of the u loop it would be 48 live addresses).  The slot
        // part is kept non-negative (ds_read immediates): FLIP 1 walks bj downwards from 3.  FLIP 0's layout spreads the 32 lanes
        // of a gather over 32 distinct 8-byte slots (pixel pairs); the wave reads the dword of its x parity in each -- 32 banks
        // of one parity, conflict-free per half wave like the 8-byte read of round 2, but two slots (s, s + 4) now arrive as one
        // ds_read2st64_b32 register pair.
        auto gather = [&](auto role_c, auto xp_c) {
            constexpr int R = decltype(role_c)::value;
            constexpr int XP = decltype(xp_c)::value;          // the wave's x parity, as a constant: part of the read's immediate offset
            typedef GL<FLIP> L;
            constexpr int SB = L::TI - 8;                                         // FLIP 1: one column less = one displacement row more
            int l2 = lane_now();
            const int ai = (l2 & 15) >> 2, aj = l2 & 3, blk = (l2 >> 4) & 1, gg = l2 >> 5;
            const int lbase = FLIP ? ai * L::AI + 2 * gg * L::BI + (DR - 4 * blk + aj) * L::TI - 3 * SB + 32 * blk + 4 * XP
                                   : ai * L::AI + 2 * gg * L::BI + (DR + 4 * blk - aj) * L::TI + 8 * aj;
            // band test: with sp = 4 blk + bj - aj, ti = 10 +- (4 dj + sp) lies in [0, 21) iff -10 - 4 dj <= sp <= 10 - 4 dj; sp is in
            // [-3, 7], so only one side can fail for a given dj.  Out-of-band slots are read anyway (any LDS address is harmless) and
            // replaced by zero: one compare of the lane value vs = 4 blk - aj with a constant and one select per slot.
            const int vs = 4 * blk - aj;
            static_for<0, 2>([&](auto abc) {
                constexpr int ab = decltype(abc)::value;
                constexpr int a = a_blk(R, ab);
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int fi = frag_idx(R, ab, j);
                    if constexpr (fi >= 0) {
                        constexpr int dj = 2 * j - a;                             // dm = dj + blk
                        constexpr int pconst = FLIP ? -4 * dj * L::TI + 64 * j : 4 * dj * L::TI + 32 * a;
                        constexpr bool check = dj < -1 || dj + 1 > 1;             // some slot may fall outside the 21-wide band
                        const int fbase = lbase + pconst;
                        // slots (s, s + 4) -- neighbour rows 2gg and 2gg + 1, one ds_read2st64_b32 -- share a register pair: the scale
                        // is one v_pk_mul_f32 per pair
                        f2 w[4];
                        static_for<0, 8>([&](auto sc) {
                            constexpr int s = decltype(sc)::value;
                            constexpr int bjs = s & 3, bis = s >> 2;
                            constexpr int sconst = FLIP ? bis * L::BI + (3 - bjs) * SB : bis * L::BI + bjs * L::TI;
                            const int ofs = fbase + sconst;
                            float v;
                            if (VAR & 8) v = 1.0f;
                            else if constexpr (FLIP) v = *reinterpret_cast<const float *>(smem + ofs);
#ifdef FN2_ABL_GATHER64   // round 2: 8-byte gather, element picked in registers
                            else v = (*reinterpret_cast<const f2 *>(smem + ofs))[XP];
#else
                            else v = *reinterpret_cast<const float *>(smem + ofs + 4 * XP);
#endif
                            if constexpr (check) {
                                constexpr int hi = 10 - 4 * dj - bjs, lo = -10 - 4 * dj - bjs;   // lo <= vs <= hi
                                if constexpr (hi < 4) v = vs <= hi ? v : 0.0f;
                                if constexpr (lo > -3) v = vs >= lo ? v : 0.0f;
                            }
                            w[s & 3][s >> 2] = v;
                        });
                        // two-term split in registers: slots (2q, 2q+1) -> one packed pair of each fragment
                        u4 vh, vl;
#ifdef FN2_ABL_GPRESPLIT   // timing ablation (results wrong): what a gO image that arrives pre-scaled and pre-split as (h | l << 16) words
                           // would cost here -- two v_perm_b32 per pair of slots instead of scale + split
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned a0 = __builtin_bit_cast(unsigned, w[(2 * q) & 3][q >> 1]), a1 = __builtin_bit_cast(unsigned, w[(2 * q + 1) & 3][q >> 1]);
                            vh[q] = __builtin_amdgcn_perm(a1, a0, 0x05040100u);
                            vl[q] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
                        }
#else
#pragma unroll
                        for (int q = 0; q < 4; ++q) w[q] = f16s::pk_scale(w[q], sc_g2);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {   // k slots (2q, 2q + 1) -> one packed pair of each fragment
                            unsigned hq, lq;
                            split2(w[(2 * q) & 3][q >> 1], w[(2 * q + 1) & 3][q >> 1], hq, lq);
                            vh[q] = hq; vl[q] = lq;
                        }
#endif
                        gh[fi] = __builtin_bit_cast(h8, vh);
                        gl[fi] = __builtin_bit_cast(h8, vl);
                        __builtin_amdgcn_sched_barrier(0);   // one operand at a time: 8 loads in flight
                    }
                });
            });
        };
        // All MFMAs of u: D[channel][pixel] += X[channel][q] * G[q][pixel].  The X operands of a block pair and two channel
        // tiles are read once and used by both centre blocks of the wave.
        auto mma = [&](auto role_c) {
            constexpr int R = decltype(role_c)::value;
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int f0 = frag_idx(R, 0, j), f1 = frag_idx(R, 1, j);
                if constexpr (f0 >= 0 || f1 >= 0) {
                    static_for<0, 2>([&](auto chc) {
                        constexpr int ch = decltype(chc)::value;
                        const char *buf = smem + X_OFS + ch * XBUF;
                        h8 xh[2], xl[2];
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            if (VAR & 8) { xh[c2] = (h8)((_Float16)1.0f); xl[c2] = xh[c2]; }
                            else {
                                xh[c2] = *reinterpret_cast<const h8 *>(buf + xb + c2 * 16 * CHS + j * 64);
                                xl[c2] = *reinterpret_cast<const h8 *>(buf + xb + c2 * 16 * CHS + j * 64 + XTERM);
                            }
                        }
                        if (VAR & 1) {
                            asm volatile("" ::"v"(xh[0]), "v"(xl[0]), "v"(xh[1]), "v"(xl[1]));
                        } else {
                            static_for<0, 3>([&](auto prc) {
                                constexpr int pr = decltype(prc)::value;
#pragma unroll
                                for (int c2 = 0; c2 < 2; ++c2) {
                                    if constexpr (f0 >= 0)
                                        acc[0][2 * ch + c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? xl[c2] : xh[c2], pr == 1 ? gl[f0 >= 0 ? f0 : 0] : gh[f0 >= 0 ? f0 : 0], acc[0][2 * ch + c2], 0, 0, 0);
                                    if constexpr (f1 >= 0)
                                        acc[1][2 * ch + c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? xl[c2] : xh[c2], pr == 1 ? gl[f1 >= 0 ? f1 : 0] : gh[f1 >= 0 ? f1 : 0], acc[1][2 * ch + c2], 0, 0, 0);
                                }
                            });
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            });
        };
        auto gather_x = [&](auto xp_c) {
            switch (role) {
            case 0: gather(std::integral_constant<int, 0>{}, xp_c); break;
            case 1: gather(std::integral_constant<int, 1>{}, xp_c); break;
            case 2: gather(std::integral_constant<int, 2>{}, xp_c); break;
            default: gather(std::integral_constant<int, 3>{}, xp_c); break;
            }
        };
        auto gather_d = [&]() {
            if (xpar) gather_x(std::integral_constant<int, 1>{});
            else gather_x(std::integral_constant<int, 0>{});
        };
        auto mma_d = [&]() {
            switch (role) {
            case 0: mma(std::integral_constant<int, 0>{}); break;
            case 1: mma(std::integral_constant<int, 1>{}); break;
            case 2: mma(std::integral_constant<int, 2>{}); break;
            default: mma(std::integral_constant<int, 3>{}); break;
            }
        };

        for (int u = 0; u < NU; ++u) {
            gather_d();                                        // phase 1
            if (first && u < 2) stamp(4 + 4 * u);
            __syncthreads();                                   // (B) both X chunks of u complete, the G image is free
            if (first && u < 2) stamp(5 + 4 * u);
            mma_d();                                           // phase 2
            if (first && u < 2) stamp(6 + 4 * u);
            __syncthreads();                                   // (A') the X buffers are free; G(u+1) complete
            if (first && u < 2) stamp(7 + 4 * u);
        }
        if (first) stamp(12);

        // epilogue: D[row = channel 4q + r][col = pixel i] -> Es[c][ai][x], 16-byte slots rotated by 8 ai + 32 ((c>>2)&1)
        auto scatter = [&](auto role_c) {
            constexpr int R = decltype(role_c)::value;
            static_for<0, 2>([&](auto abc) {
                constexpr int ab = decltype(abc)::value;
                constexpr int a = a_blk(R, ab);
                const int x = 8 * a + 2 * f_aj + xpar;
                static_for<0, NCT>([&](auto ctc) {
                    constexpr int ct = decltype(ctc)::value;
                    static_for<0, 4>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        const int c = 16 * ct + 4 * f_g + r;
                        Es[(c * 4 + f_ai) * 64 + ((x + 8 * f_ai + 32 * (f_g & 1)) & 63)] = acc[ab][ct][r];
                    });
                });
            });
        };
        switch (role) {
        case 0: scatter(std::integral_constant<int, 0>{}); break;
        case 1: scatter(std::integral_constant<int, 1>{}); break;
        case 2: scatter(std::integral_constant<int, 2>{}); break;
        default: scatter(std::integral_constant<int, 3>{}); break;
        }
        __syncthreads();
        if (first) stamp(13);
        store_rows(tk, kg_cur, itm & 1);
        __syncthreads();
        if (first) stamp(14);
    };
    __syncthreads();                                           // (A) G(0) of the first task complete, its scale exponents published
    stamp(3);
    for (int t = (int)xcd_remap(blockIdx.x, gridDim.x); t < ntasks; t += gridDim.x, ++itm) {
        const Task tk = get_task(t);
        const bool first = t < (int)gridDim.x;
        if (tk.flip) run_task(tk, std::integral_constant<int, 1>{}, first);
        else run_task(tk, std::integral_constant<int, 0>{}, first);
    }
    stamp(15);
    dump();
}

} // namespace hb

bool corr_bwd_f16x2_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    if (dtype != FN2_F32) return false;
    if (k != 1 || s1 != 1 || s2 != 2 || pad != md || md != 2 * hb::DR) return false;
    if (C % hb::CG != 0 || C < hb::CG || (H & 1) || (W % 8) != 0) return false;   // W > 64: correlation_f16x2_bwd_wide.hip
    if ((long)C * H * W * 4 >= 0x7fffffffL || (long)hb::D * hb::D * H * W * 4 >= 0x7fffffffL) return false;
    return true;
}

// which: 0 = both gradients, 1 = gradInput1 only, 2 = gradInput2 only; variant: profiling switches (fn2_debug.h)
int corr_backward_f16x2(const float *in1, const float *in2, const float *gout, float *g1, float *g2, int B, int C, int H, int W,
                        int variant, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(gout, 16) || !aligned(g1, 16) || !aligned(g2, 16)) return FN2_EALIGN;
    if (W > 64) return variant == 0 ? corr_backward_f16x2_wide(in1, in2, gout, g1, g2, B, C, H, W, s) : FN2_EINVAL;
    hb::Args a;
    a.nbr[0] = in2; a.nbr[1] = in1; a.gout = gout; a.gin[0] = g1; a.gin[1] = g2;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.NRG = (H / 2 + 3) / 4; a.NCGR = C / hb::CG;
    a.nflip = 2; a.flip0 = 0;
    a.fC = (float)C; a.rC = 1.0f / (float)C;
#ifdef FN2_DEBUG_BUILD
    a.dbg = (variant & 64) ? static_cast<unsigned long long *>(corr_f16x2_get_debug_buffer()) : nullptr;
#else
    a.dbg = nullptr;
#endif
    const long ntasks = 2L * B * 2 * a.NRG * a.NCGR;
    if (ntasks == 0) return FN2_OK;
    if (ntasks > 0x3fffffffL) return FN2_EINVAL;
    const unsigned grid = ntasks < 256 ? (unsigned)ntasks : 256u;   // persistent: one workgroup per CU
#define FN2_HB(V) case V: hipLaunchKernelGGL((hb::corr_bwd_f16x2<V>), dim3(grid), dim3(hb::NWAVES * 64), 0, s, a); return launch_status();
    switch (variant) {
        FN2_HB(0)
#ifdef FN2_DEBUG_BUILD   // profiling instantiations
        FN2_HB(1) FN2_HB(2) FN2_HB(4) FN2_HB(8) FN2_HB(16) FN2_HB(6) FN2_HB(24) FN2_HB(31) FN2_HB(64) FN2_HB(256) FN2_HB(65) FN2_HB(66) FN2_HB(68) FN2_HB(72) FN2_HB(80)
#endif
    default: return FN2_EINVAL;
    }
#undef FN2_HB
}

} // namespace fn2
