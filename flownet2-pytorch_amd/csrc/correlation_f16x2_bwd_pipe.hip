// correlation_f16x2_bwd_pipe.hip -- correlation backward (both input gradients) on the gfx950 f16 matrix cores: the banded
// contraction, tasks, operand split and epilogue of correlation_f16x2_bwd.hip, re-timed as a software pipeline of HALF steps.
//
// Replaces reference kernels correlation_backward_input1 / correlation_backward_input2 (correlation_cuda_kernel.cu:150-241,
// :243-334; one launch per batch item each, :522-554) for FlowNetC's configuration (kernel_size 1, stride1 1, stride2 2,
// pad == max_displacement == 20, fp32, maps up to 64 wide):
//     gI1[n,c,p] = (1/C) * sum_d gO[n, tc(d), p     ] * in2[n,c, p + 2d]
//     gI2[n,c,p] = (1/C) * sum_d gO[n, tc(d), p - 2d] * in1[n,c, p - 2d]          d in [-10,10]^2 (lattice units of 2 px)
// i.e. g[c, p] = sum_q X[c, q] * G[q, p] over the neighbours q of a centre pixel p on its parity lattice (FLIP 0: X = in2,
// G[q,p] = gO[q - p][p]; FLIP 1: X = in1, G[q,p] = gO[p - q][q]); M = 16 channels, K = two 4x4 neighbour blocks, N = a 4x4
// centre block; x = h + l two-term f16 split with one power-of-two scale per operand and task (f16x2_split.h), 3 MFMAs per product.
//
// What was wrong with the two-phase kernel (correlation_f16x2_bwd.hip): per neighbour row block u it alternates a gather phase
// (VALU-bound: the matrix pipes idle) and an MFMA phase (the vector ALUs idle, and the LDS-DMA of the next G image is the
// critical path), two barriers per u, because ONE G image (89 KB) and one X tile (73 KB) fill the LDS.  Here a step is HALF a u --
// the neighbour column-block pairs j = 2hh, 2hh + 1 of one u -- so that everything fits twice:
//   G half image  FLIP 1 (gO pixel = neighbour): the 32 neighbour pixels of the half, [ai pair][ti][bi][ai & 1][32 px]: one
//                 16-byte-per-lane DMA instruction = the 8 x 128 B segments (bi, ai & 1) of one (ai pair, ti).
//                 FLIP 0 (gO pixel = centre): the displacement columns ti a centre column block a meets in this half form a
//                 diagonal band in (a, ti); stored as [ai][bi][a][ti - T0(hh,a)][8 px]: 82 of the 168 (a, ti) cells, 3 DMA
//                 instructions per (ai, bi) with per-lane source offsets.
//                 Both layouts keep the 32 lanes of a gather instruction on 32 distinct banks.
//   X half tile   the units (block pairs) j = 2hh, 2hh + 1 of the unchanged X image: the two halves of a u are disjoint parts of it.
// One barrier per step.  During step s the matrix waves run the MFMAs of s (operands gathered during s - 1) and gather the
// operands of s + 1 from the OTHER G buffer -- gather VALU work now sits beside MFMAs (scripts/ubench/mfma_shadow.hip: a VALU
// instruction beside a saturated MFMA stream costs 2.6 cycles instead of 4.25) --, and, first thing in the step, issue the DMA
// of G(s + 2) into the buffer whose gather finished with the previous barrier (as inline assembly: the compiler must not make
// every later ds_read wait for it); the staging waves write X(s + 1) and request X(s + 2).  The DMA has a whole step to land.
// Role blocks {0,3} {1,2} {4,7} {5,6} have 4 + 2 or 2 + 4 (centre block, pair) products in the two halves; the two matrix waves
// of a SIMD are roles r and r + 2: 6 products per SIMD and step, as before per u.
// The G operands of the two halves live in disjoint register slots (4-lists in 0-3, 2-lists in 4-5): 48 registers, as before.
//
// STATUS (round 4): correct (bit-for-bit the sums of the two-phase kernel on every shape tried) and NOT faster: 80-87 us against
// 72-74 us back to back on the same box (8 x 256 x 48 x 64), 119 us on a slow box of the pool.  Compiled into the DEBUG library
// only (fn2_debug_correlation_backward variants 8000 + v; scripts/corr_micro.py prints its timeline); FN2_CORR_AUTO keeps the
// two-phase kernel.  What the s_memtime timelines and ablations of this kernel show:
//   - a half step takes 3.6 k ticks whether or not the MFMAs are there (variant 8065: 3.7 k): two chains of about that length
//     run side by side.  Staging: the CU's vector-memory path retires one 1 KB instruction per ~30-37 ticks (LDS-DMA and plain
//     16-byte loads alike; more for the FLIP 0 band, whose DMA instructions touch 32 cache lines each) -- 48 + 32 instructions
//     = 2.7 k ticks per half step, before any latency.  Matrix waves: 1.1-1.5 k ticks of MFMA + split work stretch to 2.2-3.0 k
//     once the operand reads are there: the LDS moves ~2.2 k cycles of traffic per half step (X fragment reads 1 k, gathers
//     0.4-0.8 k, X writes, DMA writes).
//   - i.e. per u the pipeline needs what the two-phase kernel needs (7 k), from the same two resources; overlapping the gather
//     with the MFMAs does not matter while those bound the step.  What would: fewer vector-memory instructions and fewer LDS
//     bytes per product (G amortised over 128 channels does not fit the LDS: the X tile doubles).
#ifdef FN2_DEBUG_BUILD
#include <type_traits>

#include "corr_params.h"
#include "f16x2_split.h"

namespace fn2 {
namespace hq {
using f16s::exp_stat;
using f16s::scale_exp;
using f16s::split2;
using f16s::to_sgpr;
using f16s::wave_sum;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define FN2_LDS(T) __attribute__((address_space(3))) T

constexpr int BWD_STORE_AUX = 2;   // sc1 row stores (correlation_f16x2_bwd.hip)
constexpr int DR = 10, D = 21, NU = 6, NS = 2 * NU;   // steps per task: (u, hh)
constexpr int CG = 64, NCT = CG / 16;
constexpr int CK = 32;
constexpr int CHS = 288, PARS = CK * CHS, XTERM = 2 * PARS, XBUF = 2 * XTERM;   // X image of a 32-channel chunk: 36864 B

// ---- FLIP 0 half image: cells (a, ti) of 32 B (the 8 pixels of centre column block a), block-major, per (ai, bi) plane
__host__ __device__ constexpr int T0(int hh, int a) { const int t = 16 * hh - 4 * a + 7; return t < 0 ? 0 : t; }
__host__ __device__ constexpr int T1(int hh, int a) { const int t = 16 * hh - 4 * a + 25; return t > 20 ? 20 : t; }
__host__ __device__ constexpr int NTI(int hh, int a) { const int n = T1(hh, a) - T0(hh, a) + 1; return n < 0 ? 0 : n; }
__host__ __device__ constexpr int CELL0(int hh, int a) { int c = 0; for (int i = 0; i < a; ++i) c += NTI(hh, i); return c; }
constexpr int NCELL = 82;
static_assert(CELL0(0, 8) == NCELL && CELL0(1, 8) == NCELL, "cells per plane");
struct G0 {
    static constexpr int BI = NCELL * 32;          // 2624
    static constexpr int AI = 4 * BI + 32;         // 10528: ai -> +32 B mod 256
    static constexpr int IMG = 4 * AI;             // 42112
};
// ---- FLIP 1 half image: [ai >> 1][ti][bi][ai & 1][32 px]
struct G1 {
    static constexpr int BI = 256, TI = 1028, SB = TI - 8;
    static constexpr int AP = 21600;               // bank pattern found by enumeration (32 lanes -> 32 banks)
    static constexpr int IMG = 2 * AP;             // 43200
};
constexpr int GBUF = 43200;
static_assert(G0::IMG <= GBUF && G1::IMG <= GBUF && GBUF % 16 == 0 && 20 * G1::TI + 1024 <= G1::AP, "G half image");
constexpr int X_OFS = 2 * GBUF, ZERO_OFS = X_OFS + 2 * XBUF, LDS_BYTES = ZERO_OFS + 64;   // 86400, 160128, 160192
constexpr int E_BYTES = CG * 4 * 64 * 4;
static_assert(E_BYTES <= 2 * XBUF && LDS_BYTES <= 163840 - 16, "LDS budget");

struct Args {
    const float *nbr[2];   // [0] = in2 (neighbours for gradInput1), [1] = in1 (for gradInput2)
    const float *gout;
    float *gin[2];
    long gbs;              // elements between batch items of gout
    int B, C, H, W;
    int NRG, NCGR;
    int nflip, flip0;
    float fC, rC;
    unsigned long long *dbg;
};

__host__ __device__ constexpr int a_blk(int role, int ab) { return role == 0 ? (ab ? 3 : 0) : role == 1 ? (ab ? 2 : 1) : role == 2 ? (ab ? 7 : 4) : (ab ? 6 : 5); }
__host__ __device__ constexpr bool meets(int a, int j) { return 2 * j + 1 >= a - 3 && 2 * j <= a + 3; }
// the (centre block ab, pair j) products of a role in half hh, ordered (j, ab): number and the i-th one
__host__ __device__ constexpr int nfrag(int role, int hh)
{
    int n = 0;
    for (int j = 2 * hh; j < 2 * hh + 2; ++j)
        for (int ab = 0; ab < 2; ++ab) n += meets(a_blk(role, ab), j) ? 1 : 0;
    return n;
}
__host__ __device__ constexpr int frag_j(int role, int hh, int i)
{
    int n = 0;
    for (int j = 2 * hh; j < 2 * hh + 2; ++j)
        for (int ab = 0; ab < 2; ++ab)
            if (meets(a_blk(role, ab), j)) { if (n == i) return j; ++n; }
    return -1;
}
__host__ __device__ constexpr int frag_ab(int role, int hh, int i)
{
    int n = 0;
    for (int j = 2 * hh; j < 2 * hh + 2; ++j)
        for (int ab = 0; ab < 2; ++ab)
            if (meets(a_blk(role, ab), j)) { if (n == i) return ab; ++n; }
    return -1;
}
// register slot of product i of (role, hh): 4-lists use slots 0-3, 2-lists 4-5 -- the lists of consecutive steps never share a slot
__host__ __device__ constexpr int frag_slot(int role, int hh, int i) { return nfrag(role, hh) == 4 ? i : 4 + i; }
static_assert(nfrag(0, 0) == 4 && nfrag(0, 1) == 2 && nfrag(1, 0) == 4 && nfrag(1, 1) == 2 && nfrag(2, 0) == 2 && nfrag(2, 1) == 4 &&
              nfrag(3, 0) == 2 && nfrag(3, 1) == 4, "4 + 2 products per role");

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

__device__ __forceinline__ float exact_grad(const Args &p, int flip, int n, int c, int y, int x)
{
    const long HW = (long)p.H * p.W;
    const float *X = p.nbr[flip] + ((long)n * p.C + c) * HW;
    const float *g = p.gout + (long)n * p.gbs;
    float s = 0.0f;
    for (int tj = 0; tj < D; ++tj)
        for (int ti = 0; ti < D; ++ti) {
            const int sgn = flip ? -1 : 1;
            const int yq = y + sgn * 2 * (tj - DR), xq = x + sgn * 2 * (ti - DR);
            if (yq < 0 || yq >= p.H || xq < 0 || xq >= p.W) continue;
            const long gp = flip ? (long)yq * p.W + xq : (long)y * p.W + x;
            s = fmaf(g[(long)(tj * D + ti) * HW + gp], X[(long)yq * p.W + xq], s);
        }
    return s;
}

constexpr int NSW = 4, NWAVES = NSW + 8;
constexpr int XK = 4;               // X items (2 x 16 B = 8 pixels of one row and channel) per staging lane and half step
struct XSet { u4 v[XK][2]; };

// buffer resource of a tensor slice, as four SGPRs (for the inline-assembly DMA)
__device__ __forceinline__ u4 make_rs(const void *ptr, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)ptr;
    u4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
// LDS-DMA of 16 bytes per lane: global (rs + voff + soff) -> LDS (lds + 16 * lane); out-of-range lanes write zeros.  Inline
// assembly so that the compiler does not order later LDS reads behind it (vmcnt is waited for by hand before the step's barrier)
__device__ __forceinline__ void dma16(u4 rs, unsigned lds, int voff, int soff)
{
    // (an "s" operand the compiler believes divergent is silently given a VGPR: the scalars are made uniform explicitly)
    const unsigned lds_s = __builtin_amdgcn_readfirstlane(lds);
    const int soff_s = __builtin_amdgcn_readfirstlane(soff);
    u4 rs_s;
    rs_s[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs_s[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs_s[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs_s[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_s), "v"(voff), "s"(rs_s), "s"(soff_s) : "memory");
}

// VAR: profiling switches (0 = the real kernel): 1 no MFMA, 2 no global loads / DMA, 4 no stores, 8 no gathers / operand reads,
//      16 no split / LDS staging writes, 64 s_memtime stamps
template <int VAR>
__global__ __launch_bounds__(NWAVES * 64, 3) void corr_bwd_pipe(Args p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    __shared__ int scl_k[2];   // [0] = kx + kg, [1] = kg of the task about to start (published by staging wave 0)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < NSW;
    const int w8 = is_stage ? wave : wave - NSW;
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int per_fn = 2 * p.NRG * p.NCGR;
    const int ntasks = p.nflip * p.B * per_fn;
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int lgC = pow2 ? 31 - __builtin_clz((unsigned)p.C) : 0;
    const unsigned lds0 = (unsigned)(unsigned long)(FN2_LDS(char) *)smem;
    if (tid < 16) reinterpret_cast<unsigned *>(smem + ZERO_OFS)[tid] = 0u;
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
        k.cg = t % p.NCGR; t /= p.NCGR;
        k.rg = t % p.NRG; t /= p.NRG;
        k.py = t & 1; t >>= 1;
        k.n = t % p.B;
        k.flip = p.nflip == 2 ? t / p.B : p.flip0;
        k.cg = __builtin_amdgcn_readfirstlane(k.cg); k.rg = __builtin_amdgcn_readfirstlane(k.rg);
        k.py = __builtin_amdgcn_readfirstlane(k.py); k.n = __builtin_amdgcn_readfirstlane(k.n);
        k.flip = __builtin_amdgcn_readfirstlane(k.flip);
        return k;
    };

    // ---- write-out of the epilogue image (all waves), as in correlation_f16x2_bwd.hip
    float *Es = reinterpret_cast<float *>(smem + X_OFS);
    auto store_rows = [&](const Task &tk, int ksum) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 4, xg = 4 * (ln & 15);
        constexpr int NRI = (CG + NWAVES - 1) / NWAVES;
        const int y = 2 * (4 * tk.rg + g) + tk.py;
        const bool lane_ok = 4 * tk.rg + g < HL && xg < p.W;
        const unsigned vo = lane_ok ? (unsigned)((y * p.W + xg) * 4) : 0x80000000u;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.gin[tk.flip] + (long)tk.n * p.C * HW, 0, (unsigned)(p.C * HW * 4), 0x00020000);
        auto chan = [&](int i) { return wave + NWAVES * i; };
        auto read_row = [&](int c) {
            return *reinterpret_cast<const f4 *>(Es + (c * 4 + g) * 64 + ((xg + 8 * g + 32 * ((c >> 2) & 1)) & 63));
        };
        f4 vals[NRI];
#pragma unroll
        for (int i = 0; i < NRI; ++i) vals[i] = read_row(chan(i) & (CG - 1));
        float f = 1.0f;
        if (!pow2) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "s"(p.fC));
        const int kx_mm = -ksum - lgC, kx_ex = -lgC;
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
            if (c >= CG) continue;
            if ((VAR & 31) == 0 && lane_ok &&
                (__builtin_amdgcn_classf(vals[i][0], 0x207) | __builtin_amdgcn_classf(vals[i][1], 0x207) |
                 __builtin_amdgcn_classf(vals[i][2], 0x207) | __builtin_amdgcn_classf(vals[i][3], 0x207)))
                bad |= 1u << i;
            if (!(VAR & 4))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, scaled(vals[i], kx_mm)), rso, (int)vo, (int)((tk.cg * CG + c) * HW * 4), BWD_STORE_AUX);
        }
        if (bad) {
#pragma unroll 1
            for (int i = 0; i < NRI; ++i) {
                if (!(bad >> i & 1)) continue;
                const int c = chan(i);
                f4 val = read_row(c);
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    const float cur = e == 0 ? val[0] : e == 1 ? val[1] : e == 2 ? val[2] : val[3];
                    const bool nonfin = (__builtin_bit_cast(unsigned, cur) & 0x7f800000u) == 0x7f800000u;
                    const float ex = nonfin ? exact_grad(p, tk.flip, tk.n, tk.cg * CG + c, y, xg + e) : __builtin_ldexpf(cur, -ksum);
                    val[0] = e == 0 ? ex : val[0]; val[1] = e == 1 ? ex : val[1];
                    val[2] = e == 2 ? ex : val[2]; val[3] = e == 3 ? ex : val[3];
                }
                *reinterpret_cast<f4 *>(p.gin[tk.flip] + (((long)tk.n * p.C + tk.cg * CG + c) * p.H + y) * p.W + xg) = scaled(val, kx_ex);
            }
        }
    };

    const unsigned xbytes = (unsigned)(p.C * HW * 4), gbytes = (unsigned)(D * D * HW * 4);

    if (is_stage) {
        // ================= staging waves: the X half tiles, the G DMA and the operand sample =================
        // Raised priority: a staging wave shares its SIMD with two matrix waves; what it issues (loads, DMA) is what the NEXT steps
        // wait for, so it goes first and the matrix waves fill the rest of the step.
        if (!(VAR & 256)) __builtin_amdgcn_s_setprio(3);
        // lane = (column block of the half pc, row, channel chs); item k of a half step: channel 16 k + 4 w8 + chs of the 64
        const int s_pc = lane & 3, s_row = (lane >> 2) & 3, s_chs = lane >> 4;
        // 16-byte unit 4 j + 2 gg + blk of the channel row (block m = 2 j + blk, rows 2 gg, 2 gg + 1), 8 bytes per row: a 16-lane
        // group (pc, row) covers a 128-byte window
        const int w_lane = s_chs * CHS + (s_pc >> 1) * 64 + (s_pc & 1) * 16 + (s_row >> 1) * 32 + (s_row & 1) * 8;
        struct XCtx { u4 rs; unsigned vo; int soff0; };
        auto x_ctx = [&](const Task &tk, int u, int hh) {
            XCtx c;
            c.rs = make_rs(p.nbr[tk.flip] + (long)tk.n * p.C * HW, xbytes);
            const int il = 4 * tk.rg - DR + 4 * u + s_row;
            const int x = 8 * (4 * hh + s_pc);
            const bool ok = il >= 0 && il < HL && x < p.W;
            c.vo = ok ? (unsigned)((s_chs * HW + (long)(2 * il + tk.py) * p.W + x) * 4) : 0x80000000u;
            c.soff0 = (int)((tk.cg * CG + 4 * w8) * HW * 4);
            return c;
        };
        auto x_issue1 = [&](XSet &L, const XCtx &c, int k) {
            const int soff = __builtin_amdgcn_readfirstlane(c.soff0 + 16 * k * (int)(HW * 4));
            u4 rs;   // (see dma16: scalars are made uniform explicitly)
            rs[0] = __builtin_amdgcn_readfirstlane(c.rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(c.rs[1]);
            rs[2] = __builtin_amdgcn_readfirstlane(c.rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(c.rs[3]);
            if (VAR & 2) { L.v[k][0] = (u4)(0x3c000000u + lane); L.v[k][1] = L.v[k][0]; return; }
            // inline assembly: every vector-memory instruction of the step loop is placed and waited for by hand (x_ready)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(L.v[k][0]) : "v"(c.vo), "s"(rs), "s"(soff) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(L.v[k][1]) : "v"(c.vo), "s"(rs), "s"(soff) : "memory");
        };
        auto x_issue = [&](XSet &L, const Task &tk, int u, int hh) {
            const XCtx c = x_ctx(tk, u, hh);
#pragma unroll
            for (int k = 0; k < XK; ++k) x_issue1(L, c, k);
        };
        // the data of a set are there once at most `cnt` younger vector-memory instructions are outstanding; the registers are tied to
        // the wait so that no use of them is scheduled above it
        auto x_ready = [&](XSet &L, auto cntc) {
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(L.v[0][0]), "+v"(L.v[0][1]), "+v"(L.v[1][0]), "+v"(L.v[1][1]), "+v"(L.v[2][0]), "+v"(L.v[2][1]), "+v"(L.v[3][0]), "+v"(L.v[3][1])
                         : "n"(decltype(cntc)::value)
                         : "memory");
        };
        auto x_write1 = [&](const XSet &L, int k, int hh, f16s::scale2_t sc_x) {
            if (VAR & 16) { asm volatile("" ::"v"(L.v[k][0]), "v"(L.v[k][1])); return; }
            const f4 x0 = f16s::pk_scale4(__builtin_bit_cast(f4, L.v[k][0]), sc_x), x1 = f16s::pk_scale4(__builtin_bit_cast(f4, L.v[k][1]), sc_x);
            // channel 16 k + 4 w8 + chs: chunk k >> 1, channel (16 (k & 1) + 4 w8) + chs of the chunk
            char *dst = smem + X_OFS + (k >> 1) * XBUF + (16 * (k & 1) + 4 * w8) * CHS + w_lane + hh * 128;
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
        auto x_write = [&](const XSet &L, int hh, f16s::scale2_t sc_x) {
#pragma unroll
            for (int k = 0; k < XK; ++k) x_write1(L, k, hh, sc_x);
        };
        // operand sample of a task (f16x2_split.h; as in correlation_f16x2_bwd.hip): X from the neighbour rows of u = 2, G from the gO
        // image of the same u; every staging wave loads the same values and derives the same two exponents
        constexpr int U0 = 2;
        struct Samp { u2 x, g; };
        auto sample_issue = [&](const Task &tk, Samp &S) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const u4 rsx4 = make_rs(p.nbr[tk.flip] + (long)tk.n * p.C * HW, xbytes);
            const u4 rsg4 = make_rs(p.gout + (long)tk.n * p.gbs, gbytes);
            const int ai = ln & 3, bi = (ln >> 2) & 3, q = ln >> 4;
            const int tj = tk.flip ? 20 - 4 * U0 - bi + ai : 4 * U0 + bi - ai;
            const int ilg = tk.flip ? 4 * tk.rg - DR + 4 * U0 + bi : 4 * tk.rg + ai;
            const int x = 2 * (((((5 * ln) >> 1) & 31) * (p.W >> 1)) >> 5);
            const int c = tk.cg * CG + ln, ilx = 4 * tk.rg - DR + 4 * U0 + (ln & 3);
            const int ti = (5 * q + (ln & 3) + bi) % D;
            const unsigned ox = (ilx >= 0 && ilx < HL) ? (unsigned)((c * HW + (long)(2 * ilx + tk.py) * p.W + x) * 4) : 0x80000000u;
            const unsigned og = (ilg >= 0 && ilg < HL) ? (unsigned)((((tj * D + ti) * p.H + 2 * ilg + tk.py) * p.W + x) * 4) : 0x80000000u;
            if (VAR & 2) { S.x = (u2)0x3f800000u; S.g = (u2)0x3f800000u; return; }
            const int zero = __builtin_amdgcn_readfirstlane(0);
            asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(S.x) : "v"(ox), "s"(rsx4), "s"(zero) : "memory");
            asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(S.g) : "v"(og), "s"(rsg4), "s"(zero) : "memory");
        };
        auto sample_scales = [&](Samp &S, int &kx, int &kg) {
            asm volatile("" : "+v"(S.x), "+v"(S.g));   // (requested two steps earlier; every wait since then covers them)
            const unsigned tx = exp_stat(S.x[0]) + exp_stat(S.x[1]), tg = exp_stat(S.g[0]) + exp_stat(S.g[1]);
            kx = scale_exp(wave_sum(tx));
            kg = scale_exp(wave_sum(tg));
        };
        auto publish = [&](int kx, int kg) { if (tid == 0) { scl_k[0] = kx + kg; scl_k[1] = kg; } };

        // ---- DMA of the G half image of (task, u, hh) into G buffer `buf` (the 4 staging waves share it)
        // FLIP 0: per-lane source offsets of the 3 instructions of a plane, both halves (kernel constants): piece q = 64 i + lane ->
        // cell q >> 1 = (a, ti), 16-byte half q & 1
        int v0[2][3];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = 64 * i + lane, cell = q >> 1;
                int a = 0, tiv = 0;
#pragma unroll
                for (int aa = 0; aa < 8; ++aa)
                    if (NTI(hh, aa) > 0 && cell >= CELL0(hh, aa) && cell < CELL0(hh, aa) + NTI(hh, aa)) { a = aa; tiv = T0(hh, aa) + cell - CELL0(hh, aa); }
                const int x = 8 * a + 4 * (q & 1);
                v0[hh][i] = (q < 2 * NCELL && x < p.W) ? (int)((tiv * HW + x) * 4) : (int)0x80000000u;
            }
        auto g_dma = [&](const Task &tk, int u, int hh, int buf) {
            if (VAR & 2) return;
            const float *gbase = p.gout + (long)tk.n * p.gbs;
            const unsigned dst = lds0 + buf * GBUF;
            if (tk.flip == 0) {
                // staging wave w8: the four planes (ai = w8, bi); gO row = centre row ai, tj = 4u + bi - ai
                const int ai = w8;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int bi = k;
                    const int tj = 4 * u + bi - ai, il = 4 * tk.rg + ai;
                    const bool ok = tj >= 0 && tj < D && il < HL;
                    const u4 rs = make_rs(gbase, ok ? gbytes : 0u);
                    const int soff = ok ? (int)(((long)tj * D * p.H + 2 * il + tk.py) * p.W * 4) : 0;
                    const unsigned d = dst + ai * G0::AI + bi * G0::BI;
                    const int va = hh ? v0[1][0] : v0[0][0], vb = hh ? v0[1][1] : v0[0][1], vc = hh ? v0[1][2] : v0[0][2];
                    dma16(rs, d, va, soff);
                    dma16(rs, d + 1024, vb, soff);
                    if (lane < 2 * NCELL - 128) dma16(rs, d + 2048, vc, soff);
                }
            } else {
                // staging wave w8: ai pair al = w8 >> 1, displacement columns ti = (w8 & 1) + 2 k; lane = (bi, e = ai & 1, 16-byte piece)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int al = w8 >> 1, bi = ln >> 4, e = (ln >> 3) & 1, pc = ln & 7;
                const int tj = 20 - 4 * u - bi + 2 * al + e, il = 4 * tk.rg - DR + 4 * u + bi;
                const int x = 32 * hh + 4 * pc;
                const bool ok = tj >= 0 && tj < D && il >= 0 && il < HL && x < p.W;
                const int vo = ok ? (int)((((long)tj * D * p.H + 2 * il + tk.py) * p.W + x) * 4) : (int)0x80000000u;
                const u4 rs = make_rs(gbase, gbytes);
                const unsigned d = dst + al * G1::AP;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const int ti = (w8 & 1) + 2 * k;
                    if (ti < D) dma16(rs, d + ti * G1::TI, vo, ti * (int)(HW * 4));
                }
            }
        };

        XSet S0, S1;   // steps of even / odd index
        int t = (int)xcd_remap(blockIdx.x, gridDim.x);
        Samp SM;
        int kx_n = 0, kg_n = 0;
        if (t < ntasks) {
            const Task tk = get_task(t);
            sample_issue(tk, SM);
            x_issue(S0, tk, 0, 0);
            g_dma(tk, 0, 0, 0);
            g_dma(tk, 0, 1, 1);
            x_issue(S1, tk, 0, 1);
            x_ready(S0, std::integral_constant<int, 0>{});     // everything requested so far is there: G(0), G(1) landed
            x_ready(S1, std::integral_constant<int, 0>{});
            sample_scales(SM, kx_n, kg_n);
            publish(kx_n, kg_n);
            x_write(S0, 0, f16s::scale2_from_exp(kx_n));
        }
        stamp(1);
        __syncthreads();                                       // (P) G(0) landed, X(0) written, the first task's exponents published
        __syncthreads();                                       // (Q) the matrix waves have gathered the operands of step 0
        stamp(2);
        for (; t < ntasks; t += gridDim.x) {
            const Task tk = get_task(t);
            const bool has_next = t + (int)gridDim.x < ntasks;
            const Task tn = get_task(has_next ? t + (int)gridDim.x : t);
            const bool first = t < (int)gridDim.x;
            const int ksum = kx_n + kg_n;
            const f16s::scale2_t sc_x = f16s::scale2_from_exp(kx_n);
            // step s = (u, hh): write X(s + 1) (its loads were requested a step ago), request X(s + 2)
            auto one_step = [&](int s, XSet &C, XSet &N) {   // C: the set that holds X(s + 1); N: free, takes X(s + 2)
                // The vector-memory path of the CU is the bottleneck of a step (75 KB at ~37 B/clk) and a wave stalls at every issue
                // while its queue is full: the DMA of G(s + 2) goes first (into the buffer whose operands were gathered before the last
                // barrier; it has to land within this step), then the items of X(s + 1) (requested a whole step ago) are split and
                // written BETWEEN the load pairs of X(s + 2) (whose set was written out a step ago), so the splits fill the stalls.
                const bool more = s + 2 < NS || has_next;
                const bool wr = s + 1 < NS;
                if (wr) x_ready(C, std::integral_constant<int, 0>{});
                if (s + 2 < NS) g_dma(tk, (s + 2) >> 1, s & 1, s & 1);
                else if (has_next) g_dma(tn, 0, s & 1, s & 1);
                if (first && (s == 2 || s == 3)) stamp(4 * s - 5);          // 3 / 7: DMA issued
                Task tx = tk;    // the task X(s + 2) belongs to (field by field: a select of whole structs goes through scratch memory,
                int ux = (s + 2) >> 1;   // and scratch accesses use the same counter as the loads that are waited for by hand here)
                if (s + 2 >= NS) { tx.flip = tn.flip; tx.n = tn.n; tx.py = tn.py; tx.rg = tn.rg; tx.cg = tn.cg; ux = 0; }
                const XCtx xc = x_ctx(tx, ux, s & 1);
#pragma unroll
                for (int k = 0; k < XK; ++k) {
                    if (wr) x_write1(C, k, (s + 1) & 1, sc_x);
                    if (more) x_issue1(N, xc, k);
                }
                if (s == 8 && has_next) sample_issue(tn, SM);
                if (s == 10 && has_next) { sample_scales(SM, kx_n, kg_n); publish(kx_n, kg_n); }
                if (first && (s == 2 || s == 3)) stamp(4 * s - 4);          // 4 / 8: X(s + 1) written, X(s + 2) requested
                if (more && !(VAR & 2) && !(s == 8 && has_next)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XK) : "memory");
                else if (more && !(VAR & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XK + 2) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (first && (s == 2 || s == 3)) stamp(4 * s - 3);          // 5 / 9: DMA landed
                __syncthreads();
                if (first && (s == 1 || s == 2 || s == 3)) stamp(4 * s - 2);   // 2 (overwrites Q) / 6 / 10: barrier
            };
            for (int s = 0; s < NS; s += 2) {
                one_step(s, S1, S0);
                one_step(s + 1, S0, S1);
            }
            if (first) stamp(12);
            __syncthreads();                                   // epilogue image (over the X tile) complete
            store_rows(tk, ksum);
            if (first) stamp(13);
            __syncthreads();                                   // image read: the X tile is free
            if (has_next) x_write(S0, 0, f16s::scale2_from_exp(kx_n));   // X(0) of the next task (requested during step 10)
            __syncthreads();                                   // (Q')
            if (first) stamp(14);
        }
        stamp(15);
        dump();
        return;
    }

    // ================= matrix-core waves =================
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);

    h8 gh[6], gl[6];   // G operands: slots 0-3 hold the 4-list of a (role, half), 4-5 the 2-list
    f4 acc[2][NCT];

    // Gather of product i of (role R, half HH) from G buffer BUF into its register slot.  Slot s of k group g = neighbour (block
    // m = 2j + blk, row bi = 2gg + (s >> 2), column bj = s & 3), blk = g & 1, gg = g >> 1, dj = 2j - a:
    //   FLIP 0: ti = 10 + 4 (dj + blk) + bj - aj, pixel = centre (4a + aj)      FLIP 1: ti = 10 - 4 (dj + blk) - bj + aj, pixel = neighbour
    auto gather1 = [&](auto flipc, auto role_c, auto hh_c, auto buf_c, auto ic, f16s::scale2_t sc_g2) {
        constexpr int FLIP = decltype(flipc)::value, R = decltype(role_c)::value;
        const int XP = xpar;
        constexpr int HH = decltype(hh_c)::value, BUF = decltype(buf_c)::value, I = decltype(ic)::value;
        constexpr int j = frag_j(R, HH, I), ab = frag_ab(R, HH, I), a = a_blk(R, ab), fi = frag_slot(R, HH, I);
        constexpr int dj = 2 * j - a;
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int ai = (l2 & 15) >> 2, aj = l2 & 3, blk = (l2 >> 4) & 1, gg = l2 >> 5;
        int fbase;
        if constexpr (FLIP) {
            const int lbase = (ai >> 1) * G1::AP + (ai & 1) * 128 + 2 * gg * G1::BI + (DR - 4 * blk + aj) * G1::TI - 3 * G1::SB + 32 * blk + 4 * XP;
            fbase = lbase + (BUF * GBUF - 4 * dj * G1::TI + 64 * (j - 2 * HH));
        } else {
            const int lbase = ai * G0::AI + 2 * gg * G0::BI + 128 * blk + (3 - aj) * 24 + 4 * XP;
            fbase = lbase + (BUF * GBUF + CELL0(HH, a) * 32 + (10 + 4 * dj - T0(HH, a)) * 32 - 72);
        }
        constexpr bool check = dj < -1 || dj + 1 > 1;
        const int vs = 4 * blk - aj;
        f2 w[4];
        static_for<0, 8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int bjs = s & 3, bis = s >> 2;
            constexpr int sconst = FLIP ? bis * G1::BI + (3 - bjs) * G1::SB : bis * G0::BI + bjs * 32;
            float v;
            if (VAR & 8) v = 1.0f;
            else v = *reinterpret_cast<const float *>(smem + fbase + sconst);
            if constexpr (check) {
                constexpr int hi = 10 - 4 * dj - bjs, lo = -10 - 4 * dj - bjs;   // lo <= vs <= hi
                if constexpr (hi < 4) v = vs <= hi ? v : 0.0f;
                if constexpr (lo > -3) v = vs >= lo ? v : 0.0f;
            }
            w[s & 3][s >> 2] = v;
        });
        u4 vh, vl;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = f16s::pk_scale(w[q], sc_g2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned hq, lq;
            split2(w[(2 * q) & 3][q >> 1], w[(2 * q + 1) & 3][q >> 1], hq, lq);
            vh[q] = hq; vl[q] = lq;
        }
        gh[fi] = __builtin_bit_cast(h8, vh);
        gl[fi] = __builtin_bit_cast(h8, vl);
    };

    // MFMAs of pair j of (R, HH), channel chunk ch: the X operands are read once and used by both centre blocks that meet j
    auto mma1 = [&](auto role_c, auto hh_c, auto jl_c, auto chc) {
        constexpr int R = decltype(role_c)::value, HH = decltype(hh_c)::value, j = 2 * HH + decltype(jl_c)::value, ch = decltype(chc)::value;
        // the products of this j: indices into the (R, HH) list
        constexpr int n = nfrag(R, HH);
        constexpr int i0 = (n > 0 && frag_j(R, HH, 0) == j) ? 0 : (n > 1 && frag_j(R, HH, 1) == j) ? 1 : (n > 2 && frag_j(R, HH, 2) == j) ? 2 : (n > 3 && frag_j(R, HH, 3) == j) ? 3 : -1;
        if constexpr (i0 >= 0) {
            constexpr bool two = (i0 + 1 < n) && frag_j(R, HH, i0 + 1) == j;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int f_i = ln & 15, f_g = ln >> 4;
            const int xb = xpar * PARS + f_i * CHS + f_g * 16;
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
                        {
                            constexpr int ab = frag_ab(R, HH, i0), fs = frag_slot(R, HH, i0);
                            acc[ab][2 * ch + c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? xl[c2] : xh[c2], pr == 1 ? gl[fs] : gh[fs], acc[ab][2 * ch + c2], 0, 0, 0);
                        }
                        if constexpr (two) {
                            constexpr int ab = frag_ab(R, HH, i0 + 1), fs = frag_slot(R, HH, i0 + 1);
                            acc[ab][2 * ch + c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? xl[c2] : xh[c2], pr == 1 ? gl[fs] : gh[fs], acc[ab][2 * ch + c2], 0, 0, 0);
                        }
                    }
                });
            }
        }
    };

    // One step: the MFMAs of (R, HH) interleaved with the gather of the NEXT step's operands ((R, 1 - HH), from the other G buffer)
    auto step_body = [&](auto flipc, auto role_c, auto hh_c, f16s::scale2_t sc_next) {
        constexpr int R = decltype(role_c)::value, HH = decltype(hh_c)::value;
        constexpr int NN = nfrag(R, 1 - HH);
        typedef std::integral_constant<int, 1 - HH> nh_t;   // the next step's half; its G image is in buffer 1 - HH
        auto gnext = [&](auto ic) {
            if constexpr (decltype(ic)::value < NN) gather1(flipc, role_c, nh_t{}, nh_t{}, ic, sc_next);
        };
        typedef std::integral_constant<int, 0> i0_t; typedef std::integral_constant<int, 1> i1_t;
        typedef std::integral_constant<int, 2> i2_t; typedef std::integral_constant<int, 3> i3_t;
        gnext(i0_t{});
        __builtin_amdgcn_sched_barrier(0);
        mma1(role_c, hh_c, i0_t{}, i0_t{});
        __builtin_amdgcn_sched_barrier(0);
        gnext(i1_t{});
        __builtin_amdgcn_sched_barrier(0);
        mma1(role_c, hh_c, i0_t{}, i1_t{});
        __builtin_amdgcn_sched_barrier(0);
        gnext(i2_t{});
        __builtin_amdgcn_sched_barrier(0);
        mma1(role_c, hh_c, i1_t{}, i0_t{});
        __builtin_amdgcn_sched_barrier(0);
        gnext(i3_t{});
        __builtin_amdgcn_sched_barrier(0);
        mma1(role_c, hh_c, i1_t{}, i1_t{});
        __builtin_amdgcn_sched_barrier(0);
    };
    // the gather alone (operands of step 0 of a workgroup's first task)
    auto gather_only = [&](auto flipc, auto role_c, f16s::scale2_t sc) {
        typedef std::integral_constant<int, 0> z_t;
        constexpr int R = decltype(role_c)::value;
        static_for<0, nfrag(R, 0)>([&](auto ic) { gather1(flipc, role_c, z_t{}, z_t{}, ic, sc); __builtin_amdgcn_sched_barrier(0); });
    };

    // (one copy of everything below per role: the role is fixed for the life of the wave, so the whole program of a matrix wave is
    // specialised once -- a switch per step instead costs phi copies of every accumulator and operand register at each merge)
    auto run_task = [&](auto role_c, const Task &tk, const Task &tn, bool has_next, auto flipc, auto nflipc, bool first) {
        typedef std::integral_constant<int, 0> c0; typedef std::integral_constant<int, 1> c1;
        const int ksum = to_sgpr(scl_k[0]);
        const f16s::scale2_t sc_g2 = f16s::scale2_from_exp(to_sgpr(scl_k[1]));
        f16s::scale2_t sc_gn = sc_g2;        // the G scale of the next task, read after the barrier of step 10
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ab][ct] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

        for (int s = 0; s < NS; s += 2) {
            // even step (hh 0): DMA of step s + 2 into buffer 0 (its operands were gathered during step s - 1)
            if (first && s == 2) stamp(3);                      // DMA issued
            step_body(flipc, role_c, c0{}, sc_g2);
            if (first && s == 2) stamp(4);                      // MFMAs and gather done
            __syncthreads();
            if (first && s == 2) stamp(6);
            // odd step (hh 1): DMA of step s + 3 into buffer 1; the operands gathered now are those of step s + 2 -- of the NEXT task
            // after the last u: its flip decides the layout, its scale was published before the barrier above
            if (first && s == 2) stamp(7);
            if (s + 2 < NS) step_body(flipc, role_c, c1{}, sc_g2);
            else {
                if (has_next) sc_gn = f16s::scale2_from_exp(to_sgpr(scl_k[1]));
                step_body(nflipc, role_c, c1{}, sc_gn);
            }
            if (first && s == 2) stamp(8);
            __syncthreads();
            if (first && (s == 0 || s == 2)) stamp(s == 0 ? 2 : 10);
        }
        if (first) stamp(12);

        // epilogue: D[row = channel 4q + r][col = pixel i] -> Es[c][ai][x], 16-byte slots rotated by 8 ai + 32 ((c>>2)&1)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int f_i = ln & 15, f_g = ln >> 4;
        const int f_ai = f_i >> 2, f_aj = f_i & 3;
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
        scatter(role_c);
        __syncthreads();
        store_rows(tk, ksum);
        if (first) stamp(13);
        __syncthreads();
        __syncthreads();                                       // (Q') X(0) of the next task written
        if (first) stamp(14);
    };

    auto matrix_main = [&](auto role_c) {
        // prologue: G(0) and G(1) of the first task, then its step-0 operands
        int t = (int)xcd_remap(blockIdx.x, gridDim.x);
        if (t < ntasks) {
            const Task tk = get_task(t);
        }
        stamp(1);
        __syncthreads();                                           // (P)
        if (t < ntasks) {
            const Task tk = get_task(t);
            const f16s::scale2_t sc = f16s::scale2_from_exp(to_sgpr(scl_k[1]));
            if (tk.flip) gather_only(std::integral_constant<int, 1>{}, role_c, sc);
            else gather_only(std::integral_constant<int, 0>{}, role_c, sc);
        }
        __syncthreads();                                           // (Q)
        stamp(2);
        for (; t < ntasks; t += gridDim.x) {
            const Task tk = get_task(t);
            const bool has_next = t + (int)gridDim.x < ntasks;
            const Task tn = get_task(has_next ? t + (int)gridDim.x : t);
            const bool first = t < (int)gridDim.x;
            typedef std::integral_constant<int, 0> c0; typedef std::integral_constant<int, 1> c1;
            if (tk.flip) { if (tn.flip) run_task(role_c, tk, tn, has_next, c1{}, c1{}, first); else run_task(role_c, tk, tn, has_next, c1{}, c0{}, first); }
            else { if (tn.flip) run_task(role_c, tk, tn, has_next, c0{}, c1{}, first); else run_task(role_c, tk, tn, has_next, c0{}, c0{}, first); }
        }
    };
    switch (role) {
    case 0: matrix_main(std::integral_constant<int, 0>{}); break;
    case 1: matrix_main(std::integral_constant<int, 1>{}); break;
    case 2: matrix_main(std::integral_constant<int, 2>{}); break;
    default: matrix_main(std::integral_constant<int, 3>{}); break;
    }
    stamp(15);
    dump();
}

} // namespace hq

// variant: profiling switches (fn2_debug.h)
int corr_backward_f16x2_pipe(const float *in1, const float *in2, const float *gout, long gout_bs, float *g1, float *g2,
                             int B, int C, int H, int W, int variant, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(gout, 16) || !aligned(g1, 16) || !aligned(g2, 16) || (gout_bs % 4) != 0) return FN2_EALIGN;
    if (W > 64) return FN2_EUNSUPPORTED;
    hq::Args a;
    a.nbr[0] = in2; a.nbr[1] = in1; a.gout = gout; a.gin[0] = g1; a.gin[1] = g2; a.gbs = gout_bs;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.NRG = (H / 2 + 3) / 4; a.NCGR = C / hq::CG;
    a.nflip = 2; a.flip0 = 0;
    a.fC = (float)C; a.rC = 1.0f / (float)C;
#ifdef FN2_DEBUG_BUILD
    a.dbg = (variant & 64) ? static_cast<unsigned long long *>(corr_f16x2_get_debug_buffer()) : nullptr;
#else
    a.dbg = nullptr;
#endif
    if (variant & 1536) { a.nflip = 1; a.flip0 = (variant & 512) ? 1 : 0; variant &= ~1536; variant |= 0; }   // profiling: one gradient only
    const long ntasks = (long)a.nflip * B * 2 * a.NRG * a.NCGR;
    if (ntasks == 0) return FN2_OK;
    if (ntasks > 0x3fffffffL) return FN2_EINVAL;
    const unsigned grid = ntasks < 256 ? (unsigned)ntasks : 256u;
#define FN2_HQ(V) case V: hipLaunchKernelGGL((hq::corr_bwd_pipe<V>), dim3(grid), dim3(hq::NWAVES * 64), 0, s, a); return launch_status();
    switch (variant) {
        FN2_HQ(0)
#ifdef FN2_DEBUG_BUILD
        FN2_HQ(1) FN2_HQ(2) FN2_HQ(8) FN2_HQ(16) FN2_HQ(64) FN2_HQ(65) FN2_HQ(66) FN2_HQ(72) FN2_HQ(80) FN2_HQ(74) FN2_HQ(75) FN2_HQ(256)
#endif
    default: return FN2_EINVAL;
    }
#undef FN2_HQ
}

} // namespace fn2
#endif   // FN2_DEBUG_BUILD
