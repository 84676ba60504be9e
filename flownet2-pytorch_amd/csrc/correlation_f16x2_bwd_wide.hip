// correlation_f16x2_bwd_wide.hip -- correlation backward (both input gradients) of correlation_f16x2_bwd.hip for maps WIDER
// than 64 pixels (Sintel-size inputs; the reference kernels have no width limit: correlation_cuda_kernel.cu:150-334).
//
// Same contraction, numerics (block-scaled two-term f16 split, f16x2_split.h), LDS images and wave specialisation as the
// narrow kernel; read its header first.  What the width adds: on a parity lattice a centre column block a (4 lattice columns =
// 8 pixels) has neighbours in the column blocks a-3 .. a+3.  A task now owns a CENTRE WINDOW of 8 column blocks (64 pixels,
// window index xw) of its 4 centre rows and walks the neighbour blocks 8xw-3 .. 8xw+12 in two PASSES of 8 blocks (64 pixels:
// exactly the X chunk and the G image of the narrow kernel), each pass over the 6 neighbour row blocks u:
//     pass 0: neighbour blocks 8xw-3 .. 8xw+4        pass 1: 8xw+5 .. 8xw+12 (skipped when it lies right of the image)
// The sums of both passes stay in the same accumulators (no atomics, deterministic).  The centre blocks of a matrix wave are
// paired {0,7} {1,6} {2,5} {3,4}: every wave then has 5 (centre block, neighbour block pair) products in pass 0 and 3 in pass 1
// (the narrow kernel: 6) -- 96 MFMAs per wave, u and 64 channels instead of 72, plus a second round of gathers and X chunks:
// a wide map costs ~1.7x the narrow kernel's time per pixel (the fp32 matrix-core kernel it replaces: 3x).
// FLIP 0 (gradInput1): the gO pixel is the centre -> the G image holds the centre window's columns in both passes;
// FLIP 1 (gradInput2): the gO pixel is the neighbour -> the G image holds the pass's neighbour columns.  Columns outside the
// image are zeros from the buffer range check (X loads and G DMA alike).
#include <type_traits>

#include "corr_params.h"
#include "f16x2_split.h"

namespace fn2 {
namespace hbw {
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

constexpr int DR = 10, D = 21, NU = 6;
constexpr int CG = 64, NCT = CG / 16;             // channels per task, channel tiles of 16
constexpr int CK = 32;                            // channels per X chunk (2 tiles)
constexpr int CHS = 288, PARS = CK * CHS, XTERM = 2 * PARS, XBUF = 2 * XTERM;
// G image: the narrow kernel's layout (strides chosen for conflict-free gathers; see GL there)
template <int FLIP> struct GL {
    static constexpr int BI = 256;
    static constexpr int TI = FLIP ? 1024 + 4 : 1024 + 32;
    static constexpr int AI = FLIP ? D * TI + 76 : D * TI + 128;
    static constexpr int IMG = 3 * AI + D * TI;
};
constexpr int GIMG = GL<0>::IMG;
static_assert(GL<1>::IMG <= GIMG && GIMG % 16 == 0, "G image");
constexpr int X_OFS = GIMG, LDS_BYTES = X_OFS + 2 * XBUF;
constexpr int E_BYTES = CG * 4 * 64 * 4;          // epilogue image [64 channels][4 rows][64 x] floats, aliases the X buffers
static_assert(E_BYTES <= 2 * XBUF && LDS_BYTES <= 163840, "LDS budget");
constexpr int WPX = 64;                           // pixels of a centre window / of a pass's neighbour window

struct Args {
    const float *nbr[2];   // [0] = in2 (neighbours for gradInput1), [1] = in1 (for gradInput2)
    const float *gout;
    float *gin[2];         // [0] = gradInput1, [1] = gradInput2
    int B, C, H, W;        // H even, W % 8 == 0, C % 64 == 0
    int NRG, NCGR, NXW;    // row groups per parity, channel groups, centre windows
    float fC;              // (float)C: a kernel argument so that it is an SGPR
};

// neighbour column blocks of a pass, relative to the centre window's first block: PO(ps) .. PO(ps) + 7
__host__ __device__ constexpr int PO(int ps) { return ps ? 5 : -3; }
// centre column blocks of a wave role and the block pairs (2j, 2j+1) of a pass they meet
__host__ __device__ constexpr int a_blk(int role, int ab) { return ab ? 7 - role : role; }
__host__ __device__ constexpr bool meets(int ps, int a, int j) { return PO(ps) + 2 * j + 1 >= a - 3 && PO(ps) + 2 * j <= a + 3; }
__host__ __device__ constexpr int frag_idx(int ps, int role, int ab, int j)   // index among the role's products of the pass, or -1
{
    int idx = 0;
    for (int b = 0; b < 2; ++b)
        for (int jj = 0; jj < 4; ++jj) {
            if (b == ab && jj == j) return meets(ps, a_blk(role, ab), j) ? idx : -1;
            if (meets(ps, a_blk(role, b), jj)) ++idx;
        }
    return -1;
}
__host__ __device__ constexpr int n_frags(int ps, int role)
{
    int n = 0;
    for (int b = 0; b < 2; ++b)
        for (int jj = 0; jj < 4; ++jj) n += meets(ps, a_blk(role, b), jj) ? 1 : 0;
    return n;
}
constexpr int NF = 5;   // (centre block, block pair) products of a wave in one pass, at most
static_assert(n_frags(0, 0) == 5 && n_frags(0, 1) == 5 && n_frags(0, 2) == 5 && n_frags(0, 3) == 5 && n_frags(1, 0) == 3 &&
              n_frags(1, 1) == 3 && n_frags(1, 2) == 3 && n_frags(1, 3) == 3, "5 + 3 products per role");

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

constexpr int NSW = 4, NWAVES = NSW + 8;   // staging waves, waves per workgroup (3 per SIMD)
constexpr int XK = 32 / (2 * NSW);          // X items per chunk (32 channels) and staging lane
struct XSet { u4 v[XK][2]; };

// The lane id, formed where it is needed (two instructions).  An opaque copy of a `lane` variable kept across the task loop is what
// this kernel used until round 4; once the register budget tipped, that variable was spilled and every gather began with a
// scratch reload and a vmcnt(0) wait (+18 % on the kernel).
__device__ __forceinline__ int lane_now()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__global__ __launch_bounds__(NWAVES * 64, 3) void corr_bwd_f16x2_wide(Args p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    __shared__ int scl_k[2];   // [1] = kg, the G exponent of the task about to start (see the narrow kernel; [0] unused since round 4)
    // one X scale per CHANNEL of the task (correlation_f16x2_bwd.hip): floats 2^kx at position 8 (c & 7) + (c >> 3), two task parities
    __shared__ __attribute__((aligned(16))) float scl_sx[2][CG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < NSW;
    const int w8 = is_stage ? wave : wave - NSW;
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int hw = p.H * p.W;
    const int per_fn = 2 * p.NRG * p.NXW * p.NCGR;          // tasks per (flip, batch item)
    const int ntasks = 2 * p.B * per_fn;
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int lgC = pow2 ? 31 - __builtin_clz((unsigned)p.C) : 0;

    struct Task { int flip, n, py, rg, xw, cg, nv; };
    auto get_task = [&](int t) -> Task {
        Task k;
        k.cg = t % p.NCGR; t /= p.NCGR;
        k.xw = t % p.NXW; t /= p.NXW;
        k.rg = t % p.NRG; t /= p.NRG;
        k.py = t & 1; t >>= 1;
        k.n = t % p.B;
        k.flip = t / p.B;
        k.cg = __builtin_amdgcn_readfirstlane(k.cg); k.rg = __builtin_amdgcn_readfirstlane(k.rg);
        k.xw = __builtin_amdgcn_readfirstlane(k.xw);
        k.py = __builtin_amdgcn_readfirstlane(k.py); k.n = __builtin_amdgcn_readfirstlane(k.n);
        k.flip = __builtin_amdgcn_readfirstlane(k.flip);
        k.nv = (WPX * k.xw + 8 * PO(1) < p.W) ? 2 * NU : NU;   // (pass, u) steps: pass 1 only if its first block is inside the image
        return k;
    };
    // first pixel of the neighbour window of step v (pass v / NU)
    auto nbr_x0 = [&](const Task &tk, int v) { return WPX * tk.xw + 8 * (v >= NU ? PO(1) : PO(0)); };

    // ---- write-out of the epilogue image (all waves): 256 rows (channel, centre row) of 64 floats, 4 rows per instruction
    float *Es = reinterpret_cast<float *>(smem + X_OFS);
    auto store_rows = [&](const Task &tk, int kg, int par) {
        int ln = lane_now();
        const int g = ln >> 4, xg = 4 * (ln & 15), xw = WPX * tk.xw + xg;
        constexpr int NRI = (CG + NWAVES - 1) / NWAVES;
        const int y = 2 * (4 * tk.rg + g) + tk.py;
        const bool lane_ok = 4 * tk.rg + g < HL && xw < p.W;
        const unsigned vo = lane_ok ? (unsigned)((y * p.W + xw) * 4) : 0x80000000u;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.gin[tk.flip] + (long)tk.n * p.C * HW, 0, (unsigned)(p.C * HW * 4), 0x00020000);
        auto chan = [&](int i) { return wave + NWAVES * i; };
        auto read_row = [&](int c) {   // Es[c][ai = g][x], 16-byte slots rotated by 8 ai + 32 ((c >> 2) & 1)
            return *reinterpret_cast<const f4 *>(Es + (c * 4 + g) * 64 + ((xg + 8 * g + 32 * ((c >> 2) & 1)) & 63));
        };
        f4 vals[NRI];
#pragma unroll
        for (int i = 0; i < NRI; ++i) vals[i] = read_row(chan(i) & (CG - 1));
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
            if (lane_ok && (__builtin_amdgcn_classf(vals[i][0], 0x207) | __builtin_amdgcn_classf(vals[i][1], 0x207) |
                            __builtin_amdgcn_classf(vals[i][2], 0x207) | __builtin_amdgcn_classf(vals[i][3], 0x207)))
                bad |= 1u << i;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, scaled(vals[i], -ksum_of(c) - lgC)), rso, (int)vo, (int)((tk.cg * CG + c) * HW * 4), 2);   // sc1: see correlation_f16x2_bwd.hip
        }
        if (bad) {   // an operand beyond the f16 range: those outputs again, as fp32 fma chains
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
                    const float ex = nonfin ? exact_grad(p, tk.flip, tk.n, tk.cg * CG + c, y, xw + e) : __builtin_ldexpf(cur, -ksum);
                    val[0] = e == 0 ? ex : val[0]; val[1] = e == 1 ? ex : val[1];
                    val[2] = e == 2 ? ex : val[2]; val[3] = e == 3 ? ex : val[3];
                }
                *reinterpret_cast<f4 *>(p.gin[tk.flip] + (((long)tk.n * p.C + tk.cg * CG + c) * p.H + y) * p.W + xw) = scaled(val, kx_ex);
            }
        }
    };

    if (is_stage) {
        // ================= staging waves =================
        // this lane's X chunk position: rebuilt from an opaque copy of the lane id wherever it is needed (kept in registers across
        // the task loop it is spilled, and a scratch reload waits for vmcnt(0) -- for the X loads just requested)
        auto x_ofs = [&]() {
            int ln = lane_now();
            const int piece = (ln & 3) + 4 * ((ln >> 4) & 1), row = (ln >> 2) & 3, chn = 2 * w8 + (ln >> 5);
            return chn * CHS + (piece >> 1) * 64 + (piece & 1) * 16 + (row >> 1) * 32 + (row & 1) * 8;
        };
        const unsigned xbytes = (unsigned)(p.C * HW * 4), gbytes = (unsigned)(D * D * HW * 4);

        // G image of step v = (pass, u): staging wave w copies centre row ai = w, one DMA instruction per displacement column
        // (lane = (neighbour row bi, 16-byte piece of the 64-pixel window)).  FLIP 0: the centre window's columns, FLIP 1: the
        // pass's neighbour columns.
        auto g_dma = [&](const Task &tk, int v) {
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gout + (long)tk.n * D * D * HW), 0, gbytes, 0x00020000);
            const int u = v >= NU ? v - NU : v;
            int ln = lane_now();
            const int bi = ln >> 4, pc = ln & 15;
            const int ai = w8;
            const int tj = tk.flip ? 20 - 4 * u - bi + ai : 4 * u + bi - ai;
            const int il = tk.flip ? 4 * tk.rg - DR + 4 * u + bi : 4 * tk.rg + ai;
            const int x = (tk.flip ? nbr_x0(tk, v) : WPX * tk.xw) + 4 * pc;
            const bool ok = tj >= 0 && tj < D && il >= 0 && il < HL && x >= 0 && x < p.W;
            const unsigned vo = ok ? (unsigned)(((tj * D * p.H + 2 * il + tk.py) * p.W + x) * 4) : 0x80000000u;
            const int l0 = tk.flip ? ai * GL<1>::AI : ai * GL<0>::AI, lt = tk.flip ? GL<1>::TI : GL<0>::TI;
#pragma unroll
            for (int ti = 0; ti < D; ++ti)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (FN2_LDS(void) *)(smem + l0 + ti * lt), 16, (int)vo, ti * (int)(HW * 4), 0, 0);
        };
        // X chunk (v, ch): neighbour rows 4rg - 10 + 4u .. +3, the pass's 64 neighbour columns, channels cg*64 + 32*ch .. +31
        auto x_issue = [&](XSet &L, const Task &tk, int v, int ch) {
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.nbr[tk.flip] + (long)tk.n * p.C * HW), 0, xbytes, 0x00020000);
            const int u = v >= NU ? v - NU : v;
            int ln = lane_now();   // the lane geometry is rebuilt here (opaque copy): four registers less across the task loop
            const int piece = (ln & 3) + 4 * ((ln >> 4) & 1), row = (ln >> 2) & 3, chn = 2 * w8 + (ln >> 5);
            const int il = 4 * tk.rg - DR + 4 * u + row;
            const int x = nbr_x0(tk, v) + 8 * piece;
            const bool ok = il >= 0 && il < HL && x >= 0 && x < p.W;
            const unsigned vo = ok ? (unsigned)((chn * hw + (2 * il + tk.py) * p.W + x) * 4) : 0x80000000u;   // < 2^31 (applicable())
#pragma unroll
            for (int k = 0; k < XK; ++k) {
                const int soff = (int)((tk.cg * CG + ch * CK + 2 * NSW * k) * HW * 4);
                L.v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)vo, soff, 0);
                L.v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(vo + 16), soff, 0);
            }
        };
        auto x_write1 = [&](const XSet &L, char *buf, int w_ofs, int k, float sc) {
            const f4 x0 = __builtin_bit_cast(f4, L.v[k][0]) * sc, x1 = __builtin_bit_cast(f4, L.v[k][1]) * sc;   // power of two: exact
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
        auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        // operand sample of a task (f16x2_split.h), inside the centre window: X from the neighbour rows of u = 2, G from the gO
        // image of the same u
        constexpr int U0 = 2;
        struct Samp { u2 x, x2, g; };   // X: lane = channel of the task, 4 values of it; G: as before
        auto sample_issue = [&](const Task &tk, Samp &S) {
            int ln = lane_now();
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.nbr[tk.flip] + (long)tk.n * p.C * HW), 0, xbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gout + (long)tk.n * D * D * HW), 0, gbytes, 0x00020000);
            const int ai = ln & 3, bi = (ln >> 2) & 3, q = ln >> 4;
            const int tj = tk.flip ? 20 - 4 * U0 - bi + ai : 4 * U0 + bi - ai;
            const int ilg = tk.flip ? 4 * tk.rg - DR + 4 * U0 + bi : 4 * tk.rg + ai;
            // sample columns inside the part of the window that lies in the image (the last window of a row may be narrower than 64 px,
            // but never narrower than 8: a sample taken beyond the row reads zeros and would leave the channel unscaled)
            const int xs = 2 * (((5 * ln) >> 1) & 31), xs2 = (xs + 32) & 63;
            const int x = WPX * tk.xw + (WPX * tk.xw + xs < p.W ? xs : xs & 7);
            const int c = tk.cg * CG + ln, ilx = min(4 * tk.rg + (ln & 3), HL - 1);   // X sample rows: two of the centre rows, clamped into the
            // image -- every lane (= channel) gets four real values whatever the map's height (a lane without a sample would leave its
            // channel unscaled: scripts/soak_fuzz.py, H = 2)
            const int ti = (5 * q + (ln & 3) + bi) % D;
            const unsigned ox = (ilx >= 0 && ilx < HL && x < p.W) ? (unsigned)((c * HW + (long)(2 * ilx + tk.py) * p.W + x) * 4) : 0x80000000u;
            const int ilx2 = min(4 * tk.rg + ((ln + 2) & 3), HL - 1), xb = WPX * tk.xw + (WPX * tk.xw + xs2 < p.W ? xs2 : xs2 & 7);
            const unsigned ox2 = (ilx2 >= 0 && ilx2 < HL && xb < p.W) ? (unsigned)((c * HW + (long)(2 * ilx2 + tk.py) * p.W + xb) * 4) : 0x80000000u;
            const unsigned og = (ilg >= 0 && ilg < HL && x < p.W) ? (unsigned)((((tj * D + ti) * p.H + 2 * ilg + tk.py) * p.W + x) * 4) : 0x80000000u;
            S.x = __builtin_amdgcn_raw_buffer_load_b64(rsx, (int)ox, 0, 0);
            S.x2 = __builtin_amdgcn_raw_buffer_load_b64(rsx, (int)ox2, 0, 0);
            S.g = __builtin_amdgcn_raw_buffer_load_b64(rsg, (int)og, 0, 0);
        };
        auto sample_scales = [&](const Samp &S, int &kx, int &kg) {   // kx: this lane's channel
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
                int ln = lane_now();
                scl_sx[par][8 * (ln & 7) + (ln >> 3)] = f16s::scale_from_exp(kx);
                if (ln == 0) scl_k[1] = kg;
            }
        };
        XSet XA0, XA1, XB0, XB1;
        int t = (int)xcd_remap(blockIdx.x, gridDim.x);
        Samp SM;
        int kx_n = 0, kg_n = 0;
        int it = 0;
        if (t < ntasks) {
            const Task tk = get_task(t);
            sample_issue(tk, SM);
            x_issue(XA0, tk, 0, 0);
            x_issue(XA1, tk, 0, 1);
            g_dma(tk, 0);
            dma_wait();
            sample_scales(SM, kx_n, kg_n);
            publish(0, kx_n, kg_n);
        }
        __syncthreads();                                       // (A) G(0) complete, the first task's exponents published
        for (; t < ntasks; t += gridDim.x) {
            const Task tk = get_task(t);
            const bool has_next = t + (int)gridDim.x < ntasks;
            const Task tn = get_task(has_next ? t + (int)gridDim.x : t);
            const int kg_cur = kg_n, par = it & 1;
            auto one_v = [&](int v, XSet &C0, XSet &C1, XSet &N0, XSet &N1) {
                // phase 1 (the matrix waves gather the G operands of v): request the next X chunks, write both X chunks of v
                int l5 = lane_now();
                const float *sxp = &scl_sx[par][8 * (2 * w8 + (l5 >> 5))];   // the scales of the lane's eight channels, read one item ahead
                float sc = sxp[0];
                if (v + 1 < tk.nv) { x_issue(N0, tk, v + 1, 0); x_issue(N1, tk, v + 1, 1); }
                else if (has_next) { x_issue(N0, tn, 0, 0); x_issue(N1, tn, 0, 1); }
                const int w_ofs = x_ofs();
#pragma unroll
                for (int k = 0; k < 2 * XK; ++k) {
                    const float nx = sxp[k + 1 < 2 * XK ? k + 1 : k];
                    if (k < XK) x_write1(C0, smem + X_OFS, w_ofs, k, sc);
                    else x_write1(C1, smem + X_OFS + XBUF, w_ofs, k - XK, sc);
                    sc = nx;
                }
                __syncthreads();                               // (B) the G image is free, the X chunks complete
                // phase 2 (all MFMAs of v): the next G image by DMA (and the next task's operand sample, ahead of it)
                if (v + 1 < tk.nv) g_dma(tk, v + 1);
                else if (has_next) { sample_issue(tn, SM); g_dma(tn, 0); }
                dma_wait();
                __syncthreads();                               // (A') the X buffers are free, the next G image complete
            };
            for (int v = 0; v < tk.nv; v += 2) {
                one_v(v, XA0, XA1, XB0, XB1);
                one_v(v + 1, XB0, XB1, XA0, XA1);
            }
            if (has_next) { sample_scales(SM, kx_n, kg_n); publish(par ^ 1, kx_n, kg_n); }
            __syncthreads();                                   // epilogue image (over the X buffers) complete
            store_rows(tk, kg_cur, par);
            __syncthreads();                                   // image read: the X buffers are free for the next task
            ++it;
        }
        return;
    }

    // ================= matrix-core waves =================
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);

    int itm = 0;   // tasks done by this workgroup: parity selects the half of scl_sx
    auto run_task = [&](const Task &tk, auto flipc) {
        constexpr int FLIP = decltype(flipc)::value;
        const int kg_cur = to_sgpr(scl_k[1]);
        const f16s::scale2_t sc_g2 = f16s::scale2_from_exp(kg_cur);
        int ln = lane_now();
        const int f_i = ln & 15, f_g = ln >> 4;
        const int f_ai = f_i >> 2, f_aj = f_i & 3;
        const int xb = xpar * PARS + f_i * CHS + f_g * 16;

        f4 acc[2][NCT];
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ab][ct] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        h8 gh[NF], gl[NF];

        // Gather of the G operands of a pass (the narrow kernel's gather with dm = PO(pass) + 2j + blk - a)
        auto gather = [&](auto ps_c, auto role_c, auto xp_c) {
            constexpr int PS = decltype(ps_c)::value;
            constexpr int R = decltype(role_c)::value;
            constexpr int XP = decltype(xp_c)::value;
            typedef GL<FLIP> L;
            constexpr int SB = L::TI - 8;
            int l2 = lane_now();
            const int ai = (l2 & 15) >> 2, aj = l2 & 3, blk = (l2 >> 4) & 1, gg = l2 >> 5;
            const int lbase = FLIP ? ai * L::AI + 2 * gg * L::BI + (DR - 4 * blk + aj) * L::TI - 3 * SB + 32 * blk + 4 * XP
                                   : ai * L::AI + 2 * gg * L::BI + (DR + 4 * blk - aj) * L::TI + 8 * aj;
            const int vs = 4 * blk - aj;
            static_for<0, 2>([&](auto abc) {
                constexpr int ab = decltype(abc)::value;
                constexpr int a = a_blk(R, ab);
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int fi = frag_idx(PS, R, ab, j);
                    if constexpr (fi >= 0) {
                        constexpr int dj = PO(PS) + 2 * j - a;                    // dm = dj + blk
                        constexpr int pconst = FLIP ? -4 * dj * L::TI + 64 * j : 4 * dj * L::TI + 32 * a;
                        constexpr bool check = dj < -1 || dj > 0;                 // some slot may fall outside the 21-wide band
                        const int fbase = lbase + pconst;
                        f2 w[4];
                        static_for<0, 8>([&](auto sc) {
                            constexpr int s = decltype(sc)::value;
                            constexpr int bjs = s & 3, bis = s >> 2;
                            constexpr int sconst = FLIP ? bis * L::BI + (3 - bjs) * SB : bis * L::BI + bjs * L::TI;
                            const int ofs = fbase + sconst;
                            float v;
                            if constexpr (FLIP) v = *reinterpret_cast<const float *>(smem + ofs);
                            else v = *reinterpret_cast<const float *>(smem + ofs + 4 * XP);
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
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
        };
        // All MFMAs of a step: D[channel][pixel] += X[channel][q] * G[q][pixel]
        auto mma = [&](auto ps_c, auto role_c) {
            constexpr int PS = decltype(ps_c)::value;
            constexpr int R = decltype(role_c)::value;
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int f0 = frag_idx(PS, R, 0, j), f1 = frag_idx(PS, R, 1, j);
                if constexpr (f0 >= 0 || f1 >= 0) {
                    static_for<0, 2>([&](auto chc) {
                        constexpr int ch = decltype(chc)::value;
                        const char *buf = smem + X_OFS + ch * XBUF;
                        h8 xh[2], xl[2];
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            xh[c2] = *reinterpret_cast<const h8 *>(buf + xb + c2 * 16 * CHS + j * 64);
                            xl[c2] = *reinterpret_cast<const h8 *>(buf + xb + c2 * 16 * CHS + j * 64 + XTERM);
                        }
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
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            });
        };
        auto gather_d = [&](auto ps_c) {
            auto by_role = [&](auto xp_c) {
                switch (role) {
                case 0: gather(ps_c, std::integral_constant<int, 0>{}, xp_c); break;
                case 1: gather(ps_c, std::integral_constant<int, 1>{}, xp_c); break;
                case 2: gather(ps_c, std::integral_constant<int, 2>{}, xp_c); break;
                default: gather(ps_c, std::integral_constant<int, 3>{}, xp_c); break;
                }
            };
            if (xpar) by_role(std::integral_constant<int, 1>{});
            else by_role(std::integral_constant<int, 0>{});
        };
        auto mma_d = [&](auto ps_c) {
            switch (role) {
            case 0: mma(ps_c, std::integral_constant<int, 0>{}); break;
            case 1: mma(ps_c, std::integral_constant<int, 1>{}); break;
            case 2: mma(ps_c, std::integral_constant<int, 2>{}); break;
            default: mma(ps_c, std::integral_constant<int, 3>{}); break;
            }
        };
        auto one_pass = [&](auto ps_c) {
            for (int u = 0; u < NU; ++u) {
                gather_d(ps_c);                                    // phase 1
                __syncthreads();                                   // (B) both X chunks of the step complete, the G image is free
                mma_d(ps_c);                                       // phase 2
                __syncthreads();                                   // (A') the X buffers are free; the next G image complete
            }
        };
        one_pass(std::integral_constant<int, 0>{});
        if (tk.nv > NU) one_pass(std::integral_constant<int, 1>{});

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
        store_rows(tk, kg_cur, itm & 1);
        __syncthreads();
    };
    __syncthreads();                                           // (A) G(0) of the first task complete, its scale exponents published
    for (int t = (int)xcd_remap(blockIdx.x, gridDim.x); t < ntasks; t += gridDim.x, ++itm) {
        const Task tk = get_task(t);
        if (tk.flip) run_task(tk, std::integral_constant<int, 1>{});
        else run_task(tk, std::integral_constant<int, 0>{});
    }
}

} // namespace hbw

// maps wider than 64 pixels (called by corr_backward_f16x2; same preconditions otherwise)
int corr_backward_f16x2_wide(const float *in1, const float *in2, const float *gout, float *g1, float *g2, int B, int C, int H, int W,
                             hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(gout, 16) || !aligned(g1, 16) || !aligned(g2, 16)) return FN2_EALIGN;
    hbw::Args a;
    a.nbr[0] = in2; a.nbr[1] = in1; a.gout = gout; a.gin[0] = g1; a.gin[1] = g2;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.NRG = (H / 2 + 3) / 4; a.NCGR = C / hbw::CG; a.NXW = (W + hbw::WPX - 1) / hbw::WPX;
    a.fC = (float)C;
    const long ntasks = 2L * B * 2 * a.NRG * a.NXW * a.NCGR;
    if (ntasks == 0) return FN2_OK;
    if (ntasks > 0x3fffffffL) return FN2_EINVAL;
    const unsigned grid = ntasks < 256 ? (unsigned)ntasks : 256u;   // persistent: one workgroup per CU
    hipLaunchKernelGGL(hbw::corr_bwd_f16x2_wide, dim3(grid), dim3(hbw::NWAVES * 64), 0, s, a);
    return launch_status();
}

} // namespace fn2
