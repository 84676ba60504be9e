// correlation_f16_bwd.hip -- correlation backward (both input gradients) for HALF-precision tensors on the gfx950 f16 matrix
// cores.
//
// The reference dispatches its backward kernels for at::Half too (correlation_cuda_kernel.cu:150-334, :460-554; it sums in
// half there, :229 -- this kernel sums in fp32 like the general kernel and the oracle, the more accurate superset).  Half
// tensors ARE f16 matrix operands: the banded contraction of correlation_f16x2_bwd.hip (read its header first: task = (gradient,
// batch item, y parity, 4 centre rows, 64 channels), the workgroup walks the 6 neighbour row blocks u, 4 staging + 8 matrix
// waves) with everything the fp32 operands needed removed:
//   - no split, no block scale: ONE v_mfma_f32_16x16x32_f16 per (centre block, block pair, channel tile) instead of three
//     (24 per matrix wave and u instead of 72); products of two halfs are exact in fp32;
//   - X operand (in2 / in1): one 16-byte load = 8 pixels = the two parity chunks after four v_perm_b32 (as correlation_f16_fwd.hip),
//     the fp32 kernel's LDS image without its second term;
//   - G operand (gradOutput): the image [ai][ti][bi][x] stays fp32 -- the gather layout of the fp32 kernel (conflict-free strides,
//     one address register + immediates) is kept as it is -- so it cannot be copied by LDS-DMA: the staging waves load the half
//     rows (requested a whole step ahead, 12 loads of 16 B per lane and u), convert (v_cvt_f32_f16) and write them while the
//     matrix waves run the MFMAs; the matrix waves gather fp32 values and pack pairs with v_cvt_pk_f16_f32 (exact: the values
//     were halfs);
//   - the epilogue scales by 1/C, rounds to half (round to nearest even, as T(sum / nelems)) and stores 8 bytes per lane.
// Non-finite inputs: a matrix product multiplies an inf / nan neighbour with the ZERO G entries of the displacements outside the
// window too, which the reference never forms; every output that comes out non-finite is therefore recomputed as a plain fp32
// fma chain over its own window (exact_grad), as in the fp32 kernel.
// FlowNetC's configuration (kernel_size 1, stride1 1, stride2 2, pad == max_displacement == 20), maps up to 64 pixels wide, C % 64 == 0,
// H even, W % 8 == 0, 16-byte aligned tensors; other half shapes take the general kernel (correlation_direct.hip).
#include <type_traits>

#include "corr_params.h"

#ifndef FN2_HBH_ABL   // timing ablations (scripts/half_bwd_abl.sh): 1 no MFMA, 2 no gather, 4 no G conversion / LDS writes,
#define FN2_HBH_ABL 0 // 8 no X writes, 16 no global loads, 32 no epilogue; results are wrong unless 0
#endif

namespace fn2 {
namespace hbh {
constexpr int ABL = FN2_HBH_ABL;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define FN2_LDS(T) __attribute__((address_space(3))) T

constexpr int DR = 10, D = 21, NU = 6;
constexpr int CG = 64, NCT = CG / 16;             // channels per task, channel tiles of 16
constexpr int CK = 32;                            // channels per X chunk (2 tiles)
constexpr int CHS = 288, PARS = CK * CHS, XBUF = 2 * PARS;   // one chunk: [parity][channel][...], 18432 B
// G image (fp32): the layout of correlation_f16x2_bwd.hip (GL there: strides chosen for conflict-free gathers)
template <int FLIP> struct GL {
    static constexpr int BI = 256;
    static constexpr int TI = FLIP ? 1024 + 4 : 1024 + 32;
    static constexpr int AI = FLIP ? D * TI + 76 : D * TI + 128;
    static constexpr int IMG = 3 * AI + D * TI;
};
constexpr int GIMG = GL<0>::IMG;
static_assert(GL<1>::IMG <= GIMG && GIMG % 16 == 0, "G image");
constexpr int E_BYTES = CG * 4 * 64 * 4;          // epilogue image [64 channels][4 rows][64 x] floats, over the X buffers
constexpr int X_OFS = GIMG, LDS_BYTES = X_OFS + E_BYTES;
static_assert(2 * XBUF <= E_BYTES && LDS_BYTES <= 163840, "LDS budget");

struct Args {
    const _Float16 *nbr[2];   // [0] = in2 (neighbours for gradInput1), [1] = in1 (for gradInput2)
    const _Float16 *gout;
    _Float16 *gin[2];         // [0] = gradInput1, [1] = gradInput2
    int B, C, H, W;           // H even, W % 8 == 0, W <= 64, C % 64 == 0
    int NRG, NCGR;            // row groups per parity, channel groups
    float fC, rC;             // (float)C and 1 / C: kernel arguments so that they are SGPRs
};

// centre column blocks of a wave role and the block pairs (2j, 2j+1) they meet (as the fp32 kernel)
__host__ __device__ constexpr int a_blk(int role, int ab) { return role == 0 ? (ab ? 3 : 0) : role == 1 ? (ab ? 2 : 1) : role == 2 ? (ab ? 7 : 4) : (ab ? 6 : 5); }
__host__ __device__ constexpr bool meets(int a, int j) { return 2 * j + 1 >= a - 3 && 2 * j <= a + 3; }
__host__ __device__ constexpr int frag_idx(int role, int ab, int j)
{
    int idx = 0;
    for (int b = 0; b < 2; ++b)
        for (int jj = 0; jj < 4; ++jj) {
            if (b == ab && jj == j) return meets(a_blk(role, ab), j) ? idx : -1;
            if (meets(a_blk(role, b), jj)) ++idx;
        }
    return -1;
}
constexpr int NF = 6;
static_assert(frag_idx(0, 1, 3) == 5 && frag_idx(1, 1, 2) == 5 && frag_idx(2, 1, 3) == 5 && frag_idx(3, 1, 3) == 5, "6 products per role");

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

__device__ __forceinline__ unsigned pk_f16(float a, float b)   // v_cvt_pk_f16_f32, round to nearest even
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){a, b}, h2));
}

// one gradient element as a plain fp32 fma chain over its own displacement window (cold path: non-finite matrix results)
__device__ __forceinline__ float exact_grad(const Args &p, int flip, int n, int c, int y, int x)
{
    const long HW = (long)p.H * p.W;
    const _Float16 *X = p.nbr[flip] + ((long)n * p.C + c) * HW;
    const _Float16 *g = p.gout + (long)n * D * D * HW;
    float s = 0.0f;
    for (int tj = 0; tj < D; ++tj)
        for (int ti = 0; ti < D; ++ti) {
            const int sgn = flip ? -1 : 1;
            const int yq = y + sgn * 2 * (tj - DR), xq = x + sgn * 2 * (ti - DR);   // the neighbour pixel
            if (yq < 0 || yq >= p.H || xq < 0 || xq >= p.W) continue;
            const long gp = flip ? (long)yq * p.W + xq : (long)y * p.W + x;          // the gO pixel
            s = fmaf((float)g[(long)(tj * D + ti) * HW + gp], (float)X[(long)yq * p.W + xq], s);
        }
    return s;
}

constexpr int NSW = 4, NWAVES = NSW + 8;   // staging waves, waves per workgroup (3 per SIMD)
constexpr int XK = 32 / (2 * NSW);          // X items (8 pixels of one channel row) per chunk and staging lane
constexpr int NGL = 12;                     // G loads per lane and u: 4 neighbour rows x 3 blocks of 8 displacement columns
struct XSet { u4 v[XK]; };
struct GSet { u4 v[NGL]; };

__global__ __launch_bounds__(NWAVES * 64, 3) void corr_bwd_f16(Args p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < NSW;
    const int w8 = is_stage ? wave : wave - NSW;   // staging wave 0 .. NSW-1 / matrix wave 0 .. 7
    const int HL = p.H >> 1;
    const int hw = p.H * p.W;
    const long HW = (long)p.H * p.W;
    const int per_fn = 2 * p.NRG * p.NCGR;                  // tasks per (flip, batch item)
    const int ntasks = 2 * p.B * per_fn;
    const bool pow2 = (p.C & (p.C - 1)) == 0;

    struct Task { int flip, n, py, rg, cg; };
    auto get_task = [&](int t) -> Task {
        Task k;
        k.cg = t % p.NCGR; t /= p.NCGR;
        k.rg = t % p.NRG; t /= p.NRG;
        k.py = t & 1; t >>= 1;
        k.n = t % p.B;
        k.flip = t / p.B;
        k.cg = __builtin_amdgcn_readfirstlane(k.cg); k.rg = __builtin_amdgcn_readfirstlane(k.rg);
        k.py = __builtin_amdgcn_readfirstlane(k.py); k.n = __builtin_amdgcn_readfirstlane(k.n);
        k.flip = __builtin_amdgcn_readfirstlane(k.flip);
        return k;
    };

    // ---- write-out of the epilogue image (all waves): 256 rows (channel, centre row) of 64 floats -> half rows of 128 B
    float *Es = reinterpret_cast<float *>(smem + X_OFS);
    auto store_rows = [&](const Task &tk) {
        if (ABL & 32) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 4, xg = 4 * (ln & 15);
        constexpr int NRI = (CG + NWAVES - 1) / NWAVES;      // channels per wave (the last one partial)
        const int y = 2 * (4 * tk.rg + g) + tk.py;
        const bool lane_ok = 4 * tk.rg + g < HL && xg < p.W;
        const unsigned vo = lane_ok ? (unsigned)((y * p.W + xg) * 2) : 0x80000000u;   // out-of-range lanes store nothing
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.gin[tk.flip] + (long)tk.n * p.C * HW, 0, (unsigned)(p.C * hw * 2), 0x00020000);
        auto chan = [&](int i) { return wave + NWAVES * i; };
        auto read_row = [&](int c) {   // Es[c][ai = g][x], 16-byte slots rotated by 8 ai + 32 ((c >> 2) & 1)
            return *reinterpret_cast<const f4 *>(Es + (c * 4 + g) * 64 + ((xg + 8 * g + 32 * ((c >> 2) & 1)) & 63));
        };
        f4 vals[NRI];
#pragma unroll
        for (int i = 0; i < NRI; ++i) vals[i] = read_row(chan(i) & (CG - 1));
        float r, f = 1.0f;   // copied once, after the LDS reads and before the first store (see correlation_f16x2.hip)
        asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(p.rC));
        if (!pow2) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "s"(p.fC));
        auto finish = [&](f4 val) -> u2 {
            if (pow2) { val[0] *= r; val[1] *= r; val[2] *= r; val[3] *= r; }
            else { val[0] /= f; val[1] /= f; val[2] /= f; val[3] /= f; }
            return (u2){pk_f16(val[0], val[1]), pk_f16(val[2], val[3])};
        };
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < NRI; ++i) {
            const int c = chan(i);
            if (c >= CG) continue;                                    // uniform
            if (lane_ok && (__builtin_amdgcn_classf(vals[i][0], 0x207) | __builtin_amdgcn_classf(vals[i][1], 0x207) |
                            __builtin_amdgcn_classf(vals[i][2], 0x207) | __builtin_amdgcn_classf(vals[i][3], 0x207)))
                bad |= 1u << i;
            __builtin_amdgcn_raw_buffer_store_b64(finish(vals[i]), rso, (int)vo, (int)((tk.cg * CG + c) * hw * 2), 2);   // sc1: see correlation_f16x2_bwd.hip
        }
        if (bad) {   // non-finite sums: those outputs again, each over its own displacement window
#pragma unroll 1
            for (int i = 0; i < NRI; ++i) {
                if (!(bad >> i & 1)) continue;
                const int c = chan(i);
                f4 val = read_row(c);
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    const float cur = e == 0 ? val[0] : e == 1 ? val[1] : e == 2 ? val[2] : val[3];
                    const bool nonfin = (__builtin_bit_cast(unsigned, cur) & 0x7f800000u) == 0x7f800000u;
                    const float ex = nonfin ? exact_grad(p, tk.flip, tk.n, tk.cg * CG + c, y, xg + e) : cur;
                    val[0] = e == 0 ? ex : val[0]; val[1] = e == 1 ? ex : val[1];
                    val[2] = e == 2 ? ex : val[2]; val[3] = e == 3 ? ex : val[3];
                }
                *reinterpret_cast<u2 *>(p.gin[tk.flip] + (((long)tk.n * p.C + tk.cg * CG + c) * p.H + y) * p.W + xg) = finish(val);
            }
        }
    };

    if (is_stage) {
        // ================= staging waves =================
        const unsigned xbytes = (unsigned)(p.C * hw * 2), gbytes = (unsigned)(D * D * hw * 2);
        // X items (as the fp32 kernel): item k of a lane = channel 2 NSW k + 2 w + (lane >> 5), row (lane >> 2) & 3, 8-pixel piece
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_row = (lane >> 2) & 3;
        const int s_ch = 2 * w8 + (lane >> 5);
        const int w_ofs = s_ch * CHS + (s_piece >> 1) * 64 + (s_piece & 1) * 16 + (s_row >> 1) * 32 + (s_row & 1) * 8;
        // (`valid` false: nothing to load -- every lane gets an out-of-range offset, the loads return zeros without touching
        // memory.  One straight-line call per step instead of a call in each arm of an if / else: with two arms the register
        // allocator reused a destination of one arm's loads as an address temporary of the other and had to wait for vmcnt(0) --
        // for everything just requested -- at the join.)
        auto x_issue = [&](XSet &L, const Task &tk, int u, int ch, bool valid) {
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(p.nbr[tk.flip] + (long)tk.n * p.C * HW), 0, xbytes, 0x00020000);
            const int il = 4 * tk.rg - DR + 4 * u + s_row;
            const bool ok = valid && il >= 0 && il < HL && 8 * s_piece < p.W;
            const unsigned vo = ok ? (unsigned)((s_ch * hw + (2 * il + tk.py) * p.W + 8 * s_piece) * 2) : 0x80000000u;
#pragma unroll
            for (int k = 0; k < XK; ++k)
                L.v[k] = (ABL & 16) ? (u4)(0x3c003c00u + lane) : __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)vo, (int)((tk.cg * CG + ch * CK + 2 * NSW * k) * hw * 2), 0);
        };
        // 8 consecutive pixels (four dwords) -> the chunks (p0,p2,p4,p6) and (p1,p3,p5,p7) of the two parities
        auto x_write = [&](const XSet &L, char *buf) {
            if (ABL & 8) { asm volatile("" ::"v"(L.v[0]), "v"(L.v[1]), "v"(L.v[2]), "v"(L.v[3])); return; }
#pragma unroll
            for (int k = 0; k < XK; ++k) {
                const u4 q = L.v[k];
                char *dst = buf + w_ofs + k * 2 * NSW * CHS;
                *(FN2_LDS(u2) *)(dst) = (u2){__builtin_amdgcn_perm(q[1], q[0], 0x05040100u), __builtin_amdgcn_perm(q[3], q[2], 0x05040100u)};
                *(FN2_LDS(u2) *)(dst + PARS) = (u2){__builtin_amdgcn_perm(q[1], q[0], 0x07060302u), __builtin_amdgcn_perm(q[3], q[2], 0x07060302u)};
            }
        };
        // G image of u: staging wave w holds centre row ai = w.  Load i = 3 bi + tb of a lane: neighbour row bi (a scalar of the
        // instruction, like the displacement row tj and the gO row it implies), displacement column ti = 8 tb + (lane >> 3), 8 pixels
        // (lane & 7).  FLIP 0: tj = 4u + bi - ai, gO row = centre row ai; FLIP 1: tj = 20 - 4u - bi + ai, gO row = neighbour row bi.
        // Rows that do not exist get an out-of-range offset: the load returns zeros.  The lane mapping is chosen for the LDS writes of
        // the converted rows: with (ti, 8-pixel piece) across the lanes the dword stores of gradInput2's image (column stride = 1 bank)
        // hit 64 distinct banks; the first version (two ti, four bi, eight pieces per instruction) was 4- to 8-way conflicted there and
        // the conversion took 33 of the kernel's 77 us (ablations: scripts/half_bwd_abl.sh).
        auto g_issue = [&](GSet &S, const Task &tk, int u, bool valid) {
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(p.gout + (long)tk.n * D * D * HW), 0, gbytes, 0x00020000);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int ts = ln >> 3, pc = ln & 7, ai = w8;
            const unsigned vlo = (valid && 8 * pc < p.W) ? (unsigned)((ts * hw + 8 * pc) * 2) : 0x80000000u;
            const unsigned vhi = ts + 16 < D ? vlo : 0x80000000u;                              // tb = 2: ti = 16 + ts < 21
#pragma unroll
            for (int i = 0; i < NGL; ++i) {
                const int bi = i / 3, tb = i % 3;
                const int tj = tk.flip ? 20 - 4 * u - bi + ai : 4 * u + bi - ai;
                const int il = tk.flip ? 4 * tk.rg - DR + 4 * u + bi : 4 * tk.rg + ai;
                const bool ok = tj >= 0 && tj < D && il >= 0 && il < HL;                       // uniform
                const int so = ok ? (((tj * D + 8 * tb) * p.H + 2 * il + tk.py) * p.W) * 2 : 0;
                const unsigned v = ok ? (tb == 2 ? vhi : vlo) : 0x80000000u;
                S.v[i] = (ABL & 16) ? (u4)(0x3c003c00u + lane) : __builtin_amdgcn_raw_buffer_load_b128(rsg, (int)v, so, 0);
            }
        };
        auto g_write = [&](const GSet &S, const Task &tk) {
            if (ABL & 4) {
#pragma unroll
                for (int i = 0; i < NGL; ++i) asm volatile("" ::"v"(S.v[i]));
                return;
            }
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int ts = ln >> 3, pc = ln & 7;
            if (tk.flip) {   // gradInput2's column stride is 4 bytes off a multiple of 16 (gather banks): dword stores
                char *dst = smem + w8 * GL<1>::AI + ts * GL<1>::TI + pc * 32;
#pragma unroll
                for (int i = 0; i < NGL; ++i) {
                    const int bi = i / 3, tb = i % 3;
                    if (tb == 2 && ts + 16 >= D) continue;
                    const h8 h = __builtin_bit_cast(h8, S.v[i]);
                    FN2_LDS(float) *d = (FN2_LDS(float) *)(dst + 8 * tb * GL<1>::TI + bi * 256);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] = (float)h[e];
                }
            } else {
                char *dst = smem + w8 * GL<0>::AI + ts * GL<0>::TI + pc * 32;
#pragma unroll
                for (int i = 0; i < NGL; ++i) {
                    const int bi = i / 3, tb = i % 3;
                    if (tb == 2 && ts + 16 >= D) continue;
                    const h8 h = __builtin_bit_cast(h8, S.v[i]);
                    *(FN2_LDS(f4) *)(dst + 8 * tb * GL<0>::TI + bi * 256) = (f4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    *(FN2_LDS(f4) *)(dst + 8 * tb * GL<0>::TI + bi * 256 + 16) = (f4){(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
                }
            }
        };
        auto pick = [](bool c, const Task &a, const Task &b) {   // scalar selects, field by field
            Task r;
            r.flip = c ? a.flip : b.flip; r.n = c ? a.n : b.n; r.py = c ? a.py : b.py; r.rg = c ? a.rg : b.rg; r.cg = c ? a.cg : b.cg;
            return r;
        };
        XSet XA0, XA1, XB0, XB1;
        GSet GS;
        int t = (int)xcd_remap(blockIdx.x, gridDim.x);
        if (t < ntasks) {
            const Task tk = get_task(t);
            g_issue(GS, tk, 0, true);
            x_issue(XA0, tk, 0, 0, true);
            x_issue(XA1, tk, 0, 1, true);
            g_write(GS, tk);
            g_issue(GS, tk, 1, true);
        }
        __syncthreads();                                       // (A) G(0) of the first task complete
        for (; t < ntasks; t += gridDim.x) {
            const Task tk = get_task(t);
            const bool has_next = t + (int)gridDim.x < ntasks;
            const Task tn = get_task(has_next ? t + (int)gridDim.x : t);
            auto one_u = [&](int u, XSet &C0, XSet &C1, XSet &N0, XSet &N1) {
                // phase 1 (the matrix waves gather the G operands of u): request the next X chunks, write both X chunks of u
                const bool more = u + 1 < NU, more2 = u + 2 < NU;
                const Task t1 = pick(more, tk, tn), t2 = pick(more2, tk, tn);      // the tasks of the next step and the one after
                x_issue(N0, t1, more ? u + 1 : 0, 0, more || has_next);
                x_issue(N1, t1, more ? u + 1 : 0, 1, more || has_next);
                x_write(C0, smem + X_OFS);
                x_write(C1, smem + X_OFS + XBUF);
                __syncthreads();                               // (B) the G image is free, the X chunks complete
                // phase 2 (all MFMAs of u): convert and write the NEXT step's G image (its rows were requested a whole step
                // ago), then request the rows of the step after it into the same registers
                if (more || has_next) g_write(GS, t1);
                g_issue(GS, t2, more2 ? u + 2 : u + 2 - NU, more2 || has_next);
                __syncthreads();                               // (A') the X buffers are free, the next G image complete
            };
            for (int u = 0; u < NU; u += 2) {
                one_u(u, XA0, XA1, XB0, XB1);
                one_u(u + 1, XB0, XB1, XA0, XA1);
            }
            __syncthreads();                                   // epilogue image (over the X buffers) complete
            store_rows(tk);
            __syncthreads();                                   // image read: the X buffers are free for the next task
        }
        return;
    }

    // ================= matrix-core waves =================
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);

    auto run_task = [&](const Task &tk, auto flipc) {
        constexpr int FLIP = decltype(flipc)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int f_i = ln & 15, f_g = ln >> 4;                 // pixel / channel index, k group
        const int f_ai = f_i >> 2, f_aj = f_i & 3;
        const int xb = xpar * PARS + f_i * CHS + f_g * 16;      // X operand: lane = (channel, k group) -> the 16-byte unit 4j + g

        f4 acc[2][NCT];
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ab][ct] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        h8 gh[NF];

        // gather of the G operands: the fp32 kernel's gather (see there), the 8 values packed into one f16 fragment
        auto gather = [&](auto role_c, auto xp_c) {
            constexpr int R = decltype(role_c)::value;
            constexpr int XP = decltype(xp_c)::value;
            typedef GL<FLIP> L;
            constexpr int SB = L::TI - 8;
            int l2 = lane;
            asm volatile("" : "+v"(l2));
            const int ai = (l2 & 15) >> 2, aj = l2 & 3, blk = (l2 >> 4) & 1, gg = l2 >> 5;
            const int lbase = FLIP ? ai * L::AI + 2 * gg * L::BI + (DR - 4 * blk + aj) * L::TI - 3 * SB + 32 * blk + 4 * XP
                                   : ai * L::AI + 2 * gg * L::BI + (DR + 4 * blk - aj) * L::TI + 8 * aj;
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
                        constexpr bool check = dj < -1 || dj > 0;                 // some slot may fall outside the 21-wide band
                        const int fbase = lbase + pconst;
                        float w[8];
                        static_for<0, 8>([&](auto sc) {
                            constexpr int s = decltype(sc)::value;
                            constexpr int bjs = s & 3, bis = s >> 2;
                            constexpr int sconst = FLIP ? bis * L::BI + (3 - bjs) * SB : bis * L::BI + bjs * L::TI;
                            const int ofs = fbase + sconst;
                            float v;
                            if (ABL & 2) v = 1.0f;
                            else if constexpr (FLIP) v = *reinterpret_cast<const float *>(smem + ofs);
                            else v = *reinterpret_cast<const float *>(smem + ofs + 4 * XP);
                            if constexpr (check) {
                                constexpr int hi = 10 - 4 * dj - bjs, lo = -10 - 4 * dj - bjs;   // lo <= vs <= hi
                                if constexpr (hi < 4) v = vs <= hi ? v : 0.0f;
                                if constexpr (lo > -3) v = vs >= lo ? v : 0.0f;
                            }
                            w[s] = v;
                        });
                        const u4 vh = {pk_f16(w[0], w[1]), pk_f16(w[2], w[3]), pk_f16(w[4], w[5]), pk_f16(w[6], w[7])};
                        gh[fi] = __builtin_bit_cast(h8, vh);
                        __builtin_amdgcn_sched_barrier(0);   // one operand at a time: 8 loads in flight
                    }
                });
            });
        };
        // all MFMAs of u: D[channel][pixel] += X[channel][q] * G[q][pixel]
        auto mma = [&](auto role_c) {
            constexpr int R = decltype(role_c)::value;
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int f0 = frag_idx(R, 0, j), f1 = frag_idx(R, 1, j);
                if constexpr (f0 >= 0 || f1 >= 0) {
                    h8 x[NCT];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        x[ct] = *reinterpret_cast<const h8 *>(smem + X_OFS + (ct >> 1) * XBUF + xb + (ct & 1) * 16 * CHS + j * 64);
                    if (ABL & 1) { asm volatile("" ::"v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3])); return; }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        if constexpr (f0 >= 0) acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[ct], gh[f0 >= 0 ? f0 : 0], acc[0][ct], 0, 0, 0);
                        if constexpr (f1 >= 0) acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[ct], gh[f1 >= 0 ? f1 : 0], acc[1][ct], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        auto gather_d = [&]() {
            auto by_role = [&](auto xp_c) {
                switch (role) {
                case 0: gather(std::integral_constant<int, 0>{}, xp_c); break;
                case 1: gather(std::integral_constant<int, 1>{}, xp_c); break;
                case 2: gather(std::integral_constant<int, 2>{}, xp_c); break;
                default: gather(std::integral_constant<int, 3>{}, xp_c); break;
                }
            };
            if (xpar) by_role(std::integral_constant<int, 1>{});
            else by_role(std::integral_constant<int, 0>{});
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
            __syncthreads();                                   // (B) both X chunks of u complete, the G image is free
            mma_d();                                           // phase 2
            __syncthreads();                                   // (A') the X buffers are free; G(u+1) complete
        }
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
        if (ABL & 32) {
#pragma unroll
            for (int ab = 0; ab < 2; ++ab)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) asm volatile("" ::"v"(acc[ab][ct]));
        } else {
            switch (role) {
            case 0: scatter(std::integral_constant<int, 0>{}); break;
            case 1: scatter(std::integral_constant<int, 1>{}); break;
            case 2: scatter(std::integral_constant<int, 2>{}); break;
            default: scatter(std::integral_constant<int, 3>{}); break;
            }
        }
        __syncthreads();
        store_rows(tk);
        __syncthreads();
    };
    __syncthreads();                                           // (A) G(0) of the first task complete
    for (int t = (int)xcd_remap(blockIdx.x, gridDim.x); t < ntasks; t += gridDim.x) {
        const Task tk = get_task(t);
        if (tk.flip) run_task(tk, std::integral_constant<int, 1>{});
        else run_task(tk, std::integral_constant<int, 0>{});
    }
}

} // namespace hbh

bool corr_f16_bwd_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    if (dtype != FN2_F16) return false;
    if (k != 1 || s1 != 1 || s2 != 2 || pad != md || md != 2 * hbh::DR) return false;
    if (C % hbh::CG != 0 || C < hbh::CG || (H & 1) || (W % 8) != 0 || W > 64) return false;
    if ((long)C * H * W * 2 >= 0x7fffffffL || (long)hbh::D * hbh::D * H * W * 2 >= 0x7fffffffL) return false;   // 32-bit byte offsets
    return true;
}

// in1, in2, gout, g1, g2: half tensors
int corr_backward_f16(const void *in1, const void *in2, const void *gout, void *g1, void *g2, int B, int C, int H, int W, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(gout, 16) || !aligned(g1, 16) || !aligned(g2, 16)) return FN2_EALIGN;
    hbh::Args a;
    a.nbr[0] = static_cast<const _Float16 *>(in2); a.nbr[1] = static_cast<const _Float16 *>(in1);
    a.gout = static_cast<const _Float16 *>(gout);
    a.gin[0] = static_cast<_Float16 *>(g1); a.gin[1] = static_cast<_Float16 *>(g2);
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.NRG = (H / 2 + 3) / 4; a.NCGR = C / hbh::CG;
    a.fC = (float)C; a.rC = 1.0f / (float)C;
    const long ntasks = 2L * B * 2 * a.NRG * a.NCGR;
    if (ntasks == 0) return FN2_OK;
    if (ntasks > 0x3fffffffL) return FN2_EINVAL;
    const unsigned grid = ntasks < 256 ? (unsigned)ntasks : 256u;   // persistent: one workgroup per CU
    hipLaunchKernelGGL(hbh::corr_bwd_f16, dim3(grid), dim3(hbh::NWAVES * 64), 0, s, a);
    return launch_status();
}

} // namespace fn2
