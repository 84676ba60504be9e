// correlation_f16x2_wide.hip -- the f16x2 cost-volume kernel (correlation_f16x2.hip: numerics, wave specialisation, persistent
// task lists) for maps WIDER than 64 pixels: Sintel-size and other real-image inputs, whose conv3 maps are 128 and more pixels
// wide (FlowNetC.py:86 on 1024x436 frames).  The reference kernel has no width limit (correlation_cuda_kernel.cu:73-147).
//
// Round 6 re-tiling.  A step of these kernels is paced by what the staging waves move through the CU's
// vector-memory path and the LDS, not by the matrix cores: what counts is BLOCK PAIRS PER STAGED BLOCK.  On a parity lattice an A
// column block a (4 lattice columns = 8 pixels) meets the B column blocks a-3 .. a+3; rounds 3-5 gave a task a column window of 4 A
// blocks against the 10 B blocks they meet: 28 pairs per 14 staged blocks (the narrow kernel: 44 per 16) -- 1.5-1.7x the narrow
// kernel's time per output.  Now a task takes TWO A row groups against ONE B row block: the B rows 4b - 10 .. 4b - 7 (b = rg + u) are
// the neighbour rows of row group rg with u = b - rg AND of row group rg + 1 with u - 1, so the same staged B tile serves both:
//     slot  0 ..  3   A' blocks 0..3 of row group rg0              (32 pixels, window index xq)
//     slot  4 ..  7   A' blocks 0..3 of row group rg0 + 1
//     slot  8 .. 17   B' blocks 0..9 = image blocks 4 xq - 3 .. 4 xq + 6 of the B rows
// = 56 block pairs per 18 staged blocks (x 12/14 at the two ends of a row-group pair's seven B row blocks, where only one of
// the two A tiles has a displacement row in range).  Which row groups share a B row block b: all of max(0, b - 5) .. min(b, NRG - 1);
// they are paired from the top down -- (hi, hi - 1), (hi - 2, hi - 3), ... -- so that a block of n row groups costs ceil(n / 2) tasks
// (24 per parity and window at 56 x 128 instead of the 42 single-row-group tasks of rounds 3-5).  18 slots of 32 B per channel row + 32 B of padding = 608 B: two step buffers
// are 155 648 B of the CU's 160 KB.  B blocks left or right of the image are zeros from the buffer range check.
// 8 matrix waves = x parity x row group x A' block pair: 14 block pairs, 42 MFMAs per step of 32 channels.  The epilogue image has
// 32 planes (row group, ai, bi) of 32-pixel rows; wave w stores planes w and w + 16.
//
// Staging.  A lane issues four (waves 0-3: five) 32-byte loads per step; the source of a load must be uniform (one buffer
// descriptor per instruction): load 0 = A' of the first row group (in1), all 32 channels x 4 rows x 4 pieces; load 1 = the second
// row group; loads 2, 3 = B' 0..7 in the narrow kernel's mapping (slot k = channels 16k ..); load 4 (waves 0-3) = B' 8, 9.
#include "f16x2_common.h"
#include "f16x2_split.h"

namespace fn2 {
namespace hw {
using namespace hf;
using f16s::exp_stat;
using f16s::scale_exp;
using f16s::split2;
using f16s::to_sgpr;
using f16s::wave_sum;

constexpr int AW = 4;            // A' blocks of a window (32 pixels)
constexpr int NB = 7;            // B' blocks an A' block meets: B' a .. a + 6
constexpr int WPX = 8 * AW;      // pixels per window

// LDS image of a step: [term][x parity][channel][18 slots x (4 rows x 8 B)] with 608-byte channel rows (152 dwords = 24 mod 64: the
// four channel rows of a transposing read fall on disjoint banks, as with the narrow kernel's 288)
constexpr int NSLOT = 2 * AW + 10;
constexpr int RS = NSLOT * 32 + 32;      // 608
constexpr int WPARS = CK * RS;           // 19456
constexpr int WTERM = 2 * WPARS;         // 38912
constexpr int WBUF = 2 * WTERM;          // 77824: one step
constexpr int WLDS = 2 * WBUF;           // 155648: two steps
// epilogue image: 32 planes (row group, ai, bi) x 26 rows (21 + slack) x 32 floats, the 16-byte slots of a row rotated by 8 ai floats.  A
// ds_write_b32 is serviced in two groups of 32 lanes on 32 banks, and a wave writes one x parity only: 2-way is the floor; this layout
// reaches it, the 16-byte row reads (four groups of 16 lanes, 64 banks) are conflict-free.  (The first round-6 layout -- plane stride + 12
// floats, no rotation -- had been checked against a 64-bank model of the write: 4-way in fact, 4.3 M bank-conflict cycles per launch at
// 8 x 256 x 56 x 128, profiles/r06_v_sq_wide.log.)
constexpr int WO_RS = 32, WO_PS = O_DP * WO_RS;
static_assert((32 * WO_PS + O_SLACK * WO_RS) * 4 <= WLDS, "epilogue image must fit the operand buffers");
static_assert(WTERM + (NSLOT - 1) * 32 + 16 * RS + 8 * 3 + 3 * RS < 65536, "fragment offsets are ds_read immediates");

struct ArgsW : Args {
    int NXQ;                     // column windows: ceil(W / 32)
    unsigned magic_x;            // ceil(2^32 / NXQ)
};
struct TaskW { int n, py, rg0, db, xq; };          // A row groups rg0 (may be -1: absent) and rg0 + 1, B row block b = rg0 + db, db = 1 .. 6
struct LoadSetW { u4 a0[2], a1[2], b[2][2], b4; };   // one step of one lane: [half] x 16 B of the two A' tiles, [slot][half] of B' 0..7, 16 B of B' 8..9

// host: the task table of one batch item for row-group PAIRS (f16x2_common.h's entry format: (rg0 + 1) << 4 | py << 3 | db with
// db = b - rg0 = 1 .. 6; "real" = the B rows 4 b - 10 .. + 3 meet the image) and the division constants.  For every B row block b the
// row groups that have it in range, lo = max(0, b - 5) .. hi = min(b, NRG - 1), are paired from the top down.
inline long build_pair_table(ArgsW &a, int B, int H)
{
    const int HL = H / 2, NRG = (HL + 3) / 4;
    int R = 0, P = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int py = 0; py < 2; ++py)
            for (int b = 0; b <= NRG - 1 + NU - 1; ++b) {
                const int lo = b - (NU - 1) > 0 ? b - (NU - 1) : 0, hi = b < NRG - 1 ? b : NRG - 1;
                const int ib0 = 4 * b - DR;
                const bool real = ib0 + 3 >= 0 && ib0 < HL;
                if (real != (pass == 0)) continue;
                for (int rg1 = hi; rg1 >= lo; rg1 -= 2) {           // the pair (rg1 - 1, rg1); rg1 - 1 < lo: the lower one is absent
                    const int db = b - (rg1 - 1);                   // 1 .. 6
                    if (R + P >= MAX_TAB) return FN2_EUNSUPPORTED;
                    const unsigned e = (unsigned)((rg1 << 4) | (py << 3) | db), i = (unsigned)(R + P);
                    a.tab[i >> 1] = (i & 1u) ? (a.tab[i >> 1] | (e << 16)) : e;
                    if (real) ++R; else ++P;
                }
            }
    a.R_item = R; a.P_item = P;
    a.magic_r = R ? (unsigned)((0x100000000ull + R - 1) / R) : 0u;
    a.magic_p = P ? (unsigned)((0x100000000ull + P - 1) / P) : 0u;
    return (long)B * (R + P);
}

__global__ __launch_bounds__(1024, 4) void corr_fwd_f16x2_wide(ArgsW p)
{
    __shared__ __attribute__((aligned(16))) char smem[WLDS + 2048];   // + the operand samples (2 x 1 KB, LDS-DMA)
    __shared__ int scl_k[3];     // as in correlation_f16x2.hip: [0] ka + kb of the current task, [1], [2] ka, kb of the next one

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < 8;
    const int w8 = wave & 7;
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int nsteps = p.C / CK;       // even

    // ---- this workgroup's task list (correlation_f16x2.hip; the column window is the fastest-varying task coordinate)
    const int G = gridDim.x >> 3, strm = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int Rtot = p.B * p.R_item * p.NXQ, Ptot = p.B * p.P_item * p.NXQ;
    const int r0 = (int)((long)strm * Rtot / 8), r1 = (int)((long)(strm + 1) * Rtot / 8);
    const int q0 = (int)((long)strm * Ptot / 8), q1 = (int)((long)(strm + 1) * Ptot / 8);
    const int Rc = r1 - r0, Pc = q1 - q0;
    const int n_real = (Rc - j + G - 1) / G > 0 ? (Rc - j + G - 1) / G : 0;
    const int rem = Rc % G;
    const int pgrp = rem == 0 ? G : G - rem, pj = rem == 0 ? j : j - rem;
    const int n_pad = (pj >= 0 && Pc - pj > 0) ? (Pc - pj + pgrp - 1) / pgrp : 0;
    const int n_tasks = n_real + n_pad;
    auto get_task = [&](int i) -> TaskW {
        const bool real = i < n_real;
        const unsigned kk = (unsigned)(real ? r0 + j + G * i : q0 + pj + pgrp * (i - n_real));
        const unsigned k2 = __umulhi(kk, p.magic_x);            // kk / NXQ (exact below 2^16, checked by the launcher)
        const Task t = decode_task(p, real, (int)k2);
        TaskW w;
        w.n = t.n; w.py = t.py; w.rg0 = t.rg - 1; w.db = t.u;
        w.xq = __builtin_amdgcn_readfirstlane((int)(kk - k2 * (unsigned)p.NXQ));
        return w;
    };

    // ---- write-out of the epilogue image (all 16 waves): wave w owns the planes (row group r, ai, bi) = (r, w >> 2, w & 3), r = 0, 1; a
    // lane owns 16 bytes of the rows ti = (lane >> 3) + 8 i of the window's 32 pixels
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int lgC = pow2 ? 31 - __builtin_clz((unsigned)p.C) : 0;
    float *Os = reinterpret_cast<float *>(smem);
    auto store_rows = [&](const TaskW &tk, int ksum) {
        const int ai = wave >> 2, bi = wave & 3;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 3, xg = 4 * (ln & 7), xw = WPX * tk.xq + xg;
        constexpr int NR = (D + 7) / 8;   // 3 rows per lane, the last one only for g < 5
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (long)tk.n * p.out_bs, 0, (unsigned)(D * D * HW * 4), 0x00020000);
        const unsigned vo = xw < p.W ? (unsigned)((g * HW + xw) * 4) : 0x80000000u;   // out-of-range lanes store nothing
        float f = 1.0f, sl = 1.0f;   // (see correlation_f16x2.hip for why these are copied here)
        if (!pow2) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "s"(p.fC));
        if (p.slope != 1.0f) asm volatile("v_mov_b32 %0, %1" : "=v"(sl) : "s"(p.slope));
        const int kx_mm = -ksum - lgC, kx_ex = -lgC;   // matrix-core sums / sums of the fp32 fallback
        auto finish = [&](f4 val, int kx) {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = __builtin_ldexpf(val[e], kx);
            if (!pow2) { val[0] /= f; val[1] /= f; val[2] /= f; val[3] /= f; }
            if (p.slope != 1.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = val[e] > 0.0f ? val[e] : val[e] * sl;
            }
            return val;
        };
        // the two planes one after the other (both at once would hold 24 registers across the staging waves' in-flight loads of
        // the next task: the kernel has none to spare)
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const int tj = 4 * (tk.db - r) + bi - ai, IL = 4 * (tk.rg0 + r) + ai;   // u of row group r is db - r
            if (tj < 0 || tj >= D || IL < 0 || IL >= HL) continue;                                // the whole plane lies outside the volume (uniform)
            const int y = 2 * IL + tk.py;
            const float *src = Os + (16 * r + wave) * WO_PS + (O_SLACK + g) * WO_RS + ((xg + 8 * ai) & 31);   // + 8 i rows: immediates
            f4 vals[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) vals[i] = *reinterpret_cast<const f4 *>(src + 8 * i * WO_RS);
            const int so0 = (int)((((long)tj * D) * p.H + y) * p.W * 4);                    // row ti = 0 of this plane
            unsigned bad = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const unsigned v = (8 * i + 7 < D || g + 8 * i < D) ? vo : 0x80000000u;          // ti = g + 8 i < 21
                if (__builtin_amdgcn_classf(vals[i][0], 0x207) | __builtin_amdgcn_classf(vals[i][1], 0x207) |
                    __builtin_amdgcn_classf(vals[i][2], 0x207) | __builtin_amdgcn_classf(vals[i][3], 0x207))
                    bad |= 1u << i;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, finish(vals[i], kx_mm)), rso, (int)v, so0 + 8 * i * (int)(HW * 4), 2);
            }
            // an operand did not fit an f16 (or is inf/nan): those outputs are recomputed in fp32 and the row stored again
            if (bad) {
#pragma unroll 1
                for (int i = 0; i < NR; ++i) {
                    const int ti = g + 8 * i;
                    if (!(bad >> i & 1) || ti >= D || xw >= p.W) continue;
                    f4 val = *reinterpret_cast<const f4 *>(src + 8 * i * WO_RS);
#pragma unroll 1
                    for (int e = 0; e < 4; ++e) {
                        const float cur = e == 0 ? val[0] : e == 1 ? val[1] : e == 2 ? val[2] : val[3];
                        const bool nonfin = (__builtin_bit_cast(unsigned, cur) & 0x7f800000u) == 0x7f800000u;
                        const float ex = nonfin ? exact_corr(p, tk.n, y, xw + e, tj, ti) : __builtin_ldexpf(cur, -ksum);
                        val[0] = e == 0 ? ex : val[0]; val[1] = e == 1 ? ex : val[1];
                        val[2] = e == 2 ? ex : val[2]; val[3] = e == 3 ? ex : val[3];
                    }
                    *reinterpret_cast<f4 *>(p.out + (long)tk.n * p.out_bs + ((long)(tj * D + ti) * p.H + y) * p.W + xw) = finish(val, kx_ex);
                }
            }
        }
    };

    if (is_stage) {
        // ================= staging waves =================
        // loads 0 / 1 (the two A' tiles): lane = (channel pair member, channel + 4, row, piece): channels b, b+4 in the lower half-wave,
        // b+1, b+5 in the upper one, b = 8 (w >> 1) + 2 (w & 1)  (4 channels = 96 dwords = 32 mod 64: conflict-free 8-byte writes)
        const int a_piece = lane & 3, s_row = (lane >> 2) & 3;
        const int a_ch = 8 * (w8 >> 1) + 2 * (w8 & 1) + 4 * ((lane >> 4) & 1) + (lane >> 5);
        const int wa_ofs = a_ch * RS + a_piece * 32 + s_row * 8;                   // slots 0..3; the second row group: + 4 * 32
        // loads 2 / 3 (B' 0..7): the narrow kernel's mapping (a half-wave writes 256 contiguous bytes of one channel row)
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_ch = 2 * w8 + (lane >> 5);
        const int wb_ofs = s_ch * RS + (2 * AW + s_piece) * 32 + s_row * 8;
        // load 4 (B' 8, 9): ONE 16-byte load per lane (4 pixels = 2 lattice columns of each parity): lane = (channel, row, block, half)
        const int t_half = lane & 1, t_piece = (lane >> 1) & 1, t_row = (lane >> 2) & 3, t_ch = 4 * w8 + (lane >> 4);
        const int wt_ofs = t_ch * RS + (2 * AW + 8 + t_piece) * 32 + t_row * 8 + 4 * t_half;
        const unsigned nbytes = (unsigned)(p.C * HW * 4);
        __amdgpu_buffer_rsrc_t rs1, rs2;
        unsigned v_offa0, v_offa1, v_offb, v_offt;
        auto set_ctx = [&](const TaskW &tk, bool valid) {
            const int rg0 = tk.rg0, ib0 = 4 * (rg0 + tk.db) - DR;
            const int ila0 = 4 * rg0 + s_row, ila1 = ila0 + 4, ilb = ib0 + s_row, ilt = ib0 + t_row;
            const int xa = WPX * tk.xq + 8 * a_piece, xb = WPX * tk.xq - 24 + 8 * s_piece, xt = WPX * tk.xq + 40 + 8 * t_piece + 4 * t_half;
            const bool act0 = valid && tk.db < NU && rg0 >= 0, act1 = valid && tk.db >= 1;      // the row group exists and has a displacement row in range
            v_offa0 = (act0 && ila0 < HL && xa < p.W) ? (unsigned)((a_ch * HW + (long)(2 * ila0 + tk.py) * p.W + xa) * 4) : 0x80000000u;
            v_offa1 = (act1 && ila1 < HL && xa < p.W) ? (unsigned)((a_ch * HW + (long)(2 * ila1 + tk.py) * p.W + xa) * 4) : 0x80000000u;
            v_offb = (valid && ilb >= 0 && ilb < HL && xb >= 0 && xb < p.W) ? (unsigned)((s_ch * HW + (long)(2 * ilb + tk.py) * p.W + xb) * 4) : 0x80000000u;
            v_offt = (valid && ilt >= 0 && ilt < HL && xt < p.W) ? (unsigned)((t_ch * HW + (long)(2 * ilt + tk.py) * p.W + xt) * 4) : 0x80000000u;
            rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
        };
        // The loads of a step in two groups (register pressure: the kernel has 128 registers per lane, two full sets of a step are 72):
        // group 1 = the two A' tiles and B' 0..7 of channels 0..15 (24 registers), group 2 = B' 0..7 of channels 16..31 and B' 8..9 (12).
        // Group 2 of step s + 2 is issued after the first two items of step s + 1 have been written to LDS and their registers are free.
        auto issue_g1 = [&](LoadSetW &L, int c0) {
            const int so = (int)(c0 * HW * 4);
            L.a0[0] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)v_offa0, so, 0);
            L.a0[1] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(v_offa0 + 16), so, 0);
            L.a1[0] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)v_offa1, so, 0);
            L.a1[1] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(v_offa1 + 16), so, 0);
            L.b[0][0] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb, so, 0);
            L.b[0][1] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(v_offb + 16), so, 0);
        };
        auto issue_g2 = [&](LoadSetW &L, int c0) {
            const int so = (int)(c0 * HW * 4), sk = (int)((c0 + 16) * HW * 4);
            L.b[1][0] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb, sk, 0);
            L.b[1][1] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(v_offb + 16), sk, 0);
            L.b4 = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offt, so, 0);
        };
        // 8 consecutive pixels -> (hi, lo) x (parity 0, parity 1) chunks of 4 lattice columns, scaled by the tile's sc = 2^k
        auto split_write = [&](const u4 &q0, const u4 &q1, char *dst, f16s::scale2_t sc) {
            const f4 x0 = f16s::pk_scale4(__builtin_bit_cast(f4, q0), sc), x1 = f16s::pk_scale4(__builtin_bit_cast(f4, q1), sc);
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                unsigned h01, l01, h23, l23;
                split2(x0[par], x0[2 + par], h01, l01);
                split2(x1[par], x1[2 + par], h23, l23);
                *(FN2_LDS(u2) *)(dst + par * WPARS) = (u2){h01, h23};
                *(FN2_LDS(u2) *)(dst + WTERM + par * WPARS) = (u2){l01, l23};
            }
        };
        f16s::scale2_t sc_a = f16s::scale2_from_exp(0), sc_b = sc_a;
        // 4 pixels = 2 lattice columns of each parity -> 4-byte half chunks (B' 8..9)
        auto split_write_half = [&](const u4 &q0, char *dst, f16s::scale2_t sc) {
            const f4 x0 = f16s::pk_scale4(__builtin_bit_cast(f4, q0), sc);
            unsigned h0, l0, h1, l1;
            split2(x0[0], x0[2], h0, l0);
            split2(x0[1], x0[3], h1, l1);
            *(FN2_LDS(unsigned) *)(dst) = h0;
            *(FN2_LDS(unsigned) *)(dst + WTERM) = l0;
            *(FN2_LDS(unsigned) *)(dst + WPARS) = h1;
            *(FN2_LDS(unsigned) *)(dst + WTERM + WPARS) = l1;
        };
        auto write_g1 = [&](const LoadSetW &L, char *buf) {     // the two A' tiles
            split_write(L.a0[0], L.a0[1], buf + wa_ofs, sc_a);
            __builtin_amdgcn_sched_barrier(0);
            split_write(L.a1[0], L.a1[1], buf + wa_ofs + AW * 32, sc_a);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto write_g2 = [&](const LoadSetW &L, char *buf) {     // B' 0..9
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                split_write(L.b[k][0], L.b[k][1], buf + wb_ofs + k * 16 * RS, sc_b);
                __builtin_amdgcn_sched_barrier(0);
            }
            split_write_half(L.b4, buf + wt_ofs, sc_b);
            __builtin_amdgcn_sched_barrier(0);
        };
        // one phase of the step pipeline: loads of the step after next into `nxt`, the next step's values from `cur` into `buf`
        auto advance = [&](LoadSetW &nxt, int c0, const LoadSetW &cur, char *buf) {
            issue_g1(nxt, c0);
            write_g1(cur, buf);
            issue_g2(nxt, c0);
            write_g2(cur, buf);
        };
        // operand sample of a task (f16x2_split.h): 16 bytes per lane and tile, inside the task's column window -- by LDS-DMA (staging
        // wave 0 only): the sample is in flight across two step phases, and this kernel has no registers to hold it in
        char *smp = smem + WLDS;
        auto sample_issue = [&](const TaskW &tk) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            const int c = (ln * p.C) >> 6, r = ln & 3, sx = (5 * ln) >> 2;
            const int xa = WPX * tk.xq + 4 * (sx & 7), xb = WPX * tk.xq - 16 + 4 * (sx & 15);
            // A rows: the row group that has most displacement rows in range for this B row block (the first one up to db = 3)
            const int rg0 = tk.rg0, rga = (tk.db <= 3 && rg0 >= 0) ? rg0 : rg0 + 1;
            const int ila = 4 * rga + r, ilb = 4 * (rg0 + tk.db) - DR + r;
            const unsigned oa = (ila < HL && xa < p.W) ? (unsigned)((c * HW + (long)(2 * ila + tk.py) * p.W + xa) * 4) : 0x80000000u;
            const unsigned ob = (ilb >= 0 && ilb < HL && xb >= 0 && xb < p.W) ? (unsigned)((c * HW + (long)(2 * ilb + tk.py) * p.W + xb) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (FN2_LDS(void) *)(smp), 16, (int)oa, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (FN2_LDS(void) *)(smp + 1024), 16, (int)ob, 0, 0, 0);
        };
        // ... evaluated after the 18 operand loads issued behind it (vector-memory operations complete in order: at most 18
        // outstanding = the two sample transfers have landed)
        auto sample_scales = [&](int &ka, int &kb) {
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            const u4 sa = *(FN2_LDS(u4) *)(smp + 16 * lane), sb = *(FN2_LDS(u4) *)(smp + 1024 + 16 * lane);
            unsigned ta = 0u, tb = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ta += exp_stat(sa[i]); tb += exp_stat(sb[i]); }
            ka = scale_exp(wave_sum(ta));
            kb = scale_exp(wave_sum(tb));
        };

        // the step pipeline of correlation_f16x2.hip: during step s buffer s&1 is read, register set (s+1)&1 holds step s+1,
        // set s&1 receives step s+2 (of this task or, in the last two steps, of the next real one)
        LoadSetW L0, L1;
        int ka_n = 0, kb_n = 0;
        if (n_real > 0) {
            if (wave == 0) sample_issue(get_task(0));
            set_ctx(get_task(0), true);
            issue_g1(L0, 0); issue_g2(L0, 0);
            issue_g1(L1, CK); issue_g2(L1, CK);
            if (wave == 0) {
                sample_scales(ka_n, kb_n);
                if (lane == 0) { scl_k[1] = ka_n; scl_k[2] = kb_n; }
            }
        }
        __syncthreads();   // the first task's scale exponents are published (the matrix waves take part in this barrier too)
        if (n_real > 0 && wave != 0) { ka_n = to_sgpr(scl_k[1]); kb_n = to_sgpr(scl_k[2]); }
        for (int it = 0; it < n_real; ++it) {
            const TaskW tk = get_task(it);
            const bool has_next = it + 1 < n_real;
            const int ksum = ka_n + kb_n;
            sc_a = f16s::scale2_from_exp(ka_n); sc_b = f16s::scale2_from_exp(kb_n);
            if (tid == 0) scl_k[0] = ksum;
            write_g1(L0, smem); write_g2(L0, smem);
            __syncthreads();
            for (int s = 0; s + 2 < nsteps; s += 2) {
                advance(L0, (s + 2) * CK, L1, smem + WBUF);
                __syncthreads();
                advance(L1, (s + 3) * CK, L0, smem);
                __syncthreads();
            }
            if (has_next && wave == 0) sample_issue(get_task(it + 1));
            set_ctx(get_task(has_next ? it + 1 : it), has_next);
            advance(L0, 0, L1, smem + WBUF);
            __syncthreads();
            issue_g1(L1, CK); issue_g2(L1, CK);
            __syncthreads();
            if (has_next && wave == 0) {
                sample_scales(ka_n, kb_n);
                if (lane == 0) { scl_k[1] = ka_n; scl_k[2] = kb_n; }
            }
            __syncthreads();   // the epilogue image is complete
            if (has_next && wave != 0) { ka_n = to_sgpr(scl_k[1]); kb_n = to_sgpr(scl_k[2]); }
            store_rows(tk, ksum);
            __syncthreads();   // ... and has been read: the buffers are free
        }
        for (int it = n_real; it < n_tasks; ++it) {   // zero-only tasks
            __syncthreads();
            store_rows(get_task(it), 0);
            __syncthreads();
        }
        return;
    }

    // ================= matrix-core waves =================
    __builtin_amdgcn_s_setprio(2);
    const int xpar = w8 & 1;
    const int rsel = __builtin_amdgcn_readfirstlane((w8 >> 1) & 1);   // the wave's A row group (0: rg0, 1: rg0 + 1)
    const int half = __builtin_amdgcn_readfirstlane(w8 >> 2);         // ... and its A' block pair {2 half, 2 half + 1}
    // slots of the wave's operands: A' block ab -> 4 rsel + 2 half + ab, B' block jj -> 8 + 2 half + jj (jj = 0..7): both move with
    // `half` by the same 64 bytes, so one base register serves all fragment reads (offsets are immediates)
    const int r_base = xpar * WPARS + (4 * (lane >> 4) + ((lane & 15) >> 2)) * RS + (lane & 3) * 8 + 64 * half;
    const int ra_base = r_base + 128 * rsel;
    auto frag_at = [&](const char *ptr) -> h8 {
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr + 16 * RS));
        return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    constexpr int NP2 = 2 * NB;      // 14 block pairs: acc[7 ab + dmi] = A' block ab x B' block jj = ab + dmi
    f4 acc[NP2];
    __syncthreads();                 // (the staging waves publish the first task's scale exponents)
    // One step: D = (in2 block) x (in1 block): rows = B pixels (bi = lane>>4, bj = register), columns = A pixels (lane & 15).
    // The B' blocks are taken two at a time (consecutive MFMAs use different accumulators), the next two are fetched meanwhile.
    auto step = [&](const char *cur) {
        h8 ah[2], al[2];
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) { ah[ab] = frag_at(cur + ra_base + ab * 32); al[ab] = frag_at(cur + ra_base + WTERM + ab * 32); }
        h8 bh[2][2], bl[2][2];
        auto fetch = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                bh[g & 1][e] = frag_at(cur + r_base + (2 * AW + 2 * g + e) * 32);
                bl[g & 1][e] = frag_at(cur + r_base + WTERM + (2 * AW + 2 * g + e) * 32);
            }
        };
        fetch(std::integral_constant<int, 0>{});
        static_for<0, 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < 4) fetch(std::integral_constant<int, g + 1>{});
            static_for<0, 3>([&](auto prc) {
                constexpr int pr = decltype(prc)::value;
                static_for<0, 2>([&](auto ec) {
                    constexpr int e = decltype(ec)::value, jj = 2 * g + e;
                    static_for<0, 2>([&](auto abc) {
                        constexpr int ab = decltype(abc)::value, dmi = jj - ab;
                        if constexpr (dmi >= 0 && dmi < NB)
                            acc[NB * ab + dmi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? bl[g & 1][e] : bh[g & 1][e], pr == 1 ? al[ab] : ah[ab],
                                                                                        acc[NB * ab + dmi], 0, 0, 0);
                    });
                });
            });
        });
    };
    // epilogue, first half: accumulators -> LDS [plane = 16 rsel + 4 ai + bi][ti + slack][x]
    auto scatter = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e_ai = (ln & 15) >> 2, e_aj = ln & 3, e_bi = ln >> 4;
        const int rbase = (16 * rsel + 4 * e_ai + e_bi) * WO_PS + (O_SLACK + DR - 12 - e_aj) * WO_RS;     // row of (dm = -3, r = 0)
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
            float *dst = Os + rbase + ((8 * (2 * half + ab) + 2 * e_aj + xpar + 8 * e_ai) & 31);
#pragma unroll
            for (int dmi = 0; dmi < NB; ++dmi)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(4 * dmi + r) * WO_RS] = acc[NB * ab + dmi][r];   // ti = 4 (dmi - 3) + r - e_aj + DR
        }
    };
    auto active = [&](const TaskW &tk) { return rsel ? tk.db >= 1 : (tk.db < NU && tk.rg0 >= 0); };   // this wave's row group has a displacement row in range
    auto epilogue = [&](const TaskW &tk, int ksum) {
        if (active(tk)) scatter();             // (the planes of the other case are never stored)
        __syncthreads();
        store_rows(tk, ksum);
        __syncthreads();
    };
    for (int it = 0; it < n_real; ++it) {
        const TaskW tk = get_task(it);
        const bool act = active(tk);
#pragma unroll
        for (int i = 0; i < NP2; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        const int ksum = to_sgpr(scl_k[0]);
        for (int s = 0; s < nsteps; s += 2) {
            if (act) step(smem);
            __syncthreads();
            if (act) step(smem + WBUF);
            __syncthreads();
        }
        epilogue(tk, ksum);
    }
#pragma unroll
    for (int i = 0; i < NP2; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = n_real; it < n_tasks; ++it) epilogue(get_task(it), 0);   // zero-only tasks
}

} // namespace hw

// maps wider than 64 pixels (called by corr_forward_f16x2; same preconditions otherwise)
int corr_forward_f16x2_wide(const float *in1, const float *in2, float *out, long out_bs, float slope, int B, int C, int H, int W,
                            hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(out, 16) || (out_bs % 4) != 0) return FN2_EALIGN;
    hw::ArgsW a;
    a.in1 = in1; a.in2 = in2; a.out = out; a.out_bs = out_bs; a.slope = slope;
    a.fC = (float)C; a.rC = 1.0f / (float)C;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.dbg = nullptr;
    a.NXQ = (W + hw::WPX - 1) / hw::WPX;
    a.magic_x = (unsigned)((0x100000000ull + a.NXQ - 1) / a.NXQ);
    const long per_window = hw::build_pair_table(a, B, H);
    if (per_window < 0) return (int)per_window;
    if ((long)B * (a.R_item > a.P_item ? a.R_item : a.P_item) * a.NXQ >= 65536) return FN2_EUNSUPPORTED;   // exact magic divisions
    const long ntasks = per_window * a.NXQ;
    if (ntasks == 0) return FN2_OK;
    const long per_stream = (ntasks + 7) / 8;
    const int G = per_stream < 32 ? (int)per_stream : 32;
    hipLaunchKernelGGL(hw::corr_fwd_f16x2_wide, dim3(8u * G), dim3(1024), 0, s, a);
    return launch_status();
}

} // namespace fn2
