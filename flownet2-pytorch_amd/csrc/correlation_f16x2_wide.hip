// correlation_f16x2_wide.hip -- the f16x2 cost-volume kernel (correlation_f16x2.hip: numerics, LDS image, wave specialisation,
// persistent task lists) for maps WIDER than 64 pixels: Sintel-size and other real-image inputs, whose conv3 maps are 128 and
// more pixels wide (FlowNetC.py:86 on 1024x436 frames).  The reference kernel has no width limit
// (correlation_cuda_kernel.cu:73-147); round 2 sent such maps to the fp32 matrix-core kernel (2.5-3x slower per output).
//
// What changes with the width.  On a parity lattice an A column block a (4 lattice columns = 8 pixels) meets the B column
// blocks a-3 .. a+3.  Up to 64 pixels both tiles of a task are the whole row (8 + 8 blocks).  Beyond that a task takes a
// COLUMN WINDOW: 4 A blocks (32 pixels, window index xq) against the 10 B blocks 4xq-3 .. 4xq+6 they meet -- with two spare
// slots the same 16 block slots per channel row, i.e. exactly the LDS image of the narrow kernel:
//     slot  0 ..  3   A' blocks 0..3              (tile 0, blocks 0..3)
//     slot  4 ..  7   B' blocks 0..3              (tile 0, blocks 4..7)
//     slot  8 .. 15   B' blocks 4..11             (tile 1; B' 10 and 11 are never used and never loaded)
// B blocks left or right of the image are zeros from the buffer range check (their products are computed, not skipped: the
// window position is a run-time value).  8 matrix waves = x parity x A' block: 7 block pairs, 21 MFMAs per step of 32 channels
// (33 in the narrow kernel, for the same staging work: 28 block pairs per 16 staged blocks instead of 44 -- a wide map costs
// ~1.5x the narrow kernel's time per output).  Epilogue rows are 32 pixels (128 B per lane group of 8).
//
// Staging.  A lane still issues four 32-byte loads per step, but the source of a load must be uniform (one buffer descriptor
// per instruction), so the lanes are re-mapped: load 0 = A' (in1), all 32 channels of the step x 4 rows x 4 pieces; load 1 =
// B' 0..3 (in2), same mapping; loads 2, 3 = B' 4..11, the narrow kernel's mapping (slot k = channels 16k ..).  In loads 0/1 a
// half-wave holds channels c and c + 4 (bank offset 32 dwords: conflict-free 8-byte writes).
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

struct ArgsW : Args {
    int NXQ;                     // column windows: ceil(W / 32)
    unsigned magic_x;            // ceil(2^32 / NXQ)
};
struct TaskW { int n, py, rg, u, xq; };
struct LoadSetW { u4 a[2], b0[2], b1[2][2]; };   // one step of one lane: [half] x 16 B of A', of B' 0..3, [slot][half] of B' 4..11

__global__ __launch_bounds__(1024, 4) void corr_fwd_f16x2_wide(ArgsW p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
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
        w.n = t.n; w.py = t.py; w.rg = t.rg; w.u = t.u;
        w.xq = __builtin_amdgcn_readfirstlane((int)(kk - k2 * (unsigned)p.NXQ));
        return w;
    };

    // ---- write-out of the epilogue image (all 16 waves): wave w owns plane w = (ai, bi); a lane owns 16 bytes of the rows
    // ti = (lane >> 3) + 8 i of the window's 32 pixels
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int lgC = pow2 ? 31 - __builtin_clz((unsigned)p.C) : 0;
    float *Os = reinterpret_cast<float *>(smem);
    auto store_rows = [&](const TaskW &tk, int ksum) {
        const int pl = wave, ai = pl >> 2, bi = pl & 3;
        const int tj = 4 * tk.u + bi - ai, IL = 4 * tk.rg + ai;
        if (tj < 0 || tj >= D || IL >= HL) return;             // the whole plane lies outside the volume (uniform)
        const int y = 2 * IL + tk.py;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 3, xg = 4 * (ln & 7), xw = WPX * tk.xq + xg;
        constexpr int NR = (D + 7) / 8;   // 3 rows per lane, the last one only for g < 5
        const float *src = Os + (pl * O_DP + O_SLACK + g) * O_RS + ((xg + 4 * (4 * bi + ai)) & 63);   // + 8 i rows: immediates
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (long)tk.n * p.out_bs, 0, (unsigned)(D * D * HW * 4), 0x00020000);
        const unsigned vo = xw < p.W ? (unsigned)((g * HW + xw) * 4) : 0x80000000u;   // out-of-range lanes store nothing
        const int so0 = (int)((((long)tj * D) * p.H + y) * p.W * 4);                    // row ti = 0 of this plane
        f4 vals[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) vals[i] = *reinterpret_cast<const f4 *>(src + 8 * i * O_RS);
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
                f4 val = *reinterpret_cast<const f4 *>(src + 8 * i * O_RS);
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
    };

    if (is_stage) {
        // ================= staging waves =================
        // loads 0 / 1 (A', B' 0..3): lane = (channel pair member, channel + 4, row, piece): channels b, b+4 in the lower half-wave,
        // b+1, b+5 in the upper one, b = 8 (w >> 1) + 2 (w & 1)
        const int a_piece = lane & 3, s_row = (lane >> 2) & 3;
        const int a_ch = 8 * (w8 >> 1) + 2 * (w8 & 1) + 4 * ((lane >> 4) & 1) + (lane >> 5);
        const int wa_ofs = a_ch * CHS + a_piece * 32 + s_row * 8;                  // A': tile 0, blocks 0..3; B' 0..3: + 128
        // loads 2 / 3 (B' 4..11 = tile 1): the narrow kernel's mapping
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_ch = 2 * w8 + (lane >> 5);
        const int wb_ofs = TILE + s_ch * CHS + s_piece * 32 + s_row * 8;
        const unsigned nbytes = (unsigned)(p.C * HW * 4);
        __amdgpu_buffer_rsrc_t rs1, rs2;
        unsigned v_offa, v_offb0, v_offb1;
        auto set_ctx = [&](const TaskW &tk, bool valid) {
            const int ib0 = 4 * tk.rg - DR + 4 * tk.u;
            const int ila = 4 * tk.rg + s_row, ilb = ib0 + s_row;
            const int xa = WPX * tk.xq + 8 * a_piece, xb0 = WPX * tk.xq - 24 + 8 * a_piece, xb1 = WPX * tk.xq + 8 + 8 * s_piece;
            const bool okb = valid && ilb >= 0 && ilb < HL;
            v_offa = (valid && ila < HL && xa < p.W) ? (unsigned)((a_ch * HW + (long)(2 * ila + tk.py) * p.W + xa) * 4) : 0x80000000u;
            v_offb0 = (okb && xb0 >= 0 && xb0 < p.W) ? (unsigned)((a_ch * HW + (long)(2 * ilb + tk.py) * p.W + xb0) * 4) : 0x80000000u;
            v_offb1 = (okb && s_piece < 6 && xb1 < p.W) ? (unsigned)((s_ch * HW + (long)(2 * ilb + tk.py) * p.W + xb1) * 4) : 0x80000000u;
            rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
        };
        auto issue_loads = [&](LoadSetW &L, int c0) {
            const int so = (int)(c0 * HW * 4);
            L.a[0] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)v_offa, so, 0);
            L.a[1] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(v_offa + 16), so, 0);
            L.b0[0] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb0, so, 0);
            L.b0[1] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(v_offb0 + 16), so, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int sk = (int)((c0 + 16 * k) * HW * 4);
                L.b1[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb1, sk, 0);
                L.b1[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(v_offb1 + 16), sk, 0);
            }
        };
        // 8 consecutive pixels -> (hi, lo) x (parity 0, parity 1) chunks of 4 lattice columns, scaled by the tile's sc = 2^k
        auto split_write = [&](const u4 &q0, const u4 &q1, char *dst, f16s::scale2_t sc) {
            const f4 x0 = f16s::pk_scale4(__builtin_bit_cast(f4, q0), sc), x1 = f16s::pk_scale4(__builtin_bit_cast(f4, q1), sc);
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                unsigned h01, l01, h23, l23;
                split2(x0[par], x0[2 + par], h01, l01);
                split2(x1[par], x1[2 + par], h23, l23);
                *(FN2_LDS(u2) *)(dst + par * PARS) = (u2){h01, h23};
                *(FN2_LDS(u2) *)(dst + TERM + par * PARS) = (u2){l01, l23};
            }
        };
        f16s::scale2_t sc_a = f16s::scale2_from_exp(0), sc_b = sc_a;
        auto stage_write = [&](const LoadSetW &L, char *buf) {
            split_write(L.a[0], L.a[1], buf + wa_ofs, sc_a);
            __builtin_amdgcn_sched_barrier(0);
            split_write(L.b0[0], L.b0[1], buf + wa_ofs + 4 * 32, sc_b);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                split_write(L.b1[k][0], L.b1[k][1], buf + wb_ofs + k * 16 * CHS, sc_b);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // operand sample of a task (f16x2_split.h): one 16-byte load per lane and tile, inside the task's column window
        struct Samp { u4 a, b; };
        auto sample_issue = [&](const TaskW &tk, Samp &S) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            const int c = (ln * p.C) >> 6, r = ln & 3, sx = (5 * ln) >> 2;
            const int xa = WPX * tk.xq + 4 * (sx & 7), xb = WPX * tk.xq - 16 + 4 * (sx & 15);
            const int ila = 4 * tk.rg + r, ilb = 4 * tk.rg - DR + 4 * tk.u + r;
            const unsigned oa = (ila < HL && xa < p.W) ? (unsigned)((c * HW + (long)(2 * ila + tk.py) * p.W + xa) * 4) : 0x80000000u;
            const unsigned ob = (ilb >= 0 && ilb < HL && xb >= 0 && xb < p.W) ? (unsigned)((c * HW + (long)(2 * ilb + tk.py) * p.W + xb) * 4) : 0x80000000u;
            S.a = __builtin_amdgcn_raw_buffer_load_b128(r1, (int)oa, 0, 0);
            S.b = __builtin_amdgcn_raw_buffer_load_b128(r2, (int)ob, 0, 0);
        };
        auto sample_scales = [&](const Samp &S, int &ka, int &kb) {
            unsigned ta = 0u, tb = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ta += exp_stat(S.a[i]); tb += exp_stat(S.b[i]); }
            ka = scale_exp(wave_sum(ta));
            kb = scale_exp(wave_sum(tb));
        };

        // the step pipeline of correlation_f16x2.hip: during step s buffer s&1 is read, register set (s+1)&1 holds step s+1,
        // set s&1 receives step s+2 (of this task or, in the last two steps, of the next real one)
        LoadSetW L0, L1;
        Samp SM;
        int ka_n = 0, kb_n = 0;
        if (n_real > 0) {
            sample_issue(get_task(0), SM);
            set_ctx(get_task(0), true);
            issue_loads(L0, 0);
            issue_loads(L1, CK);
            sample_scales(SM, ka_n, kb_n);
        }
        for (int it = 0; it < n_real; ++it) {
            const TaskW tk = get_task(it);
            const bool has_next = it + 1 < n_real;
            const int ksum = ka_n + kb_n;
            sc_a = f16s::scale2_from_exp(ka_n); sc_b = f16s::scale2_from_exp(kb_n);
            if (tid == 0) scl_k[0] = ksum;
            stage_write(L0, smem);
            __syncthreads();
            for (int s = 0; s + 2 < nsteps; s += 2) {
                issue_loads(L0, (s + 2) * CK);
                stage_write(L1, smem + BUF);
                __syncthreads();
                issue_loads(L1, (s + 3) * CK);
                stage_write(L0, smem);
                __syncthreads();
            }
            if (has_next && wave == 0) sample_issue(get_task(it + 1), SM);
            set_ctx(get_task(has_next ? it + 1 : it), has_next);
            issue_loads(L0, 0);
            stage_write(L1, smem + BUF);
            __syncthreads();
            issue_loads(L1, CK);
            __syncthreads();
            if (has_next && wave == 0) {
                sample_scales(SM, ka_n, kb_n);
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
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);   // the wave's A' block
    const int r_base = xpar * PARS + (4 * (lane >> 4) + ((lane & 15) >> 2)) * CHS + (lane & 3) * 8;
    auto frag = [&](const char *buf, int slot, int term) -> h8 {   // block slot 0..15 of the channel rows (see the header)
        const char *ptr = buf + r_base + (slot >> 3) * TILE + term * TERM + (slot & 7) * 32;
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr + 16 * CHS));
        return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f4 acc[NB];
    // One step: D = (in2 block) x (in1 block): rows = B pixels (bi = lane>>4, bj = register), columns = A pixels (lane & 15).
    // The B' blocks are taken two at a time (consecutive MFMAs use different accumulators), the next two are fetched meanwhile.
    auto step = [&](auto role_c, const char *cur) {
        constexpr int R = decltype(role_c)::value;
        const h8 ah = frag(cur, R, 0), al = frag(cur, R, 1);
        h8 bh[2][2], bl[2][2];
        auto fetch = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (2 * g + e < NB) { bh[g & 1][e] = frag(cur, AW + R + 2 * g + e, 0); bl[g & 1][e] = frag(cur, AW + R + 2 * g + e, 1); }
        };
        fetch(std::integral_constant<int, 0>{});
        static_for<0, (NB + 1) / 2>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (2 * (g + 1) < NB) fetch(std::integral_constant<int, g + 1>{});
            static_for<0, 3>([&](auto prc) {
                constexpr int pr = decltype(prc)::value;
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    if (2 * g + e < NB)
                        acc[2 * g + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? bl[g & 1][e] : bh[g & 1][e], pr == 1 ? al : ah, acc[2 * g + e], 0, 0, 0);
            });
        });
    };
    auto step_dispatch = [&](const char *cur) {
        switch (role) {
        case 0: step(std::integral_constant<int, 0>{}, cur); break;
        case 1: step(std::integral_constant<int, 1>{}, cur); break;
        case 2: step(std::integral_constant<int, 2>{}, cur); break;
        default: step(std::integral_constant<int, 3>{}, cur); break;
        }
    };
    // epilogue, first half: accumulators -> LDS [plane = 4 ai + bi][ti + slack][x], 16-byte slots rotated by 4 bi + ai
    auto scatter = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e_ai = (ln & 15) >> 2, e_aj = ln & 3, e_bi = ln >> 4;
        const int rot = 4 * (4 * e_bi + e_ai);
        const int rbase = ((4 * e_ai + e_bi) * O_DP + O_SLACK + DR - 12 - e_aj) * O_RS;     // row of (dm = -3, r = 0)
        float *dst = Os + rbase + ((8 * role + 2 * e_aj + xpar + rot) & 63);
#pragma unroll
        for (int dmi = 0; dmi < NB; ++dmi)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(4 * dmi + r) * O_RS] = acc[dmi][r];   // ti = 4 (dmi - 3) + r - e_aj + DR
    };
    auto epilogue = [&](const TaskW &tk, int ksum) {
        scatter();
        __syncthreads();
        store_rows(tk, ksum);
        __syncthreads();
    };
    for (int it = 0; it < n_real; ++it) {
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        const int ksum = to_sgpr(scl_k[0]);
        for (int s = 0; s < nsteps; s += 2) {
            step_dispatch(smem);
            __syncthreads();
            step_dispatch(smem + BUF);
            __syncthreads();
        }
        epilogue(get_task(it), ksum);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
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
    const long per_window = hf::build_task_table(a, B, H);
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
