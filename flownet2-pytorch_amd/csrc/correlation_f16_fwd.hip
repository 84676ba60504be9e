// correlation_f16_fwd.hip -- FlowNetC cost volume (correlation forward) for HALF-precision tensors on the gfx950 f16 matrix
// cores.
//
// The reference dispatches its correlation kernels for at::Half too (correlation_cuda_kernel.cu:386-415: product in half,
// float accumulation :112,:124, output T(acc / C) :139-143); FlowNet2's fp16 inference (BASELINE configs[3], models.py:44-49)
// wraps the layer in tofp32 / tofp16 instead only because that path is broken.  Half inputs ARE f16 matrix
// operands: no split, no scale, ONE v_mfma_f32_16x16x32_f16 per block product where the fp32 kernel needs three, products exact in
// fp32 (the reference rounds each product to half first: this kernel is the more accurate of the two), half the input and
// output bytes.  Same configuration as correlation_f16x2.hip (kernel_size 1, stride1 1, stride2 2, pad == max_displacement ==
// 20, maps up to 64 wide), same task decomposition, LDS image, wave roles and epilogue -- see that file; what differs:
//   - a step is 64 channels: the LDS image [tile][term][parity][channel 32][column block][row] of the fp32 kernel holds channels
//     0..31 / 32..63 of the step where it held the hi / lo terms, so a task of 256 channels is 4 steps (4 barriers) of 22 MFMAs
//     per matrix wave;
//   - staging: one 16-byte load = 8 pixels of a row = the two parity chunks (4 lattice columns each) after four v_perm_b32 -- no
//     conversion arithmetic at all;
//   - the epilogue multiplies by 1/C (divides for a C that is no power of two), applies the optional LeakyReLU in fp32, rounds
//     to half (round to nearest even, as T(acc / C)) and stores 8 bytes per lane;
//   - no out-of-range path: every half value, inf and nan included, is its own matrix operand.
// C % 128 == 0 (an even number of steps), H even, W % 8 == 0, 16-byte aligned tensors; other half shapes take the general
// kernel (correlation_direct.hip).  Maps wider than 64 pixels: the windowed variant at the end of this file.
#include "f16x2_common.h"

namespace fn2 {
namespace hh {
using namespace hf;

typedef ArgsT<_Float16> ArgsH;
constexpr int CKH = 2 * CK;                       // channels per step
struct LoadSetH { u4 a[4], b[4]; };               // one step of one lane: slot k = channels 16k .. 16k+15, 8 pixels (16 B) each tile

__global__ __launch_bounds__(1024, 4) void corr_fwd_f16(ArgsH p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < 8;    // waves 0-7 load and fill the LDS buffers; waves 8-15 run the matrix cores
    const int w8 = wave & 7;
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int nsteps = p.C / CKH;      // even (launcher)

    // ---- this workgroup's task list (as correlation_f16x2.hip: 8 streams = XCDs, one batch-item share each)
    const int G = gridDim.x >> 3, strm = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int Rtot = p.B * p.R_item, Ptot = p.B * p.P_item;
    const int r0 = (int)((long)strm * Rtot / 8), r1 = (int)((long)(strm + 1) * Rtot / 8);
    const int q0 = (int)((long)strm * Ptot / 8), q1 = (int)((long)(strm + 1) * Ptot / 8);
    const int Rc = r1 - r0, Pc = q1 - q0;
    const int n_real = (Rc - j + G - 1) / G > 0 ? (Rc - j + G - 1) / G : 0;
    const int rem = Rc % G;
    const int pgrp = rem == 0 ? G : G - rem, pj = rem == 0 ? j : j - rem;
    const int n_pad = (pj >= 0 && Pc - pj > 0) ? (Pc - pj + pgrp - 1) / pgrp : 0;
    const int n_tasks = n_real + n_pad;
    auto get_task = [&](int i) -> Task {
        if (i < n_real) return decode_task(p, true, r0 + j + G * i);
        return decode_task(p, false, q0 + pj + pgrp * (i - n_real));
    };

    // ---- write-out of the epilogue image (all 16 waves): wave w owns plane w = (ai, bi); a lane owns 4 pixels of the rows
    // ti = (lane >> 4) + 4 i; fp32 image in LDS -> scaled, activated, rounded to half -> 8-byte buffer stores (128-byte rows)
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    float *Os = reinterpret_cast<float *>(smem);
    auto store_rows = [&](const Task &tk) {
        const int pl = wave, ai = pl >> 2, bi = pl & 3;
        const int tj = 4 * tk.u + bi - ai, IL = 4 * tk.rg + ai;
        if (tj < 0 || tj >= D || IL >= HL) return;             // the whole plane lies outside the volume (uniform)
        const int y = 2 * IL + tk.py;
        int ln = lane;
        asm volatile("" : "+v"(ln));   // keeps the lane geometry from being hoisted out of the task loop
        const int g = ln >> 4, xg = 4 * (ln & 15);
        constexpr int NR = (D + 3) / 4;
        const float *src = Os + (pl * O_DP + O_SLACK + g) * O_RS + ((xg + 4 * (4 * bi + ai)) & 63);
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (long)tk.n * p.out_bs, 0, (unsigned)(D * D * HW * 2), 0x00020000);
        const unsigned vo = xg < p.W ? (unsigned)((g * HW + xg) * 2) : 0x80000000u;   // out-of-range lanes store nothing
        const int so0 = (int)((((long)tj * D) * p.H + y) * p.W * 2);                    // row ti = 0 of this plane
        f4 vals[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) vals[i] = *reinterpret_cast<const f4 *>(src + 4 * i * O_RS);
        // copied from the kernel arguments (SGPRs) once per call, after the LDS reads and BEFORE the first store (see
        // correlation_f16x2.hip for why not between the stores)
        float r, f = 1.0f, sl = 1.0f;
        asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(p.rC));
        if (!pow2) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "s"(p.fC));
        if (p.slope != 1.0f) asm volatile("v_mov_b32 %0, %1" : "=v"(sl) : "s"(p.slope));
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned v = (4 * i + 3 < D || g == 0) ? vo : 0x80000000u;          // ti = g + 4 i < 21
            f4 val = vals[i];
            if (pow2) { val[0] *= r; val[1] *= r; val[2] *= r; val[3] *= r; }
            else { val[0] /= f; val[1] /= f; val[2] /= f; val[3] /= f; }
            if (p.slope != 1.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = val[e] > 0.0f ? val[e] : val[e] * sl;
            }
            const u2 packed = {pk_f16(val[0], val[1]), pk_f16(val[2], val[3])};
            __builtin_amdgcn_raw_buffer_store_b64(packed, rso, (int)v, so0 + 4 * i * (int)(HW * 2), 2);
        }
    };

    if (is_stage) {
        // ================= staging waves =================
        // A step has 64 channels x 4 rows x 8 pieces (8 pixels = 16 B = 4 lattice columns of each parity) per tile; slot k of a
        // lane covers channels 16k + 2w + (lane >> 5).  Lane = (channel, piece >> 2, row, piece & 3) as in the fp32 kernel.
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_row = (lane >> 2) & 3;
        const int s_ch = 2 * w8 + (lane >> 5);
        const int s_x = 8 * s_piece;
        const int w_ofs = s_ch * CHS + s_piece * 32 + s_row * 8;   // this lane's chunk inside a (tile, channel half, parity, slot) plane
        const unsigned nbytes = (unsigned)(p.C * HW * 2);
        __amdgpu_buffer_rsrc_t rs1, rs2;
        unsigned v_offa, v_offb;
        auto set_ctx = [&](const Task &tk, bool valid) {
            const int ib0 = 4 * tk.rg - DR + 4 * tk.u;
            const int s_ila = 4 * tk.rg + s_row, s_ilb = ib0 + s_row;
            const bool s_oka = valid && (s_ila < HL) && (s_x < p.W);
            const bool s_okb = valid && (s_ilb >= 0) && (s_ilb < HL) && (s_x < p.W);
            v_offa = s_oka ? (unsigned)((s_ch * HW + (long)(2 * s_ila + tk.py) * p.W + s_x) * 2) : 0x80000000u;
            v_offb = s_okb ? (unsigned)((s_ch * HW + (long)(2 * s_ilb + tk.py) * p.W + s_x) * 2) : 0x80000000u;
            rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
        };
        auto issue_loads = [&](LoadSetH &L, int c0) {   // rows outside the image: out-of-range offset, the load returns zeros
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int soff = (int)((c0 + 16 * k) * HW * 2);
                L.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)v_offa, soff, 0);
                L.b[k] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb, soff, 0);
            }
        };
        // 8 consecutive pixels (p0..p7 as four dwords) -> the chunks (p0,p2,p4,p6) and (p1,p3,p5,p7) of the two parities
        auto perm_write = [&](const u4 &q, char *dst) {
            const u2 even = {__builtin_amdgcn_perm(q[1], q[0], 0x05040100u), __builtin_amdgcn_perm(q[3], q[2], 0x05040100u)};
            const u2 odd = {__builtin_amdgcn_perm(q[1], q[0], 0x07060302u), __builtin_amdgcn_perm(q[3], q[2], 0x07060302u)};
            *(FN2_LDS(u2) *)(dst) = even;
            *(FN2_LDS(u2) *)(dst + PARS) = odd;
        };
        auto stage_write = [&](const LoadSetH &L, char *buf) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // slot k: channel half k >> 1 (the fp32 kernel's "term" index), channels 16 (k & 1) + s_ch of it
                char *dst = buf + w_ofs + (k >> 1) * TERM + (k & 1) * 16 * CHS;
                perm_write(L.a[k], dst);
                __builtin_amdgcn_sched_barrier(0);   // one item at a time (register budget of the staging branch)
                perm_write(L.b[k], dst + TILE);
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // Invariant at the top of a real task: its steps 0 and 1 are in flight in L0 and L1 (correlation_f16x2.hip)
        LoadSetH L0, L1;
        if (n_real > 0) {
            set_ctx(get_task(0), true);
            issue_loads(L0, 0);
            issue_loads(L1, CKH);
        }
        for (int it = 0; it < n_real; ++it) {
            const Task tk = get_task(it);
            const bool has_next = it + 1 < n_real;
            stage_write(L0, smem);
            __syncthreads();
            for (int s = 0; s + 2 < nsteps; s += 2) {
                issue_loads(L0, (s + 2) * CKH);
                stage_write(L1, smem + BUF);
                __syncthreads();
                issue_loads(L1, (s + 3) * CKH);
                stage_write(L0, smem);
                __syncthreads();
            }
            set_ctx(get_task(has_next ? it + 1 : it), has_next);
            issue_loads(L0, 0);
            stage_write(L1, smem + BUF);
            __syncthreads();
            issue_loads(L1, CKH);
            __syncthreads();
            __syncthreads();   // the epilogue image is complete
            store_rows(tk);
            __syncthreads();   // ... and has been read: the buffers are free
        }
        for (int it = n_real; it < n_tasks; ++it) {   // zero-only tasks
            __syncthreads();
            store_rows(get_task(it));
            __syncthreads();
        }
        return;
    }

    // ================= matrix-core waves =================
    __builtin_amdgcn_s_setprio(2);
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);
    const int r_base = xpar * PARS + (4 * (lane >> 4) + ((lane & 15) >> 2)) * CHS + (lane & 3) * 8;
    auto frag = [&](const char *buf, int tile, int half, int blk) -> h8 {   // 32 channels of one pixel block: two transposing reads
        const char *ptr = buf + r_base + tile * TILE + half * TERM + blk * 32;
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr + 16 * CHS));
        return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f4 acc[NP];
    auto step = [&](auto role_c, const char *cur) {
        constexpr int R = decltype(role_c)::value;
        constexpr int NM = m_hi(R) - m_lo(R) + 1;
        static_for<0, 2>([&](auto hc) {
            constexpr int hfi = decltype(hc)::value;             // channels 32 hfi .. 32 hfi + 31 of the step
            h8 a[NAB], b[2];
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) a[ab] = frag(cur, 0, hfi, a_blk(R, ab));
            b[0] = frag(cur, 1, hfi, m_lo(R));
            static_for<0, NM>([&](auto jc) {
                constexpr int jj = decltype(jc)::value, m = m_lo(R) + jj;
                if constexpr (jj + 1 < NM) b[(jj + 1) & 1] = frag(cur, 1, hfi, m + 1);
                static_for<0, NAB>([&](auto abc) {
                    constexpr int ab = decltype(abc)::value;
                    constexpr int pi = pair_idx(R, ab, m);
                    if constexpr (pi >= 0) acc[pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[jj & 1], a[ab], acc[pi], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);   // fragments one block ahead, not all of them (register budget)
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
    // accumulators -> LDS [plane = 4 ai + bi][ti + slack][x], as correlation_f16x2.hip
    auto scatter = [&](auto role_c) {
        constexpr int R = decltype(role_c)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e_ai = (ln & 15) >> 2, e_aj = ln & 3, e_bi = ln >> 4;
        const int rot = 4 * (4 * e_bi + e_ai);
        const int rbase = ((4 * e_ai + e_bi) * O_DP + O_SLACK + DR - 12 - e_aj) * O_RS;     // row of (dm = -3, r = 0)
        static_for<0, NAB>([&](auto abc) {
            constexpr int ab = decltype(abc)::value;
            constexpr int a = a_blk(R, ab);
            float *dst = Os + rbase + ((8 * a + 2 * e_aj + xpar + rot) & 63);
            static_for<0, 7>([&](auto dmc) {
                constexpr int dm = decltype(dmc)::value - 3;
                constexpr int pi = pair_idx(R, ab, a + dm);
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;   // r = bj; ti = 4 dm + r - e_aj + DR
                    float v = 0.0f;                          // B block outside the image: zeros
                    if constexpr (pi >= 0) v = acc[pi][r];
                    dst[(4 * (dm + 3) + r) * O_RS] = v;
                });
            });
        });
    };
    auto epilogue = [&](const Task &tk) {
        switch (role) {
        case 0: scatter(std::integral_constant<int, 0>{}); break;
        case 1: scatter(std::integral_constant<int, 1>{}); break;
        case 2: scatter(std::integral_constant<int, 2>{}); break;
        default: scatter(std::integral_constant<int, 3>{}); break;
        }
        __syncthreads();
        store_rows(tk);
        __syncthreads();
    };
    for (int it = 0; it < n_real; ++it) {
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        for (int s = 0; s < nsteps; s += 2) {
            step_dispatch(smem);
            __syncthreads();
            step_dispatch(smem + BUF);
            __syncthreads();
        }
        epilogue(get_task(it));
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = n_real; it < n_tasks; ++it) epilogue(get_task(it));   // zero-only tasks
}


// ---------------------------------------------------------------------------------------------------------------------
// Maps wider than 64 pixels: the column windows of correlation_f16x2_wide.hip (4 A' blocks = 32 pixels against the 10 B' blocks
// they meet, in the same 16 block slots per channel row; lanes re-mapped so that every load instruction has one source tensor)
// with this file's half staging and single product.  7 block pairs x 2 channel halves = 14 MFMAs per matrix wave and step.
constexpr int AW = 4, NB = 7, WPX = 8 * AW;
struct ArgsHW : ArgsH {
    int NXQ;                     // column windows: ceil(W / 32)
    unsigned magic_x;            // ceil(2^32 / NXQ)
};
struct TaskW { int n, py, rg, u, xq; };
struct LoadSetHW { u4 a[2], b0[2], b1[4]; };   // [channel half] of A' and of B' 0..3, [slot] of B' 4..11

__global__ __launch_bounds__(1024, 4) void corr_fwd_f16_wide(ArgsHW p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < 8;
    const int w8 = wave & 7;
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int nsteps = p.C / CKH;      // even (launcher)

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

    // rows of the window's 32 pixels: a lane owns 4 pixels (8 bytes of half output) of the rows ti = (lane >> 3) + 8 i
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    float *Os = reinterpret_cast<float *>(smem);
    auto store_rows = [&](const TaskW &tk) {
        const int pl = wave, ai = pl >> 2, bi = pl & 3;
        const int tj = 4 * tk.u + bi - ai, IL = 4 * tk.rg + ai;
        if (tj < 0 || tj >= D || IL >= HL) return;
        const int y = 2 * IL + tk.py;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 3, xg = 4 * (ln & 7), xw = WPX * tk.xq + xg;
        constexpr int NR = (D + 7) / 8;
        const float *src = Os + (pl * O_DP + O_SLACK + g) * O_RS + ((xg + 4 * (4 * bi + ai)) & 63);
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (long)tk.n * p.out_bs, 0, (unsigned)(D * D * HW * 2), 0x00020000);
        const unsigned vo = xw < p.W ? (unsigned)((g * HW + xw) * 2) : 0x80000000u;
        const int so0 = (int)((((long)tj * D) * p.H + y) * p.W * 2);
        f4 vals[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) vals[i] = *reinterpret_cast<const f4 *>(src + 8 * i * O_RS);
        float r, f = 1.0f, sl = 1.0f;
        asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(p.rC));
        if (!pow2) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "s"(p.fC));
        if (p.slope != 1.0f) asm volatile("v_mov_b32 %0, %1" : "=v"(sl) : "s"(p.slope));
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned v = (8 * i + 7 < D || g + 8 * i < D) ? vo : 0x80000000u;          // ti = g + 8 i < 21
            f4 val = vals[i];
            if (pow2) { val[0] *= r; val[1] *= r; val[2] *= r; val[3] *= r; }
            else { val[0] /= f; val[1] /= f; val[2] /= f; val[3] /= f; }
            if (p.slope != 1.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = val[e] > 0.0f ? val[e] : val[e] * sl;
            }
            const u2 packed = {pk_f16(val[0], val[1]), pk_f16(val[2], val[3])};
            __builtin_amdgcn_raw_buffer_store_b64(packed, rso, (int)v, so0 + 8 * i * (int)(HW * 2), 2);
        }
    };

    if (is_stage) {
        // loads a / b0 (A', B' 0..3): lane = (channel + 4, channel + 1, row, piece), channels b, b+4 | b+1, b+5 of channel half k2,
        // b = 8 (w >> 1) + 2 (w & 1); loads b1 (B' 4..11): the narrow kernel's mapping, slot k = channels 16k + 2w + (lane >> 5)
        const int a_piece = lane & 3, s_row = (lane >> 2) & 3;
        const int a_ch = 8 * (w8 >> 1) + 2 * (w8 & 1) + 4 * ((lane >> 4) & 1) + (lane >> 5);
        const int wa_ofs = a_ch * CHS + a_piece * 32 + s_row * 8;
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_ch = 2 * w8 + (lane >> 5);
        const int wb_ofs = TILE + s_ch * CHS + s_piece * 32 + s_row * 8;
        const unsigned nbytes = (unsigned)(p.C * HW * 2);
        __amdgpu_buffer_rsrc_t rs1, rs2;
        unsigned v_offa, v_offb0, v_offb1;
        auto set_ctx = [&](const TaskW &tk, bool valid) {
            const int ila = 4 * tk.rg + s_row, ilb = 4 * tk.rg - DR + 4 * tk.u + s_row;
            const int xa = WPX * tk.xq + 8 * a_piece, xb0 = WPX * tk.xq - 24 + 8 * a_piece, xb1 = WPX * tk.xq + 8 + 8 * s_piece;
            const bool okb = valid && ilb >= 0 && ilb < HL;
            v_offa = (valid && ila < HL && xa < p.W) ? (unsigned)((a_ch * HW + (long)(2 * ila + tk.py) * p.W + xa) * 2) : 0x80000000u;
            v_offb0 = (okb && xb0 >= 0 && xb0 < p.W) ? (unsigned)((a_ch * HW + (long)(2 * ilb + tk.py) * p.W + xb0) * 2) : 0x80000000u;
            v_offb1 = (okb && s_piece < 6 && xb1 < p.W) ? (unsigned)((s_ch * HW + (long)(2 * ilb + tk.py) * p.W + xb1) * 2) : 0x80000000u;
            rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
        };
        auto issue_loads = [&](LoadSetHW &L, int c0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int so = (int)((c0 + 32 * k2) * HW * 2);
                L.a[k2] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)v_offa, so, 0);
                L.b0[k2] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb0, so, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) L.b1[k] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb1, (int)((c0 + 16 * k) * HW * 2), 0);
        };
        auto perm_write = [&](const u4 &q, char *dst) {
            const u2 even = {__builtin_amdgcn_perm(q[1], q[0], 0x05040100u), __builtin_amdgcn_perm(q[3], q[2], 0x05040100u)};
            const u2 odd = {__builtin_amdgcn_perm(q[1], q[0], 0x07060302u), __builtin_amdgcn_perm(q[3], q[2], 0x07060302u)};
            *(FN2_LDS(u2) *)(dst) = even;
            *(FN2_LDS(u2) *)(dst + PARS) = odd;
        };
        auto stage_write = [&](const LoadSetHW &L, char *buf) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                perm_write(L.a[k2], buf + wa_ofs + k2 * TERM);
                __builtin_amdgcn_sched_barrier(0);
                perm_write(L.b0[k2], buf + wa_ofs + 4 * 32 + k2 * TERM);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                perm_write(L.b1[k], buf + wb_ofs + (k >> 1) * TERM + (k & 1) * 16 * CHS);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        LoadSetHW L0, L1;
        if (n_real > 0) {
            set_ctx(get_task(0), true);
            issue_loads(L0, 0);
            issue_loads(L1, CKH);
        }
        for (int it = 0; it < n_real; ++it) {
            const TaskW tk = get_task(it);
            const bool has_next = it + 1 < n_real;
            stage_write(L0, smem);
            __syncthreads();
            for (int s = 0; s + 2 < nsteps; s += 2) {
                issue_loads(L0, (s + 2) * CKH);
                stage_write(L1, smem + BUF);
                __syncthreads();
                issue_loads(L1, (s + 3) * CKH);
                stage_write(L0, smem);
                __syncthreads();
            }
            set_ctx(get_task(has_next ? it + 1 : it), has_next);
            issue_loads(L0, 0);
            stage_write(L1, smem + BUF);
            __syncthreads();
            issue_loads(L1, CKH);
            __syncthreads();
            __syncthreads();   // the epilogue image is complete
            store_rows(tk);
            __syncthreads();
        }
        for (int it = n_real; it < n_tasks; ++it) {   // zero-only tasks
            __syncthreads();
            store_rows(get_task(it));
            __syncthreads();
        }
        return;
    }

    __builtin_amdgcn_s_setprio(2);
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);   // the wave's A' block
    const int r_base = xpar * PARS + (4 * (lane >> 4) + ((lane & 15) >> 2)) * CHS + (lane & 3) * 8;
    auto frag = [&](const char *buf, int slot, int half) -> h8 {
        const char *ptr = buf + r_base + (slot >> 3) * TILE + half * TERM + (slot & 7) * 32;
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr + 16 * CHS));
        return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    f4 acc[NB];
    auto step = [&](auto role_c, const char *cur) {
        constexpr int R = decltype(role_c)::value;
        static_for<0, 2>([&](auto hc) {
            constexpr int hfi = decltype(hc)::value;
            const h8 a = frag(cur, R, hfi);
            h8 b[2];
            b[0] = frag(cur, AW + R, hfi);
            static_for<0, NB>([&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                if constexpr (jj + 1 < NB) b[(jj + 1) & 1] = frag(cur, AW + R + jj + 1, hfi);
                acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[jj & 1], a, acc[jj], 0, 0, 0);
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
    auto epilogue = [&](const TaskW &tk) {
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int e_ai = (ln & 15) >> 2, e_aj = ln & 3, e_bi = ln >> 4;
            const int rot = 4 * (4 * e_bi + e_ai);
            const int rbase = ((4 * e_ai + e_bi) * O_DP + O_SLACK + DR - 12 - e_aj) * O_RS;
            float *dst = Os + rbase + ((8 * role + 2 * e_aj + xpar + rot) & 63);
#pragma unroll
            for (int dmi = 0; dmi < NB; ++dmi)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(4 * dmi + r) * O_RS] = acc[dmi][r];
        }
        __syncthreads();
        store_rows(tk);
        __syncthreads();
    };
    for (int it = 0; it < n_real; ++it) {
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        for (int s = 0; s < nsteps; s += 2) {
            step_dispatch(smem);
            __syncthreads();
            step_dispatch(smem + BUF);
            __syncthreads();
        }
        epilogue(get_task(it));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = n_real; it < n_tasks; ++it) epilogue(get_task(it));
}

} // namespace hh

bool corr_f16_fwd_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    if (dtype != FN2_F16) return false;
    if (k != 1 || s1 != 1 || s2 != 2 || pad != md || md / 2 != hf::DR || (md & 1)) return false;
    if (C % (2 * hh::CKH) != 0 || (H & 1) || (W % 8) != 0) return false;
    if ((long)C * H * W * 2 >= 0x7fffffffL) return false;   // 32-bit buffer offsets per batch item
    if ((long)hf::D * hf::D * H * W * 2 >= 0x7fffffffL) return false;   // ... of the output (cf. corr_f16x2_applicable)
    return true;
}

// in1, in2, out: half tensors; out_bs in elements
int corr_forward_f16(const void *in1, const void *in2, void *out, long out_bs, float slope, int B, int C, int H, int W, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(out, 16) || (out_bs % 4) != 0) return FN2_EALIGN;
    hh::ArgsHW a;
    a.in1 = static_cast<const _Float16 *>(in1); a.in2 = static_cast<const _Float16 *>(in2); a.out = static_cast<_Float16 *>(out);
    a.out_bs = out_bs; a.slope = slope;
    a.fC = (float)C; a.rC = 1.0f / (float)C;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.dbg = nullptr;
    a.NXQ = W > 64 ? (W + hh::WPX - 1) / hh::WPX : 1;   // W > 64: column windows (corr_fwd_f16_wide)
    a.magic_x = (unsigned)((0x100000000ull + a.NXQ - 1) / a.NXQ);
    const long per_window = hf::build_task_table(a, B, H);
    if (per_window < 0) return (int)per_window;
    if ((long)B * (a.R_item > a.P_item ? a.R_item : a.P_item) * a.NXQ >= 65536) return FN2_EUNSUPPORTED;   // exact magic divisions
    const long ntasks = per_window * a.NXQ;
    if (ntasks == 0) return FN2_OK;
    const long per_stream = (ntasks + 7) / 8;
    const int G = per_stream < 32 ? (int)per_stream : 32;
    if (W > 64) hipLaunchKernelGGL(hh::corr_fwd_f16_wide, dim3(8u * G), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(hh::corr_fwd_f16, dim3(8u * G), dim3(1024), 0, s, static_cast<const hh::ArgsH &>(a));
    return launch_status();
}

} // namespace fn2
