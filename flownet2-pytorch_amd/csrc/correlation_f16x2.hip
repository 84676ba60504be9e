// correlation_f16x2.hip -- FlowNetC cost volume (correlation forward) on the gfx950 f16 matrix cores with a
// two-term split of the fp32 operands that is done ONCE per staged value.
//
// Replaces reference kernels channels_first + correlation_forward (correlation_cuda_kernel.cu:46-70, :73-147) for
// FlowNetC's configuration (FlowNetC.py:28,31: kernel_size 1, stride1 1, stride2 2, pad_size == max_displacement == 20,
// fp32) on maps up to 64 pixels wide:
//     out[n, tj*21 + ti, y, x] = (1/C) * sum_c in1[n,c,y,x] * in2[n,c, y + 2(tj-10), x + 2(ti-10)]      (in2 = 0 outside)
//
// Numerics.  Every fp32 operand x is written as x = h + l with h = RNE_f16(x) and l = RNE_f16(x - h) (x - h is exact in
// fp32), and a product is formed as  ah*bh + ah*bl + al*bh  on v_mfma_f32_16x16x32_f16 with fp32 accumulation.  The
// representation error of an operand is max(2^-22 |x|, 2^-25), the dropped term al*bl is below 2^-22 |ab|: fp32-class
// results (measured against fp64 in tests/test_gpu_parity.py) from 3 matrix instructions per 16x16x32 block product
// instead of the 6 of the bf16x3 kernel, and none of its per-fragment split work.  |x| >= 65520 does not fit an f16: h
// becomes inf, every output that involves the value comes out non-finite, and the store loop recomputes exactly those
// outputs with a plain fp32 fma chain (exact_corr below) -- correct for any finite fp32 input, fast for |x| < 65520.
//
// Mapping.  stride2 = 2 keeps pixel parity, so each of the 4 parity classes is a dense lattice (I, J) = (y>>1, x>>1) on
// which the displacement window is the contiguous 21x21 box.  A 4x4 block of lattice pixels of in1 ("A block") against
// a 4x4 block of in2 ("B block") over 32 channels is one MFMA chain; both block grids are ALIGNED to 4 lattice columns
// (blocks are then whole 8-byte chunks of the LDS image, which is what the transposing LDS read needs), rows of B blocks
// start at 4*rg - 10 + 4*u.  An A column block a meets the B column blocks a-3 .. a+3 that exist (44 block pairs per
// parity and row-block pair instead of 48 on a grid offset by the radius).
//
// Work decomposition.  One workgroup (16 waves, one per CU) = one task (n, y parity, 4 lattice rows rg, B row block u):
// A tile and B tile of 4 lattice rows x 64 pixels.  Waves are specialised: waves 8-15 stage (global loads, operand split,
// LDS writes), waves 0-7 run the matrix cores: wave w takes x parity w&1 and the two A column blocks of role w>>1:
// {0,3}, {1,2}, {4,7}, {5,6} -- 11 block pairs each, 6 or 7 B fragments.  Each SIMD hosts two waves of each kind, so the
// split's VALU work and the MFMAs come from different instruction streams and overlap.
//
// Per step of 32 channels (one barrier per step, LDS double-buffered, two register sets for the loads):
//   - the loads of step s+2 are issued (8 x 16 B per lane; buffer loads: rows outside the image come back as zeros from the
//     range check, no select);
//   - the values of step s+1, loaded during step s-1, are split (cvt_pk / fma_mix / cvt_pk: 2 VALU per value) and written as
//     8-byte chunks [4 lattice columns of one parity] of the image [tile][term][parity][channel][column block][row] of
//     the other LDS buffer (staging waves) -- while
//   - the MFMAs of step s: every wave reads its operands with ds_read_b64_tr_b16 (the LDS transpose read: 4 channel rows
//     x 16 pixels -> per lane 4 channels of one pixel), two reads per operand, one address register and immediates for
//     all 36 reads; 33 MFMAs.
// Epilogue: accumulators -> LDS [plane (ai,bi)][ti][x] (64 floats per row, 16-byte slots rotated by 4 bi + ai: <= 2-way write
// conflicts, which a ds_write_b32 does not pay for) -> rows leave as 16 B per lane, 4 rows per instruction.
#include "f16x2_common.h"
#include "f16x2_split.h"

namespace fn2 {
namespace hf {
using f16s::exp_stat;
using f16s::scale_exp;
using f16s::split2;
using f16s::to_sgpr;
using f16s::wave_sum;

struct LoadSet { u4 a[2][2], b[2][2]; };   // one step of one lane: [slot][half] x 16 B of the A tile and of the B tile

// VAR: profiling switches (0 = the real kernel): 1 no MFMA, 2 no global loads, 4 no global stores, 8 no operand reads,
//      16 no split / LDS staging writes, 32 no epilogue (no scatter, no stores), 64 s_memtime stamps dumped over the output,
//      4096 every step loads channel chunk 0 (all loads hit the L2: measured no faster -- misses do not pace the steps)
//
// Persistent: the grid is 8 x G workgroups (G <= 32, one per CU); workgroup b belongs to stream b % 8 (= its XCD, so the
// tasks of one batch item share an L2) and walks a fixed list: every G-th real task of the stream's share, then its share
// of the zero-only tasks (handed to the workgroups with one real task fewer).  The staging waves run ahead: the loads of
// the next task's first two steps are issued during the last two steps of the current one, so only the first task of a
// workgroup sees the global-memory latency, and the output rows of a task drain while the next one is computed.
template <int VAR>
__global__ __launch_bounds__(1024, 4) void corr_fwd_f16x2(Args p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    // scale exponents (f16x2_split.h).  [0] = ka + kb of the CURRENT task: written by staging wave 0 before the task's first
    // barrier, read by the matrix waves after it (they undo the scales in the epilogue).  [1], [2] = ka, kb of the NEXT task:
    // written by staging wave 0 -- the only wave that samples after a workgroup's first task -- while the matrix waves scatter,
    // read by the other staging waves after the barrier that follows.
    __shared__ int scl_k[3];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < 8;    // waves 0-7 (dispatched first) load, split and fill the LDS buffers; waves 8-15 run the matrix cores
    const int w8 = wave & 7;
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int nsteps = p.C / CK;       // even
    constexpr bool PADS_FIRST = (VAR & 2048) != 0;   // zero-only tasks before the real ones

    // VAR 64 (profiling): s_memtime stamps of wave 0 and wave 8, dumped over the start of the output at the end
    unsigned long long ts[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ts[i] = 0;
    auto stamp = [&](int i) __attribute__((always_inline)) { if (VAR & 64) ts[i] = __builtin_amdgcn_s_memtime(); };
    auto dump = [&]() __attribute__((always_inline)) {
        if ((VAR & 64) && p.dbg && lane == 0 && (wave == 0 || wave == 8)) {   // slot 0: staging wave 0, slot 1: matrix wave 8
            unsigned long long *d = p.dbg + (blockIdx.x * 2 + (wave >> 3)) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = ts[i];
        }
    };
    stamp(0);

    // ---- this workgroup's task list
    const int G = gridDim.x >> 3, strm = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int Rtot = p.B * p.R_item, Ptot = p.B * p.P_item;
    const int r0 = (int)((long)strm * Rtot / 8), r1 = (int)((long)(strm + 1) * Rtot / 8);
    const int q0 = (int)((long)strm * Ptot / 8), q1 = (int)((long)(strm + 1) * Ptot / 8);
    const int Rc = r1 - r0, Pc = q1 - q0;
    const int n_real = (Rc - j + G - 1) / G > 0 ? (Rc - j + G - 1) / G : 0;
    const int rem = Rc % G;                                   // workgroups rem .. G-1 have one real task fewer
    const int pgrp = rem == 0 ? G : G - rem, pj = rem == 0 ? j : j - rem;
    const int n_pad = (pj >= 0 && Pc - pj > 0) ? (Pc - pj + pgrp - 1) / pgrp : 0;
    const int n_tasks = n_real + n_pad;
    auto get_task = [&](int i) -> Task {
        if (i < n_real) return decode_task(p, true, r0 + j + G * i);
        return decode_task(p, false, q0 + pj + pgrp * (i - n_real));
    };

    // ---- write-out of the epilogue image (all 16 waves).  The epilogue is bound by VALU issue like everything else in this
    // kernel (per SIMD the old version executed ~1.3 k VALU instructions per task: row index divisions, 64-bit addresses,
    // band selects), so the work is laid out to need almost none: wave w owns plane w = (ai, bi) -- its displacement row tj,
    // image row and validity are scalars --, a lane owns 16 bytes of the rows ti = (lane >> 4) + 4 i: LDS offsets are one
    // register + immediates, the global rows are a buffer store with one lane offset and scalar row offsets.
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int lgC = pow2 ? 31 - __builtin_clz((unsigned)p.C) : 0;
    float *Os = reinterpret_cast<float *>(smem);
    // ksum = ka + kb: the matrix-core sums carry 2^ksum (the operand scales of the task); removed with the 1/C, exactly
    auto store_rows = [&](const Task &tk, int ksum) {
        if (VAR & 32) return;
        const int pl = wave, ai = pl >> 2, bi = pl & 3;
        const int tj = 4 * tk.u + bi - ai, IL = 4 * tk.rg + ai;
        if (tj < 0 || tj >= D || IL >= HL) return;             // the whole plane lies outside the volume (uniform)
        const int y = 2 * IL + tk.py;
        int ln = lane;
        asm volatile("" : "+v"(ln));   // keeps the lane geometry from being hoisted out of the task loop (and spilled)
        const int g = ln >> 4, xg = 4 * (ln & 15);
        constexpr int NR = (D + 3) / 4;   // 6 rows per lane, the last one only for g == 0
        const float *src = Os + (pl * O_DP + O_SLACK + g) * O_RS + ((xg + 4 * (4 * bi + ai)) & 63);   // + 4 i rows: immediates
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (long)tk.n * p.out_bs, 0, (unsigned)(D * D * HW * 4), 0x00020000);
        const unsigned vo = xg < p.W ? (unsigned)((g * HW + xg) * 4) : 0x80000000u;   // out-of-range lanes store nothing
        const int so0 = (int)((((long)tj * D) * p.H + y) * p.W * 4);                    // row ti = 0 of this plane
        f4 vals[NR];
        // all LDS reads first, then the arithmetic and the stores: one LDS latency per task instead of one per row
#pragma unroll
        for (int i = 0; i < NR; ++i) vals[i] = *reinterpret_cast<const f4 *>(src + 4 * i * O_RS);
        // Scaling: v_ldexp_f32 by a scalar exponent -- 2^-ksum and, for a power-of-two C, the 1/C in one exact step.  C and the
        // slope are copied from the kernel arguments (SGPRs) into registers once per call, AFTER the LDS reads were issued and
        // BEFORE the first store.  (Kept in VGPRs across the task loop they get spilled, and a scratch reload waits for
        // vmcnt(0), i.e. for the acknowledgement of every row store before it.  Copied by an asm statement between the stores,
        // the copy can land in a data register of the 16-byte store just issued: the hardware needs a wait state before such a
        // register is overwritten, the compiler inserts it for its own instructions but not for inline assembly -- the last
        // lanes of the store then carry the copied value instead of their own.)
        float f = 1.0f, sl = 1.0f;
        if (!pow2) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "s"(p.fC));
        if (p.slope != 1.0f) asm volatile("v_mov_b32 %0, %1" : "=v"(sl) : "s"(p.slope));
        const int kx_mm = -ksum - lgC, kx_ex = -lgC;   // matrix-core sums / sums of the fp32 fallback
        auto finish = [&](f4 val, int kx) {
#pragma unroll
#ifdef FN2_ABL_MULEPI   // timing ablation: multiply by 2^kx (valid for -126 <= kx <= 127 only)
            for (int e = 0; e < 4; ++e) val[e] = val[e] * __builtin_bit_cast(float, (unsigned)(127 + kx) << 23);
#else
            for (int e = 0; e < 4; ++e) val[e] = __builtin_ldexpf(val[e], kx);
#endif
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
            const unsigned v = (4 * i + 3 < D || g == 0) ? vo : 0x80000000u;          // ti = g + 4 i < 21
            // inf / nan in the image: an operand did not fit an f16 (class mask: sNaN, qNaN, -inf, +inf)
            if ((VAR & 127) == 0 && (__builtin_amdgcn_classf(vals[i][0], 0x207) | __builtin_amdgcn_classf(vals[i][1], 0x207) |
                                     __builtin_amdgcn_classf(vals[i][2], 0x207) | __builtin_amdgcn_classf(vals[i][3], 0x207)))
                bad |= 1u << i;
            if (!(VAR & 4))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, finish(vals[i], kx_mm)), rso, (int)v, so0 + 4 * i * (int)(HW * 4),
#ifdef FN2_ABL_FWDTEMPORAL   // timing ablation: temporal row stores
                                                       0);
#else
                                                       (VAR & 1024) ? 0 : 2);
#endif
        }
        // An operand did not fit an f16 (or is inf/nan): a second pass recomputes exactly those outputs in fp32 and stores the
        // row again (kept out of the loop above: inlined there, its live state pushes the row values into scratch).
        if (bad) {
#pragma unroll 1
            for (int i = 0; i < NR; ++i) {
                const int ti = g + 4 * i;
                if (!(bad >> i & 1) || ti >= D || xg >= p.W) continue;
                // the finite entries take the matrix-core scale here, so that the whole row can be finished as fallback values
                f4 val = *reinterpret_cast<const f4 *>(src + 4 * i * O_RS);
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    const float cur = e == 0 ? val[0] : e == 1 ? val[1] : e == 2 ? val[2] : val[3];
                    const bool nonfin = (__builtin_bit_cast(unsigned, cur) & 0x7f800000u) == 0x7f800000u;
                    const float ex = nonfin ? exact_corr(p, tk.n, y, xg + e, tj, ti) : __builtin_ldexpf(cur, -ksum);
                    val[0] = e == 0 ? ex : val[0]; val[1] = e == 1 ? ex : val[1];
                    val[2] = e == 2 ? ex : val[2]; val[3] = e == 3 ? ex : val[3];
                }
                *reinterpret_cast<f4 *>(p.out + (long)tk.n * p.out_bs + ((long)(tj * D + ti) * p.H + y) * p.W + xg) = finish(val, kx_ex);
            }
        }
    };

    if (is_stage) {
        // ================= staging waves =================
        // A step has 32 channels x 4 rows x 8 pieces (8 pixels = 4 lattice columns of each parity) per tile; slot k (0, 1) of
        // a tile covers channels 16k .. 16k+15: staging wave w channels 16k + 2w, 16k + 2w + 1.  Lane = (channel, piece>>2,
        // row, piece&3): a 16-lane group then writes 16 distinct 8-byte slots of a 128-byte window.
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_row = (lane >> 2) & 3;
        const int s_ch = 2 * w8 + (lane >> 5);
        const int s_x = 8 * s_piece;
        const int w_ofs = s_ch * CHS + s_piece * 32 + s_row * 8;   // this lane's chunk inside a (tile, term, parity, slot) plane
        const unsigned nbytes = (unsigned)(p.C * HW * 4);
        // per-task load context: buffer descriptors of the batch item, per-lane offsets.  Buffer loads: an offset beyond
        // num_records returns 0 (the scalar offset is not part of the range check) -- rows outside the image need no select.
        __amdgpu_buffer_rsrc_t rs1, rs2;
        unsigned v_offa, v_offb;
        auto set_ctx = [&](const Task &tk, bool valid) {
            const int ib0 = 4 * tk.rg - DR + 4 * tk.u;
            const int s_ila = 4 * tk.rg + s_row, s_ilb = ib0 + s_row;
            const bool s_oka = valid && (s_ila < HL) && (s_x < p.W);
            const bool s_okb = valid && (s_ilb >= 0) && (s_ilb < HL) && (s_x < p.W);
            v_offa = s_oka ? (unsigned)((s_ch * HW + (long)(2 * s_ila + tk.py) * p.W + s_x) * 4) : 0x80000000u;
            v_offb = s_okb ? (unsigned)((s_ch * HW + (long)(2 * s_ilb + tk.py) * p.W + s_x) * 4) : 0x80000000u;
            rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
        };
        auto issue_loads = [&](LoadSet &L, int c0) {
            if (VAR & 2) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { L.a[k][h] = (u4)(0x3f800000u + lane); L.b[k][h] = (u4)(0x40000000u + lane); }
                return;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int soff = (int)((c0 + 16 * k) * HW * 4);
                L.a[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)v_offa, soff, 0);
                L.a[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(v_offa + 16), soff, 0);
                L.b[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)v_offb, soff, 0);
                L.b[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(v_offb + 16), soff, 0);
            }
        };
        // 8 consecutive pixels -> (hi, lo) x (parity 0, parity 1) chunks of 4 lattice columns, scaled by the tile's sc = 2^k
        auto split_write = [&](const u4 &q0, const u4 &q1, char *dst, f16s::scale2_t sc) {
            if (VAR & 16) {
                asm volatile("" ::"v"(q0), "v"(q1));
                return;
            }
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
        f16s::scale2_t sc_a = f16s::scale2_from_exp(0), sc_b = sc_a;   // the current task's operand scales (SGPR pairs)
        auto stage_write = [&](const LoadSet &L, char *buf) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {   // one item at a time: interleaving them costs registers this branch does not have
                split_write(L.a[k][0], L.a[k][1], buf + w_ofs + k * 16 * CHS, sc_a);
                __builtin_amdgcn_sched_barrier(0);
                split_write(L.b[k][0], L.b[k][1], buf + w_ofs + k * 16 * CHS + TILE, sc_b);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Operand sample of a task: ONE 16-byte load per lane and tile straight from global memory -- lane l: channel l C / 64, tile
        // row l & 3, 4 pixels at a pseudo-random column -- 256 values per tile.  For a workgroup's FIRST task every
        // staging wave loads the same 256 + 256 values and derives the same two exponents (no exchange, no barrier before the
        // first operands are converted); afterwards staging wave 0 alone samples -- two steps before the task starts, ahead of its
        // first operand loads -- and publishes the exponents through LDS while the matrix waves scatter the previous task's
        // accumulators.  (The vector-memory path paces the steps, and these loads touch 64 cache lines each: issued by all 8
        // waves they cost 0.8 us per launch, with four dword loads per lane and tile 1.6 us more.)
        struct Samp { u4 a, b; };
        auto sample_issue = [&](const Task &tk, Samp &S) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
            const int c = (ln * p.C) >> 6, r = ln & 3, x = 4 * (((((5 * ln) >> 2) & 15) * (p.W >> 2)) >> 4);
            const int ila = 4 * tk.rg + r, ilb = 4 * tk.rg - DR + 4 * tk.u + r;
            const unsigned oa = ila < HL ? (unsigned)((c * HW + (long)(2 * ila + tk.py) * p.W + x) * 4) : 0x80000000u;
            const unsigned ob = (ilb >= 0 && ilb < HL) ? (unsigned)((c * HW + (long)(2 * ilb + tk.py) * p.W + x) * 4) : 0x80000000u;
#ifdef FN2_ABL_NOSAMPLELOAD   // timing ablation
            S.a = (u4)0x3f000000u; S.b = (u4)0x3f000000u; (void)oa; (void)ob;
#else
            S.a = (VAR & 2) ? (u4)0x3f800000u : __builtin_amdgcn_raw_buffer_load_b128(r1, (int)oa, 0, 0);
            S.b = (VAR & 2) ? (u4)0x3f800000u : __builtin_amdgcn_raw_buffer_load_b128(r2, (int)ob, 0, 0);
#endif
        };
        auto sample_scales = [&](const Samp &S, int &ka, int &kb) {
            unsigned ta = 0u, tb = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ta += exp_stat(S.a[i]); tb += exp_stat(S.b[i]); }
            ka = scale_exp(wave_sum(ta));
            kb = scale_exp(wave_sum(tb));
        };

        // Invariant at the top of a real task: its steps 0 and 1 are in flight in L0 and L1.  During step s buffer s&1 is
        // being read, register set (s+1)&1 holds step s+1 and set s&1 is free: it receives step s+2 of this task, or -- in
        // the last two steps -- step s+2-nsteps of the next real task.
        // Channel order.  (Walking the channels backwards in every other task, to start with what is still in L2, measured no
        // gain -- the fabric reads already equal the algorithmic bytes -- and would make the summation order depend on the
        // task's place in the list; profiling switch 512 keeps the experiment.)
        auto chunk = [&](int it, int s) { return (VAR & 4096) ? 0 : ((VAR & 512) && (it & 1) ? nsteps - 1 - s : s) * CK; };
        LoadSet L0, L1;
        Samp SM;
        int ka_n = 0, kb_n = 0;                // the next task's scale exponents
        if (n_real > 0) {
            sample_issue(get_task(0), SM);
            set_ctx(get_task(0), true);
            issue_loads(L0, chunk(0, 0));
            issue_loads(L1, chunk(0, 1));
            sample_scales(SM, ka_n, kb_n);
        }
        stamp(1);
        auto zero_tasks = [&]() {
            for (int it = n_real; it < n_tasks; ++it) {
                __syncthreads();
                store_rows(get_task(it), 0);
                __syncthreads();
            }
        };
        if (PADS_FIRST) zero_tasks();
        for (int it = 0; it < n_real; ++it) {
            const Task tk = get_task(it);
            const bool has_next = it + 1 < n_real;
            const int ksum = ka_n + kb_n;
            sc_a = f16s::scale2_from_exp(ka_n); sc_b = f16s::scale2_from_exp(kb_n);
            if (tid == 0) scl_k[0] = ksum;     // (the matrix waves read the previous task's value right after that task's first barrier)
            stage_write(L0, smem);
            if (it < 2) stamp(2 + 6 * it);
            __syncthreads();
            for (int s = 0; s + 2 < nsteps; s += 2) {
                issue_loads(L0, chunk(it, s + 2));
                stage_write(L1, smem + BUF);
                __syncthreads();
                issue_loads(L1, chunk(it, s + 3));
                stage_write(L0, smem);
                __syncthreads();
            }
            // last two steps: the free register sets receive steps 0 and 1 of the next real task (after the last one the
            // offsets are out of range: the loads return zeros without touching memory)
            if (has_next && wave == 0) sample_issue(get_task(it + 1), SM);
            set_ctx(get_task(has_next ? it + 1 : it), has_next);
            issue_loads(L0, chunk(it + 1, 0));
            stage_write(L1, smem + BUF);
            __syncthreads();
            issue_loads(L1, chunk(it + 1, 1));
            __syncthreads();
            if (it < 2) stamp(3 + 6 * it);
            // while the matrix waves scatter their accumulators: the next task's scale exponents (wave 0; the others read them)
            if (has_next && wave == 0) {
                sample_scales(SM, ka_n, kb_n);
                if (lane == 0) { scl_k[1] = ka_n; scl_k[2] = kb_n; }
            }
            __syncthreads();   // the epilogue image is complete
            if (it < 2) stamp(4 + 6 * it);
            if (has_next && wave != 0) { ka_n = to_sgpr(scl_k[1]); kb_n = to_sgpr(scl_k[2]); }
            store_rows(tk, ksum);
            if (it < 2) stamp(5 + 6 * it);
            __syncthreads();   // ... and has been read: the buffers are free
            if (it < 2) stamp(6 + 6 * it);
        }
        if (!PADS_FIRST) zero_tasks();
        stamp(15);
        dump();
        return;
    }

    // ================= matrix-core waves =================
    // Transposing read: within a 16-lane group, lane 4j + c supplies the 8-byte chunk (channel row j, block row c); lane i
    // receives, for j = 0..3, element (i & 3) of the chunk of block row i >> 2 -- i.e. pixel (row i>>2, column i&3) of the
    // block for 4 channels.  Lane group g reads channels 4g + j and, in a second read, 16 + 4g + j: the 8 k-slots of lane
    // group g of a 16x16x32 operand.
    // the staging waves were dispatched first and would win every issue arbitration by age: give the matrix waves priority
    if (!(VAR & 256)) __builtin_amdgcn_s_setprio(2);
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);
    const int r_base = xpar * PARS + (4 * (lane >> 4) + ((lane & 15) >> 2)) * CHS + (lane & 3) * 8;
    auto frag = [&](const char *buf, int tile, int term, int blk) -> h8 {
        const char *ptr = buf + r_base + tile * TILE + term * TERM + blk * 32;
        if (VAR & 8) return (h8)((_Float16)1.0f);
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr + 16 * CHS));
        return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f4 acc[NP];
    // One step: D = (in2 block) x (in1 block): rows = B pixels (bi = lane>>4, bj = register), columns = A pixels (lane & 15).
    auto step = [&](auto role_c, const char *cur) {
        constexpr int R = decltype(role_c)::value;
        constexpr int NM = m_hi(R) - m_lo(R) + 1;
        // B fragments are fetched PF blocks ahead of their MFMAs
        constexpr int PF = (VAR & 128) ? 2 : 1;
        h8 ah[NAB], al[NAB], bh[PF + 1], bl[PF + 1];
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) { ah[ab] = frag(cur, 0, 0, a_blk(R, ab)); al[ab] = frag(cur, 0, 1, a_blk(R, ab)); }
#pragma unroll
        for (int i = 0; i < PF; ++i) { bh[i] = frag(cur, 1, 0, m_lo(R) + i); bl[i] = frag(cur, 1, 1, m_lo(R) + i); }
        static_for<0, NM>([&](auto jc) {
            constexpr int jj = decltype(jc)::value, m = m_lo(R) + jj;
            constexpr int cb = jj % (PF + 1), nb = (jj + PF) % (PF + 1);
            if constexpr (jj + PF < NM) { bh[nb] = frag(cur, 1, 0, m + PF); bl[nb] = frag(cur, 1, 1, m + PF); }
            if (VAR & 1) {
                asm volatile("" ::"v"(bh[cb]), "v"(bl[cb]));
            } else {
                // products bh*ah, bh*al, bl*ah; the A blocks alternate so that consecutive MFMAs use different accumulators
                static_for<0, 3>([&](auto prc) {
                    constexpr int pr = decltype(prc)::value;
                    static_for<0, NAB>([&](auto abc) {
                        constexpr int ab = decltype(abc)::value;
                        constexpr int pi = pair_idx(R, ab, m);
                        if constexpr (pi >= 0)
                            acc[pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? bl[cb] : bh[cb], pr == 1 ? al[ab] : ah[ab], acc[pi], 0, 0, 0);
                    });
                });
            }
        });
        if (VAR & 1) {
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) asm volatile("" ::"v"(ah[ab]), "v"(al[ab]));
        }
    };
    auto step_dispatch = [&](const char *cur) {
        switch (role) {
        case 0: step(std::integral_constant<int, 0>{}, cur); break;
        case 1: step(std::integral_constant<int, 1>{}, cur); break;
        case 2: step(std::integral_constant<int, 2>{}, cur); break;
        default: step(std::integral_constant<int, 3>{}, cur); break;
        }
    };
    // epilogue, first half: accumulators -> LDS [plane = 4 ai + bi][ti + 3][x], 16-byte slots rotated by 4 bi + ai.  Entries of
    // the outer block pairs that fall outside the 21-wide band land in the plane's slack rows (never read): every store is
    // one address register + an immediate, no selects.
    auto scatter = [&](auto role_c) {
        constexpr int R = decltype(role_c)::value;
        int ln = lane;   // opaque copy: see store_rows
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

    auto epilogue = [&](const Task &tk, int it, int ksum) {
        if (it < 2) stamp(3 + 6 * it);
        if (!(VAR & 32)) {
            switch (role) {
            case 0: scatter(std::integral_constant<int, 0>{}); break;
            case 1: scatter(std::integral_constant<int, 1>{}); break;
            case 2: scatter(std::integral_constant<int, 2>{}); break;
            default: scatter(std::integral_constant<int, 3>{}); break;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" ::"v"(acc[i]));
        }
        __syncthreads();
        if (it < 2) stamp(4 + 6 * it);
        store_rows(tk, ksum);
        if (it < 2) stamp(5 + 6 * it);
        __syncthreads();
        if (it < 2) stamp(6 + 6 * it);
    };
    if (PADS_FIRST) {
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = n_real; it < n_tasks; ++it) epilogue(get_task(it), 99, 0);
    }
    for (int it = 0; it < n_real; ++it) {
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        const int ksum = to_sgpr(scl_k[0]);    // ka + kb of this task's operand scales
        if (it < 2) stamp(2 + 6 * it);
        for (int s = 0; s < nsteps; s += 2) {
            step_dispatch(smem);
            __syncthreads();
            step_dispatch(smem + BUF);
            __syncthreads();
        }
        epilogue(get_task(it), it, ksum);
    }
    if (!PADS_FIRST) {
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = n_real; it < n_tasks; ++it) epilogue(get_task(it), 99, 0);   // zero-only tasks
    }
    stamp(15);
    dump();
}

} // namespace hf

#ifdef FN2_DEBUG_BUILD   // the debug library's timeline buffer (fn2_debug.h); the product library has no global state
static unsigned long long *g_debug_buffer = nullptr;
void corr_f16x2_set_debug_buffer(void *p) { g_debug_buffer = static_cast<unsigned long long *>(p); }
void *corr_f16x2_get_debug_buffer() { return g_debug_buffer; }
#endif

bool corr_f16x2_applicable(int dtype, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    if (dtype != FN2_F32) return false;
    if (k != 1 || s1 != 1 || s2 != 2 || pad != md || md / 2 != hf::DR || (md & 1)) return false;
    if (C % (2 * hf::CK) != 0 || C < 2 * hf::CK || (H & 1) || (W % 8) != 0) return false;   // W > 64: correlation_f16x2_wide.hip
    if ((long)C * H * W * 4 >= 0x7fffffffL) return false;   // 32-bit buffer offsets per batch item
    // ... of the output too: the epilogue forms (tj * D * H + y) * W * 4 and D * D * H * W * 4 in 32 bits (maps wider than 64 px
    // made this reachable: the task-table limit alone allows H * W up to ~1.6 M).  AUTO then goes on to the general kernel.
    if ((long)hf::D * hf::D * H * W * 4 >= 0x7fffffffL) return false;
    return true;
}

// variant: 0 = the kernel; other values = profiling switches (fn2_debug.h), only reachable through fn2_debug_*
int corr_forward_f16x2(const float *in1, const float *in2, float *out, long out_bs, float slope, int B, int C, int H, int W,
                       int variant, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(out, 16) || (out_bs % 4) != 0) return FN2_EALIGN;
    if (W > 64) return variant == 0 ? corr_forward_f16x2_wide(in1, in2, out, out_bs, slope, B, C, H, W, s) : FN2_EINVAL;
    hf::Args a;
    a.in1 = in1; a.in2 = in2; a.out = out; a.out_bs = out_bs; a.slope = slope;
    a.fC = (float)C; a.rC = 1.0f / (float)C;
    a.B = B; a.C = C; a.H = H; a.W = W;
#ifdef FN2_DEBUG_BUILD
    a.dbg = variant == 64 ? g_debug_buffer : nullptr;
#else
    a.dbg = nullptr;
#endif
    const long ntasks = hf::build_task_table(a, B, H);
    if (ntasks < 0) return (int)ntasks;
    if (ntasks == 0) return FN2_OK;
    // persistent grid: 8 streams (one per XCD) x G workgroups, one workgroup per CU
    const long per_stream = (ntasks + 7) / 8;
    const int G = per_stream < 32 ? (int)per_stream : 32;
#define FN2_HF(V) case V: hipLaunchKernelGGL((hf::corr_fwd_f16x2<V>), dim3(8u * G), dim3(1024), 0, s, a); return launch_status();
    switch (variant) {
        FN2_HF(0)
#ifdef FN2_DEBUG_BUILD   // profiling instantiations
        FN2_HF(1) FN2_HF(2) FN2_HF(4) FN2_HF(8) FN2_HF(16) FN2_HF(32) FN2_HF(6) FN2_HF(24) FN2_HF(25) FN2_HF(38) FN2_HF(64) FN2_HF(128) FN2_HF(256) FN2_HF(512) FN2_HF(1024) FN2_HF(1536) FN2_HF(2048) FN2_HF(2112) FN2_HF(4096)
#endif
    default: return FN2_EINVAL;
    }
#undef FN2_HF
}

} // namespace fn2
