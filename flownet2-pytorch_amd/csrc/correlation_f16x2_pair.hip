// correlation_f16x2_pair.hip -- the f16x2 correlation forward (correlation_f16x2.hip: numerics, LDS image, transposing
// operand reads, wave roles, epilogue) with ONE centre tile against TWO neighbour row blocks per workgroup.
//
// Why: the channel loop of correlation_f16x2.hip is paced by the CU's L2 -> register path (64 KB per 32-channel step at
// ~37 B/clk, scripts/ubench/lds_vmem_overlap.hip), under which the matrix work hides only partly.  A pair task
// (n, y parity, 4 centre rows rg, neighbour row blocks u = 2up and 2up+1) stages the centre tile once for two block products:
// 48 KB per product-step instead of 64, 25 % fewer split instructions and LDS writes -- and at FlowNetC's shape there are
// exactly 32 pair tasks with work per batch item: ONE per CU, one round, no second launch ramp.
//
// Per 32-channel step two half-steps, X = (A, B_u0) and Y = (A, B_u1), with two accumulator sets (88 registers) in the
// 8 matrix waves; the same four LDS tile buffers as the single-task kernel: A[step & 1], B[half-step & 1].  During
// half-step h the 4 staging waves write what comes next: the B tile of half-step h+1 and half of the A tile of the next
// step (6 items of 8 pixels per lane), from loads issued two half-steps earlier (two register sets of 12 x 16 B).
// One barrier per half-step.  12 waves per workgroup (3 per SIMD, 168 registers).  Epilogue: the two output tiles one
// after the other through the 86 KB LDS image (the rows of the first drain while the second is scattered).
#include "f16x2_common.h"

namespace fn2 {
namespace hf {

struct JobSet { u4 b[4][2], a[2][2]; };   // one lane's share of a B tile (4 slots of 8 channels) and of half an A tile (2 slots)

// pair task: (n, y parity, rg, up) -> neighbour row blocks u = 2up, 2up+1; `real`: at least one of them meets the image
struct PTask { int n, py, rg, up, real; };

__device__ __forceinline__ PTask decode_ptask(const Args &p, bool real, int k)
{
    const unsigned per = real ? (unsigned)p.R_item : (unsigned)p.P_item;
    const unsigned n = __umulhi((unsigned)k, real ? p.magic_r : p.magic_p);
    const unsigned r = (unsigned)k - n * per;
    const unsigned i = __builtin_amdgcn_readfirstlane((real ? 0u : (unsigned)p.R_item) + r);
    const unsigned e = (p.tab[i >> 1] >> (16u * (i & 1u))) & 0xffffu;
    PTask t;
    t.real = real ? 1 : 0;
    t.n = __builtin_amdgcn_readfirstlane((int)n);
    t.up = __builtin_amdgcn_readfirstlane((int)(e & 7u));
    t.py = __builtin_amdgcn_readfirstlane((int)((e >> 3) & 1u));
    t.rg = __builtin_amdgcn_readfirstlane((int)(e >> 4));
    return t;
}

// VAR: profiling switches (0 = the real kernel): 1 no MFMA, 2 no global loads, 4 no global stores, 8 no operand reads,
//      16 no split / LDS staging writes, 32 no epilogue
template <int VAR>
__global__ __launch_bounds__(768, 3) void corr_fwd_f16x2_pair(Args p)
{
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stage = wave < 4;    // waves 0-3 load, split and fill the LDS tiles; waves 4-11 run the matrix cores
    const int HL = p.H >> 1;
    const long HW = (long)p.H * p.W;
    const int nsteps = p.C / CK;
    const int nhalf = 2 * nsteps;      // half-steps per task

    // ---- this workgroup's task list (as in correlation_f16x2.hip)
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
    auto get_task = [&](int i) -> PTask {
        if (i < n_real) return decode_ptask(p, true, r0 + j + G * i);
        return decode_ptask(p, false, q0 + pj + pgrp * (i - n_real));
    };

    // ---- write-out of the epilogue image (all 12 waves): 336 rows (plane, ti) of 64 floats, 4 rows per wave instruction
    const float fC = (float)p.C;
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const float rC = 1.0f / fC;
    float *Os = reinterpret_cast<float *>(smem);
    auto store_rows = [&](const PTask &tk, int u) {
        if (VAR & 32) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));   // keeps the row geometry from being hoisted out of the task loop (and spilled)
        const int xg = 4 * (ln & 15);
        constexpr int NR = (16 * D + 47) / 48;   // rows per lane group: 7
        f4 vals[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = wave * 4 + (ln >> 4) + 48 * i;
            const int pl = row / D;
            const int rr = row < 16 * D ? row : 0;
            vals[i] = *reinterpret_cast<const f4 *>(Os + rr * O_RS + ((xg + 4 * (4 * (pl & 3) + (pl >> 2))) & 63));
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = wave * 4 + (ln >> 4) + 48 * i;
            const int pl = row / D, ti = row - pl * D;
            const int ai = pl >> 2, bi = pl & 3;
            const int tj = 4 * u + bi - ai;
            const int IL = 4 * tk.rg + ai;
            if (row >= 16 * D || tj < 0 || tj >= D || IL >= HL || xg >= p.W) continue;
            const int y = 2 * IL + tk.py;
            f4 val = vals[i];
            if ((VAR & 127) == 0) {
                const u4 bits = __builtin_bit_cast(u4, val);
                const bool bad = ((bits[0] & 0x7f800000u) == 0x7f800000u) | ((bits[1] & 0x7f800000u) == 0x7f800000u) |
                                 ((bits[2] & 0x7f800000u) == 0x7f800000u) | ((bits[3] & 0x7f800000u) == 0x7f800000u);
                if (bad) {   // an operand did not fit an f16 (or is inf/nan): recompute those outputs in fp32
#pragma unroll 1
                    for (int e = 0; e < 4; ++e) {
                        const float ex = exact_corr(p, tk.n, y, xg + e, tj, ti);
                        const unsigned be = e == 0 ? bits[0] : e == 1 ? bits[1] : e == 2 ? bits[2] : bits[3];
                        if ((be & 0x7f800000u) == 0x7f800000u) {
                            val[0] = e == 0 ? ex : val[0]; val[1] = e == 1 ? ex : val[1];
                            val[2] = e == 2 ? ex : val[2]; val[3] = e == 3 ? ex : val[3];
                        }
                    }
                }
            }
            if (pow2) val *= rC;
            else { val[0] /= fC; val[1] /= fC; val[2] /= fC; val[3] /= fC; }
            if (p.slope != 1.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = val[e] > 0.0f ? val[e] : val[e] * p.slope;
            }
            if (!(VAR & 4))
                __builtin_nontemporal_store(val, reinterpret_cast<f4 *>(p.out + (long)tk.n * p.out_bs + ((long)(tj * D + ti) * p.H + y) * p.W + xg));
        }
    };

    if (is_stage) {
        // ================= staging waves =================
        // Items of 8 pixels (4 lattice columns of each parity).  A slot covers 8 channels: staging wave w channels 2w, 2w+1;
        // lane = (channel, piece>>2, row, piece&3): a 16-lane group writes 16 distinct 8-byte slots of a 128-byte window.
        const int s_piece = (lane & 3) + 4 * ((lane >> 4) & 1);
        const int s_row = (lane >> 2) & 3;
        const int s_ch = 2 * wave + (lane >> 5);
        const int s_x = 8 * s_piece;
        const int w_ofs = s_ch * CHS + s_piece * 32 + s_row * 8;
        const unsigned nbytes = (unsigned)(p.C * HW * 4);

        for (int it = 0; it < n_tasks; ++it) {
            const PTask tk = get_task(it);
            if (tk.real) {
                // buffer loads: an offset beyond num_records returns 0 (the scalar offset is not part of the range check)
                const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in1 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in2 + (long)tk.n * p.C * HW), 0, nbytes, 0x00020000);
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int row = (ln >> 2) & 3;
                const int ila = 4 * tk.rg + row;
                const bool oka = ila < HL && s_x < p.W;
                const unsigned v_offa = oka ? (unsigned)((s_ch * HW + (long)(2 * ila + tk.py) * p.W + s_x) * 4) : 0x80000000u;
                unsigned v_offb[2];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int ilb = 4 * tk.rg - DR + 4 * (2 * tk.up + w) + row;
                    const bool okb = ilb >= 0 && ilb < HL && s_x < p.W;
                    v_offb[w] = okb ? (unsigned)((s_ch * HW + (long)(2 * ilb + tk.py) * p.W + s_x) * 4) : 0x80000000u;
                }
                // job g (-2 .. nhalf-1): what is written during half-step g: B tile g+1 (step (g+1)>>1, neighbour block (g+1)&1)
                // and A half g+2 (step (g+2)>>1, channels 16 ((g+2)&1) ..)
                auto issue_job = [&](JobSet &J, int g) {
                    const int b = g + 1, a = g + 2;
                    const bool hb = b >= 0 && b < nhalf, ha = a < nhalf;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((VAR & 2) || !hb) { J.b[k][0] = (u4)(0x40000000u + lane); J.b[k][1] = J.b[k][0]; continue; }
                        const int soff = (int)((CK * (b >> 1) + 8 * k) * HW * 4);
                        const unsigned vo = (b & 1) ? v_offb[1] : v_offb[0];
                        J.b[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)vo, soff, 0);
                        J.b[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(vo + 16), soff, 0);
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if ((VAR & 2) || !ha) { J.a[k][0] = (u4)(0x3f800000u + lane); J.a[k][1] = J.a[k][0]; continue; }
                        const int soff = (int)((CK * (a >> 1) + 16 * (a & 1) + 8 * k) * HW * 4);
                        J.a[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)v_offa, soff, 0);
                        J.a[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(v_offa + 16), soff, 0);
                    }
                };
                auto split_write = [&](const u4 &q0v, const u4 &q1v, char *dst) {
                    if (VAR & 16) { asm volatile("" ::"v"(q0v), "v"(q1v)); return; }
                    const f4 x0 = __builtin_bit_cast(f4, q0v), x1 = __builtin_bit_cast(f4, q1v);
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        const float e0 = x0[par], e1 = x0[2 + par], e2 = x1[par], e3 = x1[2 + par];
                        const unsigned h01 = pk_f16(e0, e1), h23 = pk_f16(e2, e3);
                        const unsigned l01 = pk_f16(resid_lo(h01, e0), resid_hi(h01, e1));
                        const unsigned l23 = pk_f16(resid_lo(h23, e2), resid_hi(h23, e3));
                        *(FN2_LDS(u2) *)(dst + par * PARS) = (u2){h01, h23};
                        *(FN2_LDS(u2) *)(dst + TERM + par * PARS) = (u2){l01, l23};
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one item at a time: interleaving them costs registers
                };
                auto write_job = [&](const JobSet &J, int g) {
                    const int b = g + 1, a = g + 2;
                    if (b >= 0 && b < nhalf) {
                        char *dst = smem + (b & 1) * BUF + TILE + w_ofs;
#pragma unroll
                        for (int k = 0; k < 4; ++k) split_write(J.b[k][0], J.b[k][1], dst + 8 * k * CHS);
                    }
                    if (a < nhalf) {
                        char *dst = smem + ((a >> 1) & 1) * BUF + w_ofs + 16 * (a & 1) * CHS;
#pragma unroll
                        for (int k = 0; k < 2; ++k) split_write(J.a[k][0], J.a[k][1], dst + 8 * k * CHS);
                    }
                };
                JobSet J0, J1;
                issue_job(J0, -2);
                issue_job(J1, -1);
                write_job(J0, -2);
                issue_job(J0, 0);
                write_job(J1, -1);
                issue_job(J1, 1);
                __syncthreads();                 // A(0) and B_u0(0) complete: half-step 0 may start
                for (int g = 0; g < nhalf; g += 2) {
                    write_job(J0, g);
                    issue_job(J0, g + 2);
                    __syncthreads();
                    write_job(J1, g + 1);
                    issue_job(J1, g + 3);
                    __syncthreads();
                }
            }
#pragma unroll 1
            for (int w = 0; w < 2; ++w) {
                __syncthreads();   // the epilogue image of neighbour block w is complete
                store_rows(tk, 2 * tk.up + w);
                __syncthreads();   // ... and has been read
            }
        }
        return;
    }

    // ================= matrix-core waves =================
    __builtin_amdgcn_s_setprio(2);
    const int w8 = wave - 4;
    const int xpar = w8 & 1;
    const int role = __builtin_amdgcn_readfirstlane(w8 >> 1);
    const int r_base = xpar * PARS + (4 * (lane >> 4) + ((lane & 15) >> 2)) * CHS + (lane & 3) * 8;
    auto frag = [&](const char *tile, int term, int blk) -> h8 {
        const char *ptr = tile + r_base + term * TERM + blk * 32;
        if (VAR & 8) return (h8)((_Float16)1.0f);
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FN2_LDS(s4) *)(ptr + 16 * CHS));
        return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f4 acc[2][NP];
    // one half-step: D = (in2 block) x (in1 block): rows = B pixels (bi = lane>>4, bj = register), columns = A pixels
    auto half_step = [&](auto role_c, auto wc, const char *ta, const char *tb) {
        constexpr int R = decltype(role_c)::value;
        constexpr int w = decltype(wc)::value;
        constexpr int NM = m_hi(R) - m_lo(R) + 1;
        h8 ah[NAB], al[NAB], bh[2], bl[2];
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) { ah[ab] = frag(ta, 0, a_blk(R, ab)); al[ab] = frag(ta, 1, a_blk(R, ab)); }
        bh[0] = frag(tb, 0, m_lo(R)); bl[0] = frag(tb, 1, m_lo(R));
        static_for<0, NM>([&](auto jc) {
            constexpr int jj = decltype(jc)::value, m = m_lo(R) + jj;
            constexpr int cb = jj & 1, nb = cb ^ 1;
            if constexpr (jj + 1 < NM) { bh[nb] = frag(tb, 0, m + 1); bl[nb] = frag(tb, 1, m + 1); }
            if (VAR & 1) {
                asm volatile("" ::"v"(bh[cb]), "v"(bl[cb]));
            } else {
                static_for<0, 3>([&](auto prc) {
                    constexpr int pr = decltype(prc)::value;
                    static_for<0, NAB>([&](auto abc) {
                        constexpr int ab = decltype(abc)::value;
                        constexpr int pi = pair_idx(R, ab, m);
                        if constexpr (pi >= 0)
                            acc[w][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? bl[cb] : bh[cb], pr == 1 ? al[ab] : ah[ab], acc[w][pi], 0, 0, 0);
                    });
                });
            }
        });
        if (VAR & 1) {
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) asm volatile("" ::"v"(ah[ab]), "v"(al[ab]));
        }
    };
    auto half_step_d = [&](auto wc, const char *ta, const char *tb) {
        switch (role) {
        case 0: half_step(std::integral_constant<int, 0>{}, wc, ta, tb); break;
        case 1: half_step(std::integral_constant<int, 1>{}, wc, ta, tb); break;
        case 2: half_step(std::integral_constant<int, 2>{}, wc, ta, tb); break;
        default: half_step(std::integral_constant<int, 3>{}, wc, ta, tb); break;
        }
    };
    // epilogue, first half: accumulators -> LDS [plane = 4 ai + bi][ti][x], 16-byte slots rotated by 4 bi + ai
    auto scatter = [&](auto role_c, auto wc) {
        constexpr int R = decltype(role_c)::value;
        constexpr int w = decltype(wc)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e_ai = (ln & 15) >> 2, e_aj = ln & 3, e_bi = ln >> 4;
        const int prow = (4 * e_ai + e_bi) * D;
        const int rot = 4 * (4 * e_bi + e_ai);
        static_for<0, NAB>([&](auto abc) {
            constexpr int ab = decltype(abc)::value;
            constexpr int a = a_blk(R, ab);
            const int xs = (8 * a + 2 * e_aj + xpar + rot) & 63;
            static_for<0, 7>([&](auto dmc) {
                constexpr int dm = decltype(dmc)::value - 3;
                constexpr int pi = pair_idx(R, ab, a + dm);
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;   // r = bj
                    const int ti = 4 * dm + r - e_aj + DR;
                    float v = 0.0f;                          // B block outside the image: zeros
                    if constexpr (pi >= 0) v = acc[w][pi][r];
                    if constexpr (dm >= -1 && dm <= 1) {
                        Os[(prow + ti) * O_RS + xs] = v;
                    } else {
                        const bool ok = (ti >= 0) && (ti < D);
                        Os[ok ? (prow + ti) * O_RS + xs : O_DUMMY + ln] = v;
                    }
                });
            });
        });
    };
    auto scatter_d = [&](auto wc) {
        if (VAR & 32) {
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" ::"v"(acc[decltype(wc)::value][i]));
            return;
        }
        switch (role) {
        case 0: scatter(std::integral_constant<int, 0>{}, wc); break;
        case 1: scatter(std::integral_constant<int, 1>{}, wc); break;
        case 2: scatter(std::integral_constant<int, 2>{}, wc); break;
        default: scatter(std::integral_constant<int, 3>{}, wc); break;
        }
    };

    for (int it = 0; it < n_tasks; ++it) {
        const PTask tk = get_task(it);
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[w][i] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        if (tk.real) {
            __syncthreads();
            for (int g = 0; g < nhalf; g += 2) {
                const char *ta = smem + ((g >> 1) & 1) * BUF;
                half_step_d(std::integral_constant<int, 0>{}, ta, smem + TILE);
                __syncthreads();
                half_step_d(std::integral_constant<int, 1>{}, ta, smem + BUF + TILE);
                __syncthreads();
            }
        }
        scatter_d(std::integral_constant<int, 0>{});
        __syncthreads();
        store_rows(tk, 2 * tk.up);
        __syncthreads();
        scatter_d(std::integral_constant<int, 1>{});
        __syncthreads();
        store_rows(tk, 2 * tk.up + 1);
        __syncthreads();
    }
}

} // namespace hf

// variant: 0 = the kernel; other values = profiling switches
int corr_forward_f16x2_pair(const float *in1, const float *in2, float *out, long out_bs, float slope, int B, int C, int H, int W,
                            int variant, hipStream_t s)
{
    if (!aligned(in1, 16) || !aligned(in2, 16) || !aligned(out, 16) || (out_bs % 4) != 0) return FN2_EALIGN;
    hf::Args a;
    a.in1 = in1; a.in2 = in2; a.out = out; a.out_bs = out_bs; a.slope = slope;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.dbg = nullptr;
    const int HL = H / 2, NRG = (HL + 3) / 4;
    if (NRG * hf::NU > hf::MAX_TAB) return FN2_EUNSUPPORTED;
    // table of the (py, rg, up) pair tasks of one batch item: those with a neighbour row block that meets [0, HL) first
    int R = 0, P = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int py = 0; py < 2; ++py)
            for (int g = 0; g < NRG; ++g)
                for (int up = 0; up < hf::NU / 2; ++up) {
                    bool real = false;
                    for (int w = 0; w < 2; ++w) {
                        const int ib0 = 4 * g - hf::DR + 4 * (2 * up + w);
                        real = real || (ib0 + 3 >= 0 && ib0 < HL);
                    }
                    if (real != (pass == 0)) continue;
                    const unsigned e = (unsigned)((g << 4) | (py << 3) | up), i = (unsigned)(R + P);
                    a.tab[i >> 1] = (i & 1u) ? (a.tab[i >> 1] | (e << 16)) : e;
                    if (real) ++R; else ++P;
                }
    a.R_item = R; a.P_item = P;
    a.magic_r = R ? (unsigned)((0x100000000ull + R - 1) / R) : 0u;
    a.magic_p = P ? (unsigned)((0x100000000ull + P - 1) / P) : 0u;
    if ((long)B * (R > P ? R : P) >= 65536) return FN2_EUNSUPPORTED;
    const long ntasks = (long)B * (R + P);
    if (ntasks == 0) return FN2_OK;
    const long per_stream = (ntasks + 7) / 8;
    const int G = per_stream < 32 ? (int)per_stream : 32;
#define FN2_HP(V) case V: hipLaunchKernelGGL((hf::corr_fwd_f16x2_pair<V>), dim3(8u * G), dim3(768), 0, s, a); return launch_status();
    switch (variant) {
        FN2_HP(0) FN2_HP(1) FN2_HP(2) FN2_HP(4) FN2_HP(8) FN2_HP(16) FN2_HP(32) FN2_HP(6) FN2_HP(38)
    default: return FN2_EINVAL;
    }
#undef FN2_HP
}

} // namespace fn2
