// f16x2_common.h -- definitions shared by the f16x2 correlation forward kernels (correlation_f16x2.hip: one task per
// workgroup pass; correlation_f16x2_pair.hip: one centre tile against two neighbour row blocks): LDS image geometry, wave
// roles, the two-term f16 split, the fp32 fallback for out-of-range operands, the kernel-argument block.
#pragma once
#include <type_traits>

#include "corr_params.h"

namespace fn2 {
namespace hf {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define FN2_LDS(T) __attribute__((address_space(3))) T

constexpr int DR = 10, D = 2 * DR + 1, NU = 6;   // displacement radius (lattice), planes per axis, B row blocks per A row block
constexpr int CK = 32;                            // channels per step
constexpr int CHS = 288;                          // bytes per channel of one (tile, term, parity) plane: 8 column blocks x 4 rows x 8 B + 32
constexpr int PARS = CK * CHS;                    // 9216
constexpr int TERM = 2 * PARS;                    // 18432
constexpr int TILE = 2 * TERM;                    // 36864
constexpr int BUF = 2 * TILE;                     // 73728: A tile + B tile of one step
constexpr int LDS_BYTES = 2 * BUF;                // 147456: two steps
constexpr int O_RS = 64;                          // epilogue row stride (floats)
constexpr int O_SLACK = 5;                        // the entries of a block pair reach 5 displacement columns past either end of the
constexpr int O_DP = D + O_SLACK;                 // band: slack rows take them (no select in the scatter); plane stride 26 rows --
                                                  // the top slack rows of a plane are the bottom slack rows of the next one
static_assert((16 * O_DP + O_SLACK) * O_RS * 4 <= LDS_BYTES, "epilogue image must fit the operand buffers");

constexpr int MAX_TAB = 768;                      // tasks per batch item the kernel-argument table holds (H <= 512)

template <class T> struct ArgsT {
    const T *in1, *in2;
    T *out;
    long out_bs;     // elements between batch items of `out`
    float slope;     // fused LeakyReLU slope (1 = none)
    float fC, rC;    // (float)C and 1 / C: kernel arguments, i.e. SGPRs (there is no float SALU to derive them on)
    int B, C, H, W;  // H even, W % 8 == 0, W <= 64, C % 64 == 0
    int R_item, P_item;        // tasks per batch item whose B rows meet the image / lie entirely in the padding
    unsigned magic_r, magic_p; // ceil(2^32 / R_item), ceil(2^32 / P_item)
    unsigned long long *dbg;   // profiling variant 64 only: where the s_memtime stamps go (fn2_debug_set_buffer)
    unsigned tab[MAX_TAB / 2];     // 16-bit entries (rg << 4 | py << 3 | u): the R_item real, then the P_item zero-only tasks of an item
};
typedef ArgsT<float> Args;

// A task = (batch item n, y parity, row group rg of 4 lattice rows, B row block u).  "Real" tasks have B rows inside the
// image; the others only write zeros.  Tasks are numbered item-major, separately for the two kinds; the (py, rg, u) of the
// k-th task of a kind within an item comes from a table the launcher puts into the kernel arguments (all scalar work).
struct Task { int n, py, rg, u, real; };

template <class A> __device__ __forceinline__ Task decode_task(const A &p, bool real, int k)
{
    const unsigned per = real ? (unsigned)p.R_item : (unsigned)p.P_item;
    const unsigned n = __umulhi((unsigned)k, real ? p.magic_r : p.magic_p);   // k / per (exact for k < 2^16, checked by the launcher)
    const unsigned r = (unsigned)k - n * per;
    // dword loads with a wave-uniform index: scalar loads from the kernel-argument segment
    const unsigned i = __builtin_amdgcn_readfirstlane((real ? 0u : (unsigned)p.R_item) + r);
    const unsigned e = (p.tab[i >> 1] >> (16u * (i & 1u))) & 0xffffu;
    Task t;
    t.real = real ? 1 : 0;
    t.n = __builtin_amdgcn_readfirstlane((int)n);
    t.u = __builtin_amdgcn_readfirstlane((int)(e & 7u));
    t.py = __builtin_amdgcn_readfirstlane((int)((e >> 3) & 1u));
    t.rg = __builtin_amdgcn_readfirstlane((int)(e >> 4));
    return t;
}

// host: the task table of one batch item (those whose B rows 4rg - 10 + 4u .. +3 meet [0, HL) first) and the division
// constants; returns the number of tasks of the launch, or a negative FN2_E* code for shapes the table cannot describe
template <class A> inline long build_task_table(A &a, int B, int H)
{
    const int HL = H / 2, NRG = (HL + 3) / 4;
    if (2 * NRG * NU > MAX_TAB) return FN2_EUNSUPPORTED;
    int R = 0, P = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int py = 0; py < 2; ++py)
            for (int g = 0; g < NRG; ++g)
                for (int u = 0; u < NU; ++u) {
                    const int ib0 = 4 * g - DR + 4 * u;
                    const bool real = ib0 + 3 >= 0 && ib0 < HL;
                    if (real != (pass == 0)) continue;
                    const unsigned e = (unsigned)((g << 4) | (py << 3) | u), i = (unsigned)(R + P);
                    a.tab[i >> 1] = (i & 1u) ? (a.tab[i >> 1] | (e << 16)) : e;
                    if (real) ++R; else ++P;
                }
    a.R_item = R; a.P_item = P;
    a.magic_r = R ? (unsigned)((0x100000000ull + R - 1) / R) : 0u;
    a.magic_p = P ? (unsigned)((0x100000000ull + P - 1) / P) : 0u;
    if ((long)B * (R > P ? R : P) >= 65536) return FN2_EUNSUPPORTED;   // the magic-number division is exact below 2^16
    const long ntasks = (long)B * (R + P);
    if (ntasks > 0x3fffffffL) return FN2_EINVAL;
    return ntasks;
}

// wave roles: A column blocks of role r, and the B column blocks they meet
constexpr int NAB = 2;                            // A blocks per wave
__host__ __device__ constexpr int a_blk(int role, int ab) { return role == 0 ? (ab ? 3 : 0) : role == 1 ? (ab ? 2 : 1) : role == 2 ? (ab ? 7 : 4) : (ab ? 6 : 5); }
__host__ __device__ constexpr int m_lo(int role) { return role == 0 ? 0 : role == 1 ? 0 : role == 2 ? 1 : 2; }
__host__ __device__ constexpr int m_hi(int role) { return role == 0 ? 6 : role == 1 ? 5 : 7; }
__host__ __device__ constexpr bool meets(int a, int m) { return m >= 0 && m <= 7 && m - a <= 3 && a - m <= 3; }
__host__ __device__ constexpr int pair_idx(int role, int ab, int m)
{
    int idx = 0;
    for (int mm = m_lo(role); mm <= m_hi(role); ++mm)
        for (int b = 0; b < NAB; ++b) {
            if (mm == m && b == ab) return meets(a_blk(role, ab), m) ? idx : -1;
            if (meets(a_blk(role, b), mm)) ++idx;
        }
    return -1;
}
constexpr int NP = 11;
static_assert(pair_idx(0, 1, 6) == NP - 1 && pair_idx(1, 1, 5) == NP - 1 && pair_idx(2, 1, 7) == NP - 1 && pair_idx(3, 1, 7) == NP - 1 &&
              pair_idx(0, 0, 4) == -1, "11 pairs per role");

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

// x - (float)half: one instruction, exact
__device__ __forceinline__ float resid_lo(unsigned hp, float x)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
    return r;
}
__device__ __forceinline__ float resid_hi(unsigned hp, float x)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
    return r;
}
__device__ __forceinline__ unsigned pk_f16(float a, float b)   // v_cvt_pk_f16_f32, round to nearest even
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, h2));
}

// One output element as a plain fp32 fma chain over the channels (any finite input): used for outputs whose matrix-core
// result is non-finite, i.e. an operand did not fit an f16 (or really is inf/nan).
__device__ __forceinline__ float exact_corr(const Args &p, int n, int y, int x, int tj, int ti)
{
    const long HW = (long)p.H * p.W;
    const int y2 = y + 2 * (tj - DR), x2 = x + 2 * (ti - DR);
    const float *a = p.in1 + (long)n * p.C * HW + (long)y * p.W + x;
    const bool inside = y2 >= 0 && y2 < p.H && x2 >= 0 && x2 < p.W;
    const float *b = p.in2 + (long)n * p.C * HW + (inside ? (long)y2 * p.W + x2 : 0);
    float s = 0.0f;
    for (int c = 0; c < p.C; ++c) s = fmaf(a[c * HW], inside ? b[c * HW] : 0.0f, s);
    return s;
}

} // namespace hf
} // namespace fn2
