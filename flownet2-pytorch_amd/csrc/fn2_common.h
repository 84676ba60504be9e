// fn2_common.h -- shared device/host helpers for libflownet2_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/flownet2_hip.h"

#define FN2_WAVE 64

namespace fn2 {

typedef _Float16 half_t;

static inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FN2_OK : (int)e;
}

static inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

static inline size_t dtype_size(int dtype)
{
    switch (dtype) {
    case FN2_F32: return 4;
    case FN2_F16: return 2;
    case FN2_F64: return 8;
    default: return 0;
    }
}

// Store of a finished output element / vector (written once, read by a LATER kernel): agent scope (sc1) -- written through the
// XCD's L2 as it is issued.  The eight L2s are not coherent with each other, so everything a kernel wrote must be in memory when
// it ends; left dirty (a plain store) the outputs are flushed in one burst at the end of the kernel, which the next kernel of the
// stream waits for.  Measured inside bench.py's step (scripts/gpu_bench_ab.sh, same box, two rounds): ChannelNorm backward 11.4 ->
// 9.7 us, forward 8.0 -> 7.5, Resample2d forward 21.7 -> 21.4, step 0.1976 -> 0.1953 ms (non-temporal stores instead: 0.1948, but
// ChannelNorm forward 8.6 us); the correlation kernels' rows carry the same hint on their buffer stores (-5 % of the step).
template <class V> __device__ __forceinline__ void store_out(V *p, V v)
{
#if defined(FN2_ABL_OUTTEMPORAL)   // timing ablation: plain stores
    *p = v;
#elif defined(FN2_ABL_OUTNT)       // timing ablation: non-temporal stores
    __builtin_nontemporal_store(v, p);
#else
    // (s_nop: a store of more than 64 bits needs a wait state before its data registers may be overwritten; the compiler
    // provides it for its own instructions, not for inline assembly -- cf. correlation_f16x2.hip)
    if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (sizeof(V) == 4) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else *p = v;
#endif
}

// ChannelNorm's gradient of one element (channelnorm_kernel.cu:93):
// static_cast<float>(gO) * static_cast<float>(x) / (static_cast<float>(out) + 1e-9) -- float product, double divide, rounded to float.
__device__ __forceinline__ float chnorm_grad(float go, float x, float o)
{
    const float prod = go * x;
    return (float)((double)prod / ((double)o + 1e-9));
}

// XCD-aware remap of a 1-D block index: consecutive logical tiles land on the same XCD
// (hardware dispatches block b to XCD b % 8), so neighbouring tiles share one L2.
// Bijective for any grid size (cdna guide 5 "XCD swizzle must be bijective").
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk)
{
    const unsigned NX = 8;
    const unsigned q = nblk / NX, r = nblk % NX;
    const unsigned xcd = bid % NX, idx = bid / NX;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

} // namespace fn2
