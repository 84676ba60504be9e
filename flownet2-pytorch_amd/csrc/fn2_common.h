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
