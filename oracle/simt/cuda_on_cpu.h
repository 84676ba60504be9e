/*
 * cuda_on_cpu.h -- a minimal CPU SIMT shim (TEST INFRASTRUCTURE ONLY).
 *
 * Lets g++ compile the *device code* of the reference's .cu files unchanged
 * (the __global__ kernels and __device__ helpers, sliced at build time from
 * /root/reference by build_ref.sh -- never copied into this repository) and
 * execute it on the CPU, one fibre per CUDA thread, with CUDA's 32-lane warp
 * semantics.  The resulting oracle/_ref/libfn2_ref.so is "the reference itself
 * run here": it pins the restated oracle (oracle/fn2_oracle.c) and generates
 * tests/golden/.  It is not a CUDA compatibility layer for the product: nothing
 * under flownet2-pytorch_amd/ includes or links it.
 *
 * Emulated: threadIdx/blockIdx/blockDim/gridDim, __shared__ (block-shared
 * because all fibres of a block run on one OS thread), __syncthreads(),
 * __syncwarp(), __shfl_down_sync() for one 32-lane warp per block or several,
 * atomicAdd(float*), long4/make_long4, min/max, warpSize == 32.
 * Scheduling: all fibres of a block are advanced round-robin from one barrier
 * to the next, in thread-index order (so atomicAdd order is thread order).
 */
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <vector>

using std::max;
using std::min;

struct uint3_ { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct long4 { long x, y, z, w; };
static inline long4 make_long4(long x, long y, long z, long w) { long4 r = {x, y, z, w}; return r; }

namespace at { struct Half {}; }

static thread_local uint3_ threadIdx, blockIdx;
static thread_local dim3 blockDim, gridDim;
static const int warpSize = 32;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ thread_local

namespace simt {

struct Fiber {
    ucontext_t ctx;
    char *stack;
    bool done;
};

struct Block {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    int current;
    bool in_kernel;
    /* per-lane exchange slots for shuffles (8 bytes covers float/double) */
    std::vector<uint64_t> xchg;
    const std::function<void()> *body;
};

static thread_local Block *g_block = nullptr;

static inline void yield_to_scheduler()
{
    Block *b = g_block;
    if (!b || !b->in_kernel) {
        fprintf(stderr, "simt: barrier/shuffle outside a fibre launch\n");
        abort();
    }
    Fiber &f = b->fibers[b->current];
    swapcontext(&f.ctx, &b->sched);
}

static void fiber_entry()
{
    Block *b = g_block;
    (*b->body)();
    b->fibers[b->current].done = true;
    swapcontext(&b->fibers[b->current].ctx, &b->sched);
}

/* Runs body() once per CUDA thread of a grid x block launch.  body captures the
 * kernel arguments and calls the kernel function. */
static inline void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    const size_t STACK = 256 * 1024;
    const unsigned nthreads = block.x * block.y * block.z;
    Block blk;
    blk.fibers.resize(nthreads);
    blk.xchg.assign(nthreads, 0);
    blk.body = &body;
    for (unsigned t = 0; t < nthreads; ++t) blk.fibers[t].stack = (char *)malloc(STACK);
    g_block = &blk;
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                for (unsigned t = 0; t < nthreads; ++t) {
                    Fiber &f = blk.fibers[t];
                    f.done = false;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
                }
                blk.in_kernel = true;
                bool any = true;
                while (any) { /* one pass == one barrier phase */
                    any = false;
                    for (unsigned t = 0; t < nthreads; ++t) {
                        Fiber &f = blk.fibers[t];
                        if (f.done) continue;
                        blk.current = (int)t;
                        threadIdx.x = t % block.x;
                        threadIdx.y = (t / block.x) % block.y;
                        threadIdx.z = t / (block.x * block.y);
                        swapcontext(&blk.sched, &f.ctx);
                        if (!f.done) any = true;
                    }
                }
                blk.in_kernel = false;
            }
    for (unsigned t = 0; t < nthreads; ++t) free(blk.fibers[t].stack);
    g_block = nullptr;
}

} // namespace simt

static inline void __syncthreads() { simt::yield_to_scheduler(); }
static inline void __syncwarp(unsigned = 0xffffffffu) { simt::yield_to_scheduler(); }

/* CUDA semantics: lane i receives the value of lane i+delta of ITS warp; a lane
 * whose source would be >= warpSize keeps its own value. */
template <typename V>
static inline V __shfl_down_sync(unsigned, V val, unsigned delta)
{
    static_assert(sizeof(V) <= 8, "shuffle payload");
    simt::Block *b = simt::g_block;
    const int tid = b->current;
    uint64_t raw = 0;
    memcpy(&raw, &val, sizeof(V));
    b->xchg[tid] = raw;
    simt::yield_to_scheduler(); /* everyone has published */
    const int lane = tid % warpSize;
    V got = val;
    if (lane + (int)delta < warpSize && tid + (int)delta < (int)b->xchg.size()) {
        uint64_t r = b->xchg[tid + delta];
        memcpy(&got, &r, sizeof(V));
    }
    simt::yield_to_scheduler(); /* everyone has read before the next publish */
    return got;
}

static inline float atomicAdd(float *addr, float v)
{
    float old = *addr;
    *addr = old + v;
    return old;
}
