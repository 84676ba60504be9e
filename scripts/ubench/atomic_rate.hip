// Probe (profiling aid, round 5): what bounds global_atomic_add_f32 on gfx950 -- lane-operations or memory requests, and of what size?
// flush_probe.hip: 13.5 M row-contiguous fp32 atomics take 37 us (365 G/s) however the images are placed over the XCDs and with or
// without sc1; scattered ones (one per 64-B line) were measured at 29 G/s in round 2.  365 / 29 = 12.6 ~ 16 dwords per 64 bytes.
// Here: every lane issues REPS atomics (no return value); a wave instruction's 64 lanes are laid out as groups of G consecutive
// floats, each group in a segment of its own (SEG bytes apart, start OFF floats into the segment):
//     G = 64, 32, 16, 8, 4, 2, 1;  OFF = 0 or G / 2 (a group that straddles the boundary of its natural alignment)
// Total footprint 64 MB (the Infinity Cache holds it), addresses of successive instructions advance by a large odd stride.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr size_t NFLOAT = (size_t)16 << 20;   // 64 MB
constexpr int REPS = 64;

template <int KIND>   // 0: add_f32, 1: add_f64 (G counts doubles), 2: pk_add_f16 (G counts dwords), 3: plain store of a float (reference)
__global__ __launch_bounds__(1024) void k(float *base, int G, int off, int segfloats)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int grp = lane / G, in = lane % G;
    const int groups = 64 / G;
    for (int r = 0; r < REPS; ++r) {
        // instruction index -> a block of `groups` segments; successive instructions of the whole grid never share a segment
        const size_t inst = wave * REPS + r;
        size_t seg = (inst * 40503u) % (NFLOAT / segfloats / groups) * groups + grp;
        size_t idx = seg * segfloats + off + in;
        if (KIND == 1) idx = (seg * segfloats + 2 * (off + in)) % NFLOAT;
        idx %= NFLOAT - 2;
        if (KIND == 0) unsafeAtomicAdd(base + idx, 1.0f);
        else if (KIND == 1) unsafeAtomicAdd(reinterpret_cast<double *>(base) + idx / 2, 1.0);
        else if (KIND == 2) asm volatile("global_atomic_pk_add_f16 %0, %1, off" ::"v"(base + idx), "v"(0x3c003c00u) : "memory");
        else base[idx] = 1.0f;
    }
}

template <int KIND> static void run(float *buf, const char *name, int G, int off, int segfloats)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 1024;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(1024), 0, 0, buf, G, off, segfloats);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double lanes = (double)blocks * 1024 * REPS, groups = lanes / G;
    printf("%-14s G = %2d lanes per group, offset %2d, segment %4d B: %7.1f us  %7.1f G lane-ops/s  %6.1f G groups/s\n", name, G, off, segfloats * 4,
           best * 1e3, lanes / (best * 1e-3) / 1e9, groups / (best * 1e-3) / 1e9);
}

int main()
{
    float *buf; (void)hipMalloc(&buf, NFLOAT * 4); (void)hipMemset(buf, 0, NFLOAT * 4);
    for (int G : {64, 32, 16, 8, 4, 2, 1}) {
        const int seg = G * 4 < 64 ? 16 : G;                    // segment = natural alignment of the group, at least 64 B
        run<0>(buf, "add_f32", G, 0, seg < 64 ? seg * 4 : seg); // groups spread apart (4 segments of slack), aligned
        if (G > 1) run<0>(buf, "add_f32", G, G / 2, seg < 64 ? seg * 4 : seg * 2);   // straddling its natural boundary
    }
    // groups of 16 / 32 floats at 64-B and 128-B alignments: which boundary costs a second request?
    run<0>(buf, "add_f32", 16, 0, 64); run<0>(buf, "add_f32", 16, 16, 64); run<0>(buf, "add_f32", 32, 0, 64); run<0>(buf, "add_f32", 32, 16, 64);
    for (int G : {32, 16, 8, 1}) run<1>(buf, "add_f64", G, 0, G * 2 < 16 ? 64 : G * 4);
    for (int G : {64, 16, 1}) run<2>(buf, "pk_add_f16", G, 0, G < 16 ? 64 : G * 2);
    for (int G : {64, 16, 1}) run<3>(buf, "store (ref)", G, 0, G < 16 ? 64 : G * 2);
    return 0;
}
