// Probe (profiling aid, round 5): how should the Resample2d backward's accumulation windows reach grad_input1?
// The kernel's flush is 768 tiles x (64 x 96 window) x 3 channels of fp32 adds onto an 8 x 3 x 384 x 512 tensor, as contiguous rows.
// Questions:
//   1. where does workgroup b run (HW_REG_XCC_ID vs b % 8)?
//   2. global_atomic_add_f32 as hipcc emits it for workgroup AND agent scope (no sc bits: the two scopes have ONE encoding on gfx950)
//      -- rate when an XCD owns its image (tiles through xcd_remap) against tiles in raster order (every image shared by all XCDs);
//      the same with sc1
//   3. a lock per 16 x 64 granule of the image + plain 16-byte read-modify-write (sc1 loads: L2-served, past the CU's L1; plain
//      stores: kept in the XCD's L2) -- legal only while every workgroup that touches an image runs on ONE XCD; first touch of a
//      granule stores instead of adding (no zero fill of the tensor)
//   4. the far-pixel path under that protocol: a wave takes the granule locks of one pixel's four corners and adds 12 values
// Every variant is checked against the exact count of windows covering each cell (sums of 1.0f are exact).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics flush_probe.hip -o flush_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int B = 8, C = 3, H = 384, W = 512, TH = 32, TW = 64, R = 16, WH = TH + 2 * R, WW = TW + 2 * R, NT = 1024;
constexpr int GH = 16, GW = 64, GR = H / GH, GC = W / GW;   // lock granules
constexpr int TILES_X = W / TW, TILES_Y = H / TH, NTILES = B * TILES_X * TILES_Y;

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk)
{
    const unsigned NX = 8, q = nblk / NX, r = nblk % NX, xcd = bid % NX, idx = bid / NX;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ int xcc_id()
{
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}
__device__ __forceinline__ int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

__global__ void where_kernel(int *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// MODE 0: atomics, 1: atomics sc1, 2: granule locks + RMW onto a zero-filled tensor, 3: granule locks, first touch stores
template <int MODE, bool REMAP>
__global__ __launch_bounds__(NT, 8) void flush_kernel(float *__restrict__ G, unsigned *__restrict__ locks, const int *__restrict__ offs, int *__restrict__ bad)
{
    __shared__ float pad[18 * 1024];   // 72 KB: two workgroups per CU, as the real kernel
    const int tid = threadIdx.x;
    pad[tid] = (float)tid;
    int t = REMAP ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tile = t;
    const int tx = t % TILES_X; t /= TILES_X;
    const int ty = t % TILES_Y;
    const int b = t / TILES_Y;
    const int wx0 = tx * TW - R + offs[2 * tile], wy0 = ty * TH - R + offs[2 * tile + 1];
    const long HW = (long)H * W;
    if (MODE >= 2 && tid == 0 && xcc_id() != (REMAP ? (int)(blockIdx.x % 8) : -1)) atomicAdd(bad, 1);
    if constexpr (MODE < 2) {
        int ly = tid / WW, lx = tid - ly * WW;
        for (int i = tid; i < WH * WW; i += NT) {
            const int gx = wx0 + lx, gy = wy0 + ly;
            if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
                float *p = G + (long)b * C * HW + gy * W + gx;
                const float v = 1.0f + 0.0f * pad[(tid + i) & 1023];
                if (MODE == 0) {
                    unsafeAtomicAdd(p, v); unsafeAtomicAdd(p + HW, v); unsafeAtomicAdd(p + 2 * HW, v);
                } else {
                    asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                    asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p + HW), "v"(v) : "memory");
                    asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p + 2 * HW), "v"(v) : "memory");
                }
            }
            ly += NT / WW; lx += NT % WW;
            if (lx >= WW) { lx -= WW; ++ly; }
        }
    } else {
        const int wave = tid >> 6, lane = tid & 63;
        // granules the window meets, and the two granules of the tile's own core (every granule has ONE tile responsible for
        // initialising it even if no window reaches it)
        int g0 = fdiv(wy0, GH), g1 = fdiv(wy0 + WH - 1, GH), h0 = fdiv(wx0, GW), h1 = fdiv(wx0 + WW - 1, GW);
        g0 = min(g0, ty * (TH / GH)); g1 = max(g1, ty * (TH / GH) + TH / GH - 1); h0 = min(h0, tx); h1 = max(h1, tx);
        g0 = max(g0, 0); g1 = min(g1, GR - 1); h0 = max(h0, 0); h1 = min(h1, GC - 1);
        const int ncol = h1 - h0 + 1, nitem = (g1 - g0 + 1) * ncol;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(G + (long)b * C * HW, 0, (unsigned)(C * HW * 4), 0x00020000);
        for (int item = wave; item < nitem; item += NT / 64) {
            const int gr = g0 + item / ncol, gc = h0 + item % ncol;
            unsigned *lock = locks + (b * GR + gr) * GC + gc;
            unsigned old = 0;
            if (lane == 0) {
                while ((old = atomicOr(lock, 1u)) & 1u) __builtin_amdgcn_s_sleep(2);
            }
            old = __builtin_amdgcn_readfirstlane(old);
            const bool first = (MODE == 3) && !(old & 2u);
            const int r4 = lane >> 4, c4 = lane & 15;
            const int x = gc * GW + 4 * c4;
            const bool inx = (x >= wx0) && (x < wx0 + WW);
            f4 v[4][C];
            bool in[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int y = gr * GH + 4 * q + r4;
                in[q] = inx && (y >= wy0) && (y < wy0 + WH);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    v[q][c] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
                    if (!first && in[q])
                        v[q][c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long)c * HW + y * W + x) * 4), 0, 16 /* sc1 */));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int y = gr * GH + 4 * q + r4;
                const float w = in[q] ? 1.0f + 0.0f * pad[lane] : 0.0f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (first || in[q]) {
                        f4 o = v[q][c] + (f4){w, w, w, w};
                        *reinterpret_cast<f4 *>(G + (long)b * C * HW + (long)c * HW + y * W + x) = o;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_exchange(lock, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// far pixels: K per workgroup; the pixel's four corners (2 x 2 cells) x 3 channels, through the granule locks (tensor initialised:
// state 2 everywhere) -- lanes 0..11 of a wave carry one add each.  MODE 0: the 12 atomics of today's kernel.
template <int MODE>
__global__ __launch_bounds__(NT, 8) void far_kernel(float *__restrict__ G, unsigned *__restrict__ locks, const int *__restrict__ targets, int K)
{
    __shared__ float pad[18 * 1024];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    pad[tid] = (float)tid;
    const int b = (int)(xcd_remap(blockIdx.x, gridDim.x) / (TILES_X * TILES_Y));
    const long HW = (long)H * W;
    for (int k = wave; k < K; k += NT / 64) {
        const int cell = targets[blockIdx.x * K + k];
        const int y = cell / W, x = cell % W;          // top-left corner; y < H - 1, x < W - 1
        if (MODE == 0) {
            if (lane < 12) {
                const int c = lane >> 2, dy = (lane >> 1) & 1, dx = lane & 1;
                unsafeAtomicAdd(G + ((long)b * C + c) * HW + (y + dy) * W + x + dx, 1.0f + 0.0f * pad[lane]);
            }
            continue;
        }
        // granules of the two rows / two columns, ascending order (no lock is awaited while a HIGHER one is held)
        const int ga = y / GH, gb = (y + 1) / GH, ha = x / GW, hb = (x + 1) / GW;
        unsigned *l[4] = {locks + (b * GR + ga) * GC + ha, locks + (b * GR + ga) * GC + hb, locks + (b * GR + gb) * GC + ha, locks + (b * GR + gb) * GC + hb};
        const bool need[4] = {true, hb != ha, gb != ga, (gb != ga) && (hb != ha)};
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (need[i]) while (atomicOr(l[i], 1u) & 1u) __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 12) {
            const int c = lane >> 2, dy = (lane >> 1) & 1, dx = lane & 1;
            float *p = G + ((long)b * C + c) * HW + (y + dy) * W + x + dx;
            const float old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 load
            *p = old + 1.0f + 0.0f * pad[lane];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (need[i]) __hip_atomic_exchange(l[i], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static float *G;
static unsigned *locks;
static int *offs, *bad, *targets;
static std::vector<int> h_offs(2 * NTILES);
static std::vector<float> h_G((size_t)B * C * H * W), expect((size_t)B * H * W);

static void expected_counts()
{
    std::fill(expect.begin(), expect.end(), 0.0f);
    for (int t = 0; t < NTILES; ++t) {
        const int tx = t % TILES_X, ty = (t / TILES_X) % TILES_Y, b = t / (TILES_X * TILES_Y);
        const int wx0 = tx * TW - R + h_offs[2 * t], wy0 = ty * TH - R + h_offs[2 * t + 1];
        for (int y = wy0; y < wy0 + WH; ++y)
            for (int x = wx0; x < wx0 + WW; ++x)
                if (x >= 0 && x < W && y >= 0 && y < H) expect[((size_t)b * H + y) * W + x] += 1.0f;
    }
}
static long check(float reps)
{
    hipMemcpy(h_G.data(), G, h_G.size() * 4, hipMemcpyDeviceToHost);
    long wrong = 0;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (size_t p = 0; p < (size_t)H * W; ++p)
                if (h_G[((size_t)b * C + c) * H * W + p] != reps * expect[(size_t)b * H * W + p]) ++wrong;
    return wrong;
}

template <int MODE, bool REMAP> static void run(const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, bestfill = 1e9f, sum = 0;
    const int REPS = 10;
    long wrong = -1;
    int h_bad = 0;
    for (int rep = 0; rep < REPS; ++rep) {
        hipMemset(locks, 0, B * GR * GC * 4);
        hipMemset(bad, 0, 4);
        if (MODE == 3) hipMemset(G, 0x7f, h_G.size() * 4); else hipMemset(G, 0, h_G.size() * 4);   // first touch must overwrite garbage
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((flush_kernel<MODE, REMAP>), dim3(NTILES), dim3(NT), 0, 0, G, locks, offs, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (rep >= 2) sum += ms;
        if (rep == 0) { wrong = check(1.0f); hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost); }
        // the same behind the zero fill it needs (MODE 3 needs none)
        hipEventRecord(e0);
        if (MODE != 3) hipMemsetAsync(G, 0, h_G.size() * 4, 0);
        hipMemsetAsync(locks, 0, B * GR * GC * 4, 0);
        hipLaunchKernelGGL((flush_kernel<MODE, REMAP>), dim3(NTILES), dim3(NT), 0, 0, G, locks, offs, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < bestfill) bestfill = ms;
    }
    printf("%-58s best %6.1f us  mean %6.1f us  | with its fills %6.1f us | wrong cells %ld  off-XCD workgroups %d\n", name, best * 1e3, sum / (REPS - 2) * 1e3,
           bestfill * 1e3, wrong, h_bad);
}

template <int MODE> static void run_far(const char *name, int K)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<int> h_t((size_t)NTILES * K);
    srand(7);
    for (auto &v : h_t) v = (rand() % (H - 1)) * W + rand() % (W - 1);
    hipMemcpy(targets, h_t.data(), h_t.size() * 4, hipMemcpyHostToDevice);
    float best = 1e9f;
    long wrong = 0;
    for (int rep = 0; rep < 6; ++rep) {
        hipMemset(G, 0, h_G.size() * 4);
        std::vector<unsigned> two(B * GR * GC, 2u);
        hipMemcpy(locks, two.data(), two.size() * 4, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((far_kernel<MODE>), dim3(NTILES), dim3(NT), 0, 0, G, locks, targets, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (rep == 0) {
            hipMemcpy(h_G.data(), G, h_G.size() * 4, hipMemcpyDeviceToHost);
            std::vector<float> ex(h_G.size(), 0.0f);
            for (int blk = 0; blk < NTILES; ++blk) {
                // xcd_remap on the host
                const unsigned NX = 8, q = NTILES / NX, r = NTILES % NX, xcd = blk % NX, idx = blk / NX;
                const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
                const int b = (int)((base + idx) / (TILES_X * TILES_Y));
                for (int k = 0; k < K; ++k) {
                    const int cell = h_t[(size_t)blk * K + k], y = cell / W, x = cell % W;
                    for (int c = 0; c < C; ++c)
                        for (int d = 0; d < 4; ++d) ex[((size_t)b * C + c) * H * W + (size_t)(y + (d >> 1)) * W + x + (d & 1)] += 1.0f;
                }
            }
            for (size_t i = 0; i < ex.size(); ++i) if (ex[i] != h_G[i]) ++wrong;
        }
    }
    printf("%-58s K = %3d per workgroup: best %6.1f us | wrong cells %ld\n", name, K, best * 1e3, wrong);
}

int main()
{
    hipMalloc(&G, h_G.size() * 4);
    hipMalloc(&locks, B * GR * GC * 4);
    hipMalloc(&offs, 2 * NTILES * 4);
    hipMalloc(&bad, 4);
    hipMalloc(&targets, (size_t)NTILES * 64 * 4);
    // 1. placement
    {
        int *d; hipMalloc(&d, 4096 * 4);
        hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, 0, d);
        std::vector<int> h(4096); hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
        int mism = 0; for (int i = 0; i < 4096; ++i) if (h[i] != i % 8) ++mism;
        printf("placement: %d of 4096 workgroups NOT on XCD blockIdx %% 8; first 16 ids:", mism);
        for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
        printf("\n");
    }
    for (int pass = 0; pass < 2; ++pass) {
        srand(3);
        for (int t = 0; t < NTILES; ++t) {
            h_offs[2 * t] = pass ? 4 * (rand() % 17 - 8) : 0;      // x: multiples of 4 in [-32, 32]
            h_offs[2 * t + 1] = pass ? rand() % 49 - 24 : 0;       // y: [-24, 24]
        }
        hipMemcpy(offs, h_offs.data(), h_offs.size() * 4, hipMemcpyHostToDevice);
        expected_counts();
        printf(pass ? "--- windows displaced by random offsets (x: multiples of 4 in +-32, y: +-24)\n" : "--- windows centred on their tiles\n");
        run<0, false>("atomics, tiles in raster order (images shared by XCDs)");
        run<0, true>("atomics, an XCD owns its image");
        run<1, false>("atomics sc1, raster order");
        run<1, true>("atomics sc1, an XCD owns its image");
        run<2, true>("granule locks + 16-B RMW, XCD owns its image");
        run<3, true>("granule locks, first touch stores (no zero fill)");
        run<2, false>("granule locks + RMW in RASTER order (expected WRONG)");
    }
    for (int K : {4, 20, 64}) {
        run_far<0>("far pixels: 12 atomics each", K);
        run_far<1>("far pixels: wave takes the granule locks, 12 RMW", K);
    }
    return 0;
}
