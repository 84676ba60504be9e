// Which element of D does (lane, register) hold for v_mfma_f64_16x16x4_f64?  (A: lane l supplies A[i = l % 16][k = l / 16], B: B[k = l / 16][j = l % 16].)
// Run 1: A[i][0] = i, B[0][j] = 1 -> D[i][j] = i;  run 2: A[i][0] = 1, B[0][j] = j -> D[i][j] = j.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_f64_layout scripts/ubench/mfma_f64_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out, int mode)
{
    const int l = threadIdx.x;
    const double a = (l / 16 == 0) ? (mode == 0 ? (double)(l % 16) : 1.0) : 0.0;
    const double b = (l / 16 == 0) ? (mode == 0 ? 1.0 : (double)(l % 16)) : 0.0;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main()
{
    double *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf(mode == 0 ? "row index i of D held by (lane, register):\n" : "column index j of D held by (lane, register):\n");
        for (int l = 0; l < 64; l += 1) {
            if (l % 16 == 0 || l % 16 == 1 || l % 16 == 15) printf("  lane %2d: %g %g %g %g\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
        }
    }
    return 0;
}
