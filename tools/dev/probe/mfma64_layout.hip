// probe: accumulator layout of v_mfma_f64_16x16x4_f64
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
__global__ void k(double* out) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    // A[i][k] = 100*i + k  (lane: i = j, k = g);  B[k][n] = (k == 0) ? n + 1 : 0   (lane: k = g, n = j)
    // => D[i][n] = A[i][0] * (n+1) = 100*i*(n+1)
    const double a = 100.0 * j + g;
    const double b = g == 0 ? (double)(j + 1) : 0.0;
    f64x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
int main() {
    double* o; hipMalloc(&o, 256 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    double h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane : {0, 1, 16, 17, 32, 48, 63})
        for (int r = 0; r < 4; ++r) {
            // decode: value = 100*i*(n+1): try n = lane%16
            const int n = lane % 16; const double v = h[lane * 4 + r];
            printf("lane %2d reg %d value %8.0f -> n=%d i=%g\n", lane, r, v, n, v / (100.0 * (n + 1)));
        }
    return 0;
}
