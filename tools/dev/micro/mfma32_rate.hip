// Issue rate of v_mfma_f32_16x16x4_f32 on gfx950: two independent accumulation chains per wave, one wave per SIMD.
// Prints shader cycles (s_memtime) and wall-clock (100 MHz counter) per MFMA -> cycles per instruction and the clock held.
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
__global__ void __launch_bounds__(256, 1) k(float* out, long long* cyc, int iters) {
    f64x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a0[2] + a1[3];
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 16);
    for (int grid : {1, 64, 256, 1024}) for (int iters : {128, 1024, 8192}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<grid, 256>>>(out, cyc, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<grid, 256>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        double n = 16.0 * iters;
        printf("grid %4d iters %5d: %.1f s_memtime ticks/MFMA, %.2f ns/MFMA (wall), kernel %.3f ms -> %.2f ns/MFMA, %.1f TFLOP/s\n", grid, iters,
               h[0] / n, h[1] * 10.0 / n, ms, ms * 1e6 / n / ((grid + 255) / 256), 2048.0 * n * grid * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
