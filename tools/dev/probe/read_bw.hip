// Read-only HBM bandwidth ceiling on one MI355X: every workgroup streams a contiguous `chunk` of a large buffer with
// 16-byte non-temporal loads and reduces it (so the loads cannot be dropped).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;

template <int UNR>
__global__ void __launch_bounds__(256) read_kernel(const f4* __restrict__ src, float* __restrict__ out, size_t per_wg) {
    const f4* p = src + (size_t)blockIdx.x * per_wg;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_wg; i += 256 * UNR) {
        f4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(p + i + (size_t)u * 256);
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    f4* d; float* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 1 << 20);
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {1024, 2048, 4096, 8192, 16384, 65536}) {
        const size_t per_wg = bytes / 16 / wgs;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(read_kernel<4>, dim3(wgs), dim3(256), 0, 0, d, o, per_wg);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("unr4 wgs=%6d per_wg=%8zu KB  %.1f GB/s\n", wgs, per_wg * 16 / 1024, bytes * 10.0 / (ms * 1e-3) / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(read_kernel<8>, dim3(wgs), dim3(256), 0, 0, d, o, per_wg);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("unr8 wgs=%6d per_wg=%8zu KB  %.1f GB/s\n", wgs, per_wg * 16 / 1024, bytes * 10.0 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
