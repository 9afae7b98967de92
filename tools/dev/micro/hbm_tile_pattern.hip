// What does HBM give for the refit's operand reads?  A wave reads 4 KB "tiles" of a large buffer (2 GB, far beyond the 256 MB
// Infinity Cache) at pseudo-random origins, as 32 pieces of 128 bytes `stride` bytes apart (the packed operator: a tile's columns are
// ~2 KB apart) or contiguously (stride 128: a tile-major copy).  8 bytes per lane and load, as the kernel's float2 operand loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(64) read_tiles(const float2* __restrict__ buf, size_t ntiles_total, int tiles_per_wave, int stride_f2, size_t span_f2, float* out, int align_f2) {
    const int lane = threadIdx.x;
    size_t w = blockIdx.x;
    float acc = 0.f;
    unsigned long long h = w * 0x9E3779B97F4A7C15ull + 12345;
    for (int t = 0; t < tiles_per_wave; ++t) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        const size_t origin = ((h >> 20) % (span_f2 / align_f2)) * align_f2;
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = buf[origin + (size_t)((lane >> 4) + 4 * j) * stride_f2 + (lane & 15)];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
    }
    if (acc == 1234.5f) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    float2* buf; float* out;
    hipMalloc(&buf, bytes + (1 << 20)); hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes + (1 << 20));
    const int strides[] = {16, 128, 240, 256, 512};       // float2 units: 128 B (contiguous), 1 KB, 1920 B, 2 KB, 4 KB
    for (int occ = 0; occ < 2; ++occ)
    for (int si = 0; si < 5; ++si) {
        const int waves = occ == 0 ? 2048 : 8192, tpw = occ == 0 ? 512 : 128;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        read_tiles<<<waves, 64>>>(buf, 0, tpw, strides[si], bytes / 8 - 65536, out, strides[si] == 16 ? 512 : 16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        read_tiles<<<waves, 64>>>(buf, 0, tpw, strides[si], bytes / 8 - 65536, out, strides[si] == 16 ? 512 : 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("waves %d  piece stride %5d B: %.3f ms  %.2f TB/s\n", waves, strides[si] * 8, ms, (double)waves * tpw * 4096 / ms * 1e-9);
    }
    return 0;
}
