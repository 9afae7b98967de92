#include <hip/hip_runtime.h>
#include <cstdio>
#include "diag_tile64.h"
__global__ void k(float* of, double* od) {
    const int l = threadIdx.x;
    const float2 a = bcbf::halves64((float)(l + 100));
    const double2 b = bcbf::halves64((double)(l + 100));
    of[2 * l] = a.x; of[2 * l + 1] = a.y; od[2 * l] = b.x; od[2 * l + 1] = b.y;
}
int main() {
    float* of; double* od; hipMalloc(&of, 512); hipMalloc(&od, 1024);
    k<<<1, 64>>>(of, od); float hf[128]; double hd[128];
    hipMemcpy(hf, of, 512, hipMemcpyDeviceToHost); hipMemcpy(hd, od, 1024, hipMemcpyDeviceToHost);
    for (int l : {0, 5, 31, 32, 37, 63}) printf("lane %d: f (%g, %g)  d (%g, %g)\n", l, hf[2 * l], hf[2 * l + 1], hd[2 * l], hd[2 * l + 1]);
    return 0;
}
