// Error reporting shared by all entry points.
#include "bcbf_common.h"
#include <stdio.h>
#include <string.h>

namespace bcbf {
static thread_local char g_err[256] = "";
thread_local const int* g_refit_only_bad = nullptr;

void set_error(const char* what, hipError_t err) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
}
int check_launch(const char* what) {
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { set_error(what, err); return BCBF_ELAUNCH; }
    return BCBF_OK;
}
}  // namespace bcbf

extern "C" int bcbf_version(void) { return BCBF_VERSION_MAJOR * 100 + BCBF_VERSION_MINOR; }
extern "C" const char* bcbf_last_error(void) { return bcbf::g_err; }

// ---------------------------------------------------------------------------------------------
// Read-only HBM ceiling of THIS device ("a measured ceiling from the same box"): every workgroup streams a
// contiguous chunk of the caller's buffer with 16-byte non-temporal loads and reduces it; the sum is stored only under a
// condition no data meets, so the kernel reads `bytes` and writes nothing.  The same access pattern as the roofline
// kernel's operator stream (posterior_step.hip), with no arithmetic behind it.
namespace bcbf {
using probe_f4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
__global__ void __launch_bounds__(256) hbm_read_probe_kernel(const probe_f4* __restrict__ src, float* __restrict__ sink, size_t per_wg) {
    const probe_f4* p = src + (size_t)blockIdx.x * per_wg;
    probe_f4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_wg; i += 256 * 4) {
        probe_f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + (size_t)u * 256 < per_wg) ? __builtin_nontemporal_load(p + i + (size_t)u * 256) : acc * 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[blockIdx.x & 255] = acc[0];
}
}  // namespace bcbf

extern "C" int bcbf_hbm_read_probe(const void* buf, size_t bytes, void* sink, size_t* bytes_read, void* stream) {
    if (!buf || !sink || !bytes_read || ((uintptr_t)buf & 15)) return BCBF_EINVAL;
    const int wgs = 8192;                               // 32 workgroups per CU: tools/dev/probe/read_bw.hip's best grid
    const size_t per_wg = bytes / 16 / wgs;
    *bytes_read = per_wg * 16 * wgs;
    if (per_wg == 0) return BCBF_EINVAL;
    hipLaunchKernelGGL(bcbf::hbm_read_probe_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const bcbf::probe_f4*)buf, (float*)sink, per_wg);
    return bcbf::check_launch("bcbf_hbm_read_probe");
}
