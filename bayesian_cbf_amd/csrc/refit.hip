// K1 (dense kernel-matrix build, for callers that want K_b itself) and the C entry points of K1 + K2 (+ packing).
// The factorization kernels are refit_mfma.hip (fp32) and refit_mfma64.hip (fp64): blocked left-looking Cholesky,
// one workgroup per instance, writing the packed triangular operator the per-step kernel streams (bcbf.h "Lop").
//
#include "bcbf_common.h"

namespace bcbf {

template <typename T> __device__ inline T texp2(T x);
template <> __device__ inline float texp2<float>(float x) { return expf(x); }
template <> __device__ inline double texp2<double>(double x) { return exp(x); }


// --------------------------------------------------------------------------------------------
// K1 alone: dense K_b (API completeness / parity tests)
template <typename T>
__global__ void kb_build_kernel(const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ Bm,
                                const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ jitter,
                                T* __restrict__ Kb, int N, int n, int C, const T* __restrict__ lin, int kind) {
    const int b = blockIdx.y;
    const int i = blockIdx.x;
    const T* Xb = X + (size_t)b * N * n;
    const T* UHb = UH + (size_t)b * N * C;
    const T* Bmb = Bm + (size_t)b * C * C;
    T ub[BCBF_MAX_TASK_DIM];                           // (1+m) columns, or the (1+m) n of the expanded CoGP system
#pragma unroll
    for (int c = 0; c < BCBF_MAX_TASK_DIM; ++c) {
        T s = T(0);
        if (c < C) for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmb[a * C + c];
        ub[c] = s;
    }
    const T s2 = s2p[b];
    const T linv = lin != nullptr ? lin[b] : T(0);        // optional linear part: k = s2 (exp(..) + lin x'x')
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        T d2 = T(0), uu = T(0), dot = T(0);
        for (int d = 0; d < n; ++d) {
            const T z = (Xb[(size_t)i * n + d] - Xb[(size_t)j * n + d]) / ell[(size_t)b * n + d];
            d2 += z * z;
            dot += Xb[(size_t)i * n + d] * Xb[(size_t)j * n + d];
        }
#pragma unroll
        for (int a = 0; a < BCBF_MAX_TASK_DIM; ++a)
            if (a < C) uu += ub[a] * UHb[(size_t)j * C + a];
        T shape, dshape_;                                   // (bcbf_common.h: kernel_shape -- RBF | Matern-5/2 | their product)
        kernel_shape(kind, d2, [](T v) { return texp2<T>(v); }, shape, dshape_);
        T val = s2 * (shape + linv * dot) * uu;
        if (i == j && jitter) val += jitter[(size_t)b * N + i];
        Kb[((size_t)b * N + i) * N + j] = val;
    }
}

template <typename T>
static int launch_kb_build(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2, const T* jitter,
                           T* Kb, int Bt, int N, int n, int m, void* stream, const T* lin = nullptr, int kind = 0) {
    if (Bt <= 0) return BCBF_OK;
    if (!X || !UH || !Bm || !ell || !s2 || !Kb) return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m + 1 > BCBF_MAX_TASK_DIM) return BCBF_EINVAL;
    hipLaunchKernelGGL((kb_build_kernel<T>), dim3(N, Bt), dim3(256), 0, (hipStream_t)stream, X, UH, Bm, ell, s2,
                       jitter, Kb, N, n, m + 1, lin, kind);
    return check_launch("kb_build");
}

}  // namespace bcbf

extern "C" {
size_t bcbf_lop_elems_f32(int N) { return bcbf::lop_elems<4>(bcbf::round_up(N, bcbf::NB)); }
size_t bcbf_lop_elems_f64(int N) { return bcbf::lop_elems<2>(bcbf::round_up(N, bcbf::NB)); }

int bcbf_kb_build_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                      const float* jitter, float* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<float>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream);
}
int bcbf_kb_build_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                      const double* jitter, double* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<double>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream);
}
int bcbf_kb_build_rbflin_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                             const float* lin, const float* jitter, float* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<float>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream, lin);
}
int bcbf_kb_build_rbflin_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                             const double* lin, const double* jitter, double* Kb, int Bt, int N, int n, int m,
                             void* stream) {
    return bcbf::launch_kb_build<double>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream, lin);
}
int bcbf_kb_build_matern52_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                               const float* jitter, float* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<float>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream, nullptr, 1);
}
int bcbf_kb_build_matern52_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                               const double* jitter, double* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<double>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream, nullptr, 1);
}
int bcbf_kb_build_rbfm52_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                             const float* jitter, float* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<float>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream, nullptr, 2);
}
int bcbf_kb_build_rbfm52_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                             const double* jitter, double* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<double>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream, nullptr, 2);
}
// both precisions factor on the matrix cores (refit_mfma.hip, refit_mfma64.hip)
extern "C" int bcbf_refit_mfma_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                                   const float* jitter, const float* Kdense, float* Lop, float* UHB, float* Ldense,
                                   int* info, int Bt, int N, int n, int m, void* stream);
int bcbf_refit_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                   const float* jitter, float* Lop, float* UHB, float* Ldense, int* info,
                   int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
    return bcbf_refit_mfma_f32(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, Ldense, info, Bt, N, n, m, stream);
}
extern "C" int bcbf_refit_mfma_f64(const double* X, const double* UH, const double* Bm, const double* ell,
                                   const double* s2, const double* jitter, const double* Kdense, double* Lop,
                                   double* UHB, double* Ldense, int* info, int Bt, int N, int n, int m, void* stream);
int bcbf_refit_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                   const double* jitter, double* Lop, double* UHB, double* Ldense, int* info,
                   int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
    return bcbf_refit_mfma_f64(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, Ldense, info, Bt, N, n, m, stream);
}
// The x10 jitter retry of make_psd (control_affine_model.py:903-919) WITHOUT a host round trip: factor again, with the caller's
// (raised) jitter, only the instances whose previous attempt failed (prev_info[b] != 0); the others return at once and report 0.
// A caller launches it unconditionally after bcbf_refit -- a launch in which nothing failed costs microseconds.
int bcbf_refit_retry_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                         const float* jitter, float* Lop, float* UHB, const int* prev_info, int* info,
                         int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!X || !UH || !Bm || !ell || !s2 || !UHB || !prev_info || prev_info == info) return BCBF_EINVAL;
    bcbf::g_refit_only_bad = prev_info;
    const int rc = bcbf_refit_mfma_f32(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, nullptr, info, Bt, N, n, m, stream);
    bcbf::g_refit_only_bad = nullptr;
    return rc;
}
int bcbf_refit_retry_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                         const double* jitter, double* Lop, double* UHB, const int* prev_info, int* info,
                         int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!X || !UH || !Bm || !ell || !s2 || !UHB || !prev_info || prev_info == info) return BCBF_EINVAL;
    bcbf::g_refit_only_bad = prev_info;
    const int rc = bcbf_refit_mfma_f64(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, nullptr, info, Bt, N, n, m, stream);
    bcbf::g_refit_only_bad = nullptr;
    return rc;
}
int bcbf_potrf_f32(const float* Kb, float* Lop, float* Ldense, int* info, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Kb) return BCBF_EINVAL;
    return bcbf_refit_mfma_f32(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Kb, Lop, nullptr, Ldense, info, Bt, N, 0, 0, stream);
}
int bcbf_potrf_f64(const double* Kb, double* Lop, double* Ldense, int* info, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Kb) return BCBF_EINVAL;
    return bcbf_refit_mfma_f64(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Kb, Lop, nullptr, Ldense, info, Bt, N, 0, 0, stream);
}
}
