// Full predictive covariance of the batched query API, assembled in one launch.
//
// Replaces the tail of ControlAffineRegressorExact._custom_predict_matrix / custom_predict_fullmat
// (control_affine_model.py:1051-1091, 963-980): with G[b, b'] = W_b' W'_b' the Gram of the whitened cross-covariances of the test
// points (W = L^-1 Phi: bcbf_posterior_query with want_W; the Gram itself is one plain library GEMM on the caller's side)
//   BkXX[b, b', c, d] = k(x_b, x_b') B[c, d] - G[b, b', c, d]      (+ jitter[b (1+m) + c] where b = b', c = d: the make_psd draw, :1089)
//   Kron[(b (1+m) + c) n + i, (b' (1+m) + d) n + j] = BkXX[b, b', c, d] A[i, j]     (torch_kron(Bk2, A), :978)
// which the host did with ~20 small torch launches (prior kernel, jitter scatter, transposes, Kronecker product): the published
// speed test's call is bound by exactly those launches.  One thread per OUTPUT element, consecutive threads on consecutive
// columns of the output row (coalesced 4 / 8-byte stores: the 20 x 20 grid of the speed test writes 10 MB); every thread
// evaluates its own kernel value (n subtractions and an exp: cheaper than a pass that shares it).
#include "bcbf_common.h"

namespace bcbf {

template <typename T>
__global__ void __launch_bounds__(256)
predict_assemble_kernel(const T* __restrict__ G, const T* __restrict__ Xq, const T* __restrict__ Xqp, const T* __restrict__ ell,
                        const T* __restrict__ s2p, const T* __restrict__ Bm, const T* __restrict__ A, const T* __restrict__ jitter,
                        T* __restrict__ BkXX, T* __restrict__ Kron, int b, int bp, int n, int C, int kind) {
    const int nk = Kron != nullptr ? n : 1;                       // (BkXX alone: one thread per entry of it)
    const size_t ld = (size_t)bp * C * nk, total = (size_t)b * C * nk * ld;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int r = (int)(idx / ld), q = (int)(idx - (size_t)r * ld);
    const int i = r % nk, c = (r / nk) % C, qa = r / (nk * C);
    const int j = q % nk, d = (q / nk) % C, qb = q / (nk * C);
    // data kernel at (x_qa, x'_qb): RBF (the reference's) or the opt-in Matern-5/2 (kind 1)
    T d2 = T(0);
    for (int e = 0; e < n; ++e) { const T z = (Xq[(size_t)qa * n + e] - Xqp[(size_t)qb * n + e]) / ell[e]; d2 += z * z; }
    double shp, dshp_;
    kernel_shape(kind, (double)d2, [](double q_) { return exp(q_); }, shp, dshp_);      // RBF | Matern-5/2 | their product
    const T k = s2p[0] * (T)shp;
    T v = k * Bm[c * C + d] - G[(((size_t)qa * bp + qb) * C + c) * C + d];
    if (jitter != nullptr && qa == qb && c == d) v += jitter[(size_t)qa * C + c];
    if (Kron != nullptr) Kron[idx] = v * A[i * n + j];
    if (BkXX != nullptr && i == 0 && j == 0) BkXX[(((size_t)qa * bp + qb) * C + c) * C + d] = v;
}

template <typename T>
static int launch_predict_assemble(const T* G, const T* Xq, const T* Xqp, const T* ell, const T* s2, const T* Bm, const T* A,
                                   const T* jitter, T* BkXX, T* Kron, int b, int bp, int n, int m, int kind, void* stream) {
    if (b <= 0 || bp <= 0) return BCBF_OK;
    if (!G || !Xq || !Xqp || !ell || !s2 || !Bm || (Kron && !A) || (!BkXX && !Kron)) return BCBF_EINVAL;
    if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM || kind < 0 || kind >= BCBF_KINDS) return BCBF_EINVAL;
    if (jitter && b != bp) return BCBF_EINVAL;
    const int C = m + 1, nk = Kron ? n : 1;
    const size_t total = (size_t)b * C * nk * (size_t)bp * C * nk;
    if ((total + 255) / 256 > 0x7fffffffULL) return BCBF_EINVAL;
    hipLaunchKernelGGL((predict_assemble_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, Xq,
                       Xqp, ell, s2, Bm, A, jitter, BkXX, Kron, b, bp, n, C, kind);
    return check_launch("predict_assemble");
}

}  // namespace bcbf

extern "C" int bcbf_predict_assemble_f32(const float* G, const float* Xq, const float* Xqp, const float* ell, const float* s2,
                                         const float* Bm, const float* A, const float* jitter, float* BkXX, float* Kron, int b,
                                         int bp, int n, int m, int kernel_kind, void* stream) {
    return bcbf::launch_predict_assemble<float>(G, Xq, Xqp, ell, s2, Bm, A, jitter, BkXX, Kron, b, bp, n, m, kernel_kind, stream);
}
extern "C" int bcbf_predict_assemble_f64(const double* G, const double* Xq, const double* Xqp, const double* ell, const double* s2,
                                         const double* Bm, const double* A, const double* jitter, double* BkXX, double* Kron, int b,
                                         int bp, int n, int m, int kernel_kind, void* stream) {
    return bcbf::launch_predict_assemble<double>(G, Xq, Xqp, ell, s2, Bm, A, jitter, BkXX, Kron, b, bp, n, m, kernel_kind, stream);
}
