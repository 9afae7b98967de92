// K9 + K10: batched primal-dual interior-point solver for the per-step cone program, one lane per
// instance, everything in registers (fp32 iterates for the _f32 entry points, fp64 for _f64).  Algorithm: cvxopt `coneqp` (Vandenberghe 2010) --
// Mehrotra predictor-corrector, Nesterov-Todd scaling, step 0.99, sigma = (1-step)^3 -- with the
// iterates kept in scaled coordinates and the scaling of each second-order cone kept as an
// accumulated product M = W_1 W_2 ... (J-orthogonal up to beta: M J M' = beta^2 J, so
// M^-1 = J M' J / beta^2 needs no storage).  The KKT system reduces to an nv x nv SPD solve.
#include "bcbf_common.h"
#include "coneqp_core.h"

namespace bcbf {

// --------------------------------------------------------------------------------------------
// CLF-CBF program, fixed sizes: nv = M+1, KC cones of dimension M+2.
template <typename T, int M_, int KC>
__global__ void __launch_bounds__(64)
socp_kernel(const T* __restrict__ w, const T* __restrict__ r, const T* __restrict__ cones,
            const T* __restrict__ relax_mask, const T* __restrict__ rho, T* __restrict__ y,
            int* __restrict__ status, int* __restrict__ iters, int Bt, int max_iters) {
    constexpr int NV = M_ + 1, D = M_ + 2, Q = (M_ + 1) * M_ + (M_ + 1) + M_ + 1;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    using R = T;   // the IPM iterates in the API's precision (fp32 entry: fp32 iterates)
    using Solver = ConeQP<R, NV, 0, KC, D, true>;
    Solver S;
    S.dims_fixed();
    R P[NV][NV], q[NV], G[KC * D][NV], h[KC * D], x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int j = 0; j < NV; ++j) P[i][j] = R(0);
        P[i][i] = R(2) * (R)w[(size_t)b * NV + i];
        q[i] = i < M_ ? -R(2) * (R)w[(size_t)b * NV + i] * (R)r[(size_t)b * M_ + i] : R(0);
    }
    const R rh = (R)rho[b];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const T* cn = cones + ((size_t)b * KC + k) * Q;
        const T* cA = cn;                       // [(M+1)][M]
        const T* cb = cn + (M_ + 1) * M_;       // [M+1]
        const T* cc = cb + (M_ + 1);            // [M]
        const T cd = cc[M_];
#pragma unroll
        for (int i = 0; i < M_; ++i) G[k * D][i] = -(R)cc[i];
        G[k * D][M_] = -(R)relax_mask[k];
        h[k * D] = (R)cd;
#pragma unroll
        for (int a = 0; a < M_ + 1; ++a) {
#pragma unroll
            for (int i = 0; i < M_; ++i) G[k * D + 1 + a][i] = -rh * (R)cA[a * M_ + i];
            G[k * D + 1 + a][M_] = R(0);
            h[k * D + 1 + a] = rh * (R)cb[a];
        }
    }
    int it = 0;
    const int st = S.solve(P, q, G, h, x, max_iters, &it);
#pragma unroll
    for (int i = 0; i < NV; ++i) y[(size_t)b * NV + i] = (T)x[i];
    status[b] = st;
    if (iters) iters[b] = it;
}

// Generic small cone QP with runtime dimensions (reference optimizers.py adapters).
constexpr int GNV = 6, GL = 8, GNQ = 4, GD = 6;
struct QDims { int d[GNQ]; };

__global__ void __launch_bounds__(64)
coneqp_kernel(const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ G,
              const double* __restrict__ h, int nv, int l, QDims qd, int nq, double* __restrict__ x,
              int* __restrict__ status, int* __restrict__ iters, int Bt, int max_iters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    using Solver = ConeQP<double, GNV, GL, GNQ, GD, false>;
    Solver S;
    S.dims_runtime(nv, l, qd.d, nq);
    const int K = S.K;
    double Pl[GNV][GNV], ql[GNV], Gl[Solver::MAXK][GNV], hl[Solver::MAXK], xl[GNV];
    for (int i = 0; i < GNV; ++i) {
        ql[i] = i < nv ? q[(size_t)b * nv + i] : 0.0;
        for (int j = 0; j < GNV; ++j) Pl[i][j] = (i < nv && j < nv) ? P[((size_t)b * nv + i) * nv + j] : 0.0;
    }
    for (int a = 0; a < Solver::MAXK; ++a) {
        hl[a] = a < K ? h[(size_t)b * K + a] : 0.0;
        for (int i = 0; i < GNV; ++i) Gl[a][i] = (a < K && i < nv) ? G[((size_t)b * K + a) * nv + i] : 0.0;
    }
    int it = 0;
    const int st = S.solve(Pl, ql, Gl, hl, xl, max_iters, &it);
    for (int i = 0; i < nv; ++i) x[(size_t)b * nv + i] = xl[i];
    status[b] = st;
    if (iters) iters[b] = it;
}

template <typename T, int M_>
static int dispatch_socp_k(const T* w, const T* r, const T* cones, const T* relax_mask, const T* rho, T* y,
                           int* status, int* iters, int Bt, int K, int max_iters, hipStream_t st) {
    dim3 block(64), grid((Bt + 63) / 64);
    switch (K) {
        case 1: hipLaunchKernelGGL((socp_kernel<T, M_, 1>), grid, block, 0, st, w, r, cones, relax_mask, rho, y, status, iters, Bt, max_iters); break;
        case 2: hipLaunchKernelGGL((socp_kernel<T, M_, 2>), grid, block, 0, st, w, r, cones, relax_mask, rho, y, status, iters, Bt, max_iters); break;
        case 3: hipLaunchKernelGGL((socp_kernel<T, M_, 3>), grid, block, 0, st, w, r, cones, relax_mask, rho, y, status, iters, Bt, max_iters); break;
        case 4: hipLaunchKernelGGL((socp_kernel<T, M_, 4>), grid, block, 0, st, w, r, cones, relax_mask, rho, y, status, iters, Bt, max_iters); break;
        default: return BCBF_EINVAL;
    }
    return check_launch("socp");
}

template <typename T>
static int launch_socp(const T* w, const T* r, const T* cones, const T* relax_mask, const T* rho, T* y,
                       int* status, int* iters, int Bt, int K, int m, int max_iters, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!w || !r || !cones || !relax_mask || !rho || !y || !status) return BCBF_EINVAL;
    if (max_iters <= 0) max_iters = 100;
    hipStream_t st = (hipStream_t)stream;
    switch (m) {
        case 1: return dispatch_socp_k<T, 1>(w, r, cones, relax_mask, rho, y, status, iters, Bt, K, max_iters, st);
        case 2: return dispatch_socp_k<T, 2>(w, r, cones, relax_mask, rho, y, status, iters, Bt, K, max_iters, st);
        case 3: return dispatch_socp_k<T, 3>(w, r, cones, relax_mask, rho, y, status, iters, Bt, K, max_iters, st);
        default: return BCBF_EINVAL;
    }
}

}  // namespace bcbf

extern "C" {
int bcbf_socp_f32(const float* w, const float* r, const float* cones, const float* relax_mask, const float* rho,
                  float* y, int* status, int* iters, int Bt, int K, int m, int max_iters, void* stream) {
    return bcbf::launch_socp<float>(w, r, cones, relax_mask, rho, y, status, iters, Bt, K, m, max_iters, stream);
}
int bcbf_socp_f64(const double* w, const double* r, const double* cones, const double* relax_mask, const double* rho,
                  double* y, int* status, int* iters, int Bt, int K, int m, int max_iters, void* stream) {
    return bcbf::launch_socp<double>(w, r, cones, relax_mask, rho, y, status, iters, Bt, K, m, max_iters, stream);
}
int bcbf_coneqp_f64(const double* P, const double* q, const double* G, const double* h,
                    int nv, int l, const int* qdims, int nq,
                    double* x, int* status, int* iters, int Bt, int max_iters, void* stream) {
    using namespace bcbf;
    if (Bt <= 0) return BCBF_OK;
    if (!P || !q || !G || !h || !x || !status) return BCBF_EINVAL;
    if (nv < 1 || nv > GNV || l < 0 || l > GL || nq < 0 || nq > GNQ || (nq > 0 && !qdims)) return BCBF_EINVAL;
    QDims qd;
    for (int k = 0; k < GNQ; ++k) {
        qd.d[k] = k < nq ? qdims[k] : 0;
        if (k < nq && (qdims[k] < 2 || qdims[k] > GD)) return BCBF_EINVAL;
    }
    if (l + nq == 0) return BCBF_EINVAL;
    if (max_iters <= 0) max_iters = 100;
    hipLaunchKernelGGL(coneqp_kernel, dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, q, G, h, nv, l, qd,
                       nq, x, status, iters, Bt, max_iters);
    return check_launch("coneqp");
}
}
