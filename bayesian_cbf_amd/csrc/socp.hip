// Generic small cone QP (runtime cone dimensions), one lane per instance -- the adapter behind the
// reference's optimizer_socp_* / optimizer_qp_cvxpy.  The hot CLF-CBF program is in socp_quad.hip.
// Solver core (coneqp_core.h): primal-dual interior point, everything in registers (fp32 iterates for the _f32 entry points, fp64 for _f64).  Algorithm: cvxopt `coneqp` (Vandenberghe 2010) --
// Mehrotra predictor-corrector, Nesterov-Todd scaling, step 0.99, sigma = (1-step)^3 -- with the
// iterates kept in scaled coordinates and the scaling of each second-order cone kept as an
// accumulated product M = W_1 W_2 ... (J-orthogonal up to beta: M J M' = beta^2 J, so
// M^-1 = J M' J / beta^2 needs no storage).  The KKT system reduces to an nv x nv SPD solve.
#include "bcbf_common.h"
#include "coneqp_core.h"

namespace bcbf {

// Generic small cone QP with runtime dimensions (reference optimizers.py adapters).
// Two instantiations: up to 4 second-order cones (the controllers' programs: registers), up to BCBF_MAX_CONSTRAINTS
// (programs with more cones than the quad kernels take, e.g. a third obstacle: larger arrays, slower, rare)
constexpr int GNV = 6, GL = 8, GNQ = 4, GNQ_WIDE = BCBF_MAX_CONSTRAINTS, GD = 6;
struct QDims { int d[GNQ_WIDE]; };

template <int NQMAX>
__global__ void __launch_bounds__(64)
coneqp_kernel(const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ G,
              const double* __restrict__ h, int nv, int l, QDims qd, int nq, double* __restrict__ x,
              int* __restrict__ status, int* __restrict__ iters, int Bt, int max_iters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    using Solver = ConeQP<double, GNV, GL, NQMAX, GD, false>;
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

}  // namespace bcbf

extern "C" {
int bcbf_coneqp_f64(const double* P, const double* q, const double* G, const double* h,
                    int nv, int l, const int* qdims, int nq,
                    double* x, int* status, int* iters, int Bt, int max_iters, void* stream) {
    using namespace bcbf;
    if (Bt <= 0) return BCBF_OK;
    if (!P || !q || !G || !h || !x || !status) return BCBF_EINVAL;
    if (nv < 1 || nv > GNV || l < 0 || l > GL || nq < 0 || nq > GNQ_WIDE || (nq > 0 && !qdims)) return BCBF_EINVAL;
    QDims qd;
    for (int k = 0; k < GNQ_WIDE; ++k) {
        qd.d[k] = k < nq ? qdims[k] : 0;
        if (k < nq && (qdims[k] < 2 || qdims[k] > GD)) return BCBF_EINVAL;
    }
    if (l + nq == 0) return BCBF_EINVAL;
    if (max_iters <= 0) max_iters = 100;
    if (nq <= GNQ)
        hipLaunchKernelGGL((coneqp_kernel<GNQ>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, q, G, h, nv, l,
                           qd, nq, x, status, iters, Bt, max_iters);
    else
        hipLaunchKernelGGL((coneqp_kernel<GNQ_WIDE>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, q, G, h,
                           nv, l, qd, nq, x, status, iters, Bt, max_iters);
    return check_launch("coneqp");
}
}
