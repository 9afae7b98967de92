// One host call per control step: the whole unicycle hot path (task functions -> GP posterior ->
// chance-constraint terms + SOCP (one fused launch) -> plant Euler step) enqueued on one stream.  This is the entry
// point a reference-side `ControllerCLFBayesian.control` binds to (unicycle_move_to_pose.py:926-995);
// it only sequences the kernels of the other translation units.
#include "bcbf_common.h"

namespace bcbf {
template <typename T>
__global__ void apply_control_kernel(T* __restrict__ x, const T* __restrict__ y, T dt, T L_true, int Bt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    const T th = x[b * 3 + 2];
    const T u0 = y[b * 3], u1 = y[b * 3 + 1];          // y = [u0, u1, relax]
    x[b * 3] += cos(th) * u0 * dt;
    x[b * 3 + 1] += sin(th) * u0 * dt;
    x[b * 3 + 2] += u1 / L_true * dt;
}
}  // namespace bcbf

#define BCBF_CTRL(T, SUF)                                                                                              \
    extern "C" int bcbf_unicycle_control_step_##SUF(                                                                   \
        const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2, const T* Bm, const T* M0,     \
        const T* A, T* x, const T* plan, const T* dot_plan, const T* Kp, T clf_gamma, const T* centers,               \
        const T* radii, const T* tw, const T* gammas, T L_mean, const T* w, const T* r, const T* sign,                \
        const T* relax_mask, const T* rho, T* grad, T* cst, T* fhat, T* ghat, T* Mk, T* Bk, T* cones, int* cstatus,   \
        T* y, int* status, int* iters, T dt, T L_true, int Bt, int N, int Kob, int max_iters, int shared_gp,          \
        void* ev_start, void* ev_stop, void* stream) {                                                                                 \
        if (Bt <= 0) return BCBF_OK;                                                                                   \
        hipStream_t st = (hipStream_t)stream;                                                                          \
        int rc = bcbf_unicycle_constraints_##SUF(x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas, L_mean,\
                                                 grad, cst, fhat, ghat, Bt, Kob, stream);                              \
        if (rc) return rc;                                                                                             \
        if (ev_start) hipEventRecord((hipEvent_t)ev_start, st);                                                        \
        rc = shared_gp ? bcbf_posterior_query_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, x, nullptr, Mk, Bk, nullptr, 1, \
                                                    Bt, N, 3, 2, stream)                                               \
                       : bcbf_posterior_step_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, x, nullptr, Mk, Bk, Bt, N, 3, 2,  \
                                                   stream);                                                            \
        if (ev_stop) hipEventRecord((hipEvent_t)ev_stop, st);                                                          \
        if (rc) return rc;                                                                                             \
        rc = bcbf_cbc_socp_##SUF(Mk, Bk, A, grad, cst, sign, fhat, ghat, w, r, relax_mask, rho, nullptr, cones, cstatus,  \
                                 y, status, iters, Bt, 1 + Kob, 3, 2, max_iters, stream);                              \
        if (rc) return rc;                                                                                             \
        if (dt > T(0)) {                                                                                               \
            hipLaunchKernelGGL((bcbf::apply_control_kernel<T>), dim3((Bt + 255) / 256), dim3(256), 0, st, x, y, dt,    \
                               L_true, Bt);                                                                            \
            return bcbf::check_launch("apply_control");                                                                \
        }                                                                                                              \
        return BCBF_OK;                                                                                                \
    }
BCBF_CTRL(float, f32)
BCBF_CTRL(double, f64)
