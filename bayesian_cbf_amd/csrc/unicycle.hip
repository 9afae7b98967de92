// Batched unicycle / Ackermann task functions: CLFCartesian, ObstacleCBF, prior dynamics and the
// explicit-Euler plant step (reference: bayes_cbf/unicycle_move_to_pose.py:112-139, 235-282,
// 522-615, 618-696; bayes_cbf/misc.py:317-318).  Elementwise, one lane per instance.
#include "unicycle_task.h"

namespace bcbf {

template <typename T>
__global__ void unicycle_constraints_kernel(const T* __restrict__ x, const T* __restrict__ plan,
                                            const T* __restrict__ dot_plan, const T* __restrict__ Kp, T clf_gamma,
                                            const T* __restrict__ centers, const T* __restrict__ radii,
                                            const T* __restrict__ tw, const T* __restrict__ gammas, T L_mean,
                                            T* __restrict__ grad, T* __restrict__ cst, T* __restrict__ fhat,
                                            T* __restrict__ ghat, int Bt, int Kob) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    const T px = x[b * 3], py = x[b * 3 + 1], th = x[b * 3 + 2];
    const int K = 1 + Kob;
    for (int k = 0; k < K; ++k) {
        T g[3], c;
        unicycle_row<T>(k, px, py, th, plan + (size_t)b * 3, dot_plan + (size_t)b * 3, Kp, clf_gamma,
                        centers + (size_t)b * Kob * 2, radii + (size_t)b * Kob, tw, gammas, g, c);
        T* gk = grad + ((size_t)b * K + k) * 3;
        gk[0] = g[0]; gk[1] = g[1]; gk[2] = g[2];
        cst[(size_t)b * K + k] = c;
    }
    fhat[b * 3] = fhat[b * 3 + 1] = fhat[b * 3 + 2] = T(0);
    T G[3][2];
    ackermann_g<T>(th, L_mean, G);
    T* Gp = ghat + (size_t)b * 6;
    for (int d = 0; d < 3; ++d) { Gp[d * 2] = G[d][0]; Gp[d * 2 + 1] = G[d][1]; }
}

template <typename T>
__global__ void unicycle_step_kernel(T* __restrict__ x, const T* __restrict__ u, T dt, T L_true, int Bt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    const T th = x[b * 3 + 2];
    const T u0 = u[b * 2], u1 = u[b * 2 + 1];
    x[b * 3] += cos(th) * u0 * dt;
    x[b * 3 + 1] += sin(th) * u0 * dt;
    x[b * 3 + 2] += u1 / L_true * dt;
}

// Safety bookkeeping of one closed-loop step of a Monte-Carlo rollout (rollouts.monte_carlo_safety_rollouts), one launch
// instead of eight elementwise ones: per trajectory  min_h = min(min_h, min_k cst_k / gamma_k)  over the obstacle rows
// (h_k(x_t) before the step; a non-finite h counts as -inf: it can never pass for "collision-free"), and -- only where
// the program was solved, the reference raises ValueError otherwise (unicycle_move_to_pose.py:954-964) --
// cost += sum_i w_i y_i^2;  fails += (status != 0).
template <typename T>
__global__ void rollout_stats_kernel(const T* __restrict__ cst, const T* __restrict__ y, const int* __restrict__ status,
                                     const T* __restrict__ w, const T* __restrict__ gammas, T* __restrict__ min_h,
                                     T* __restrict__ cost, int* __restrict__ fails, int Bt, int Kob, int nv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    T h = INFINITY;
    for (int k = 0; k < Kob; ++k) {
        T hk = cst[(size_t)b * (1 + Kob) + 1 + k] / gammas[k];
        if (!(hk == hk)) hk = -INFINITY;
        h = hk < h ? hk : h;
    }
    if (Kob > 0) min_h[b] = h < min_h[b] ? h : min_h[b];
    if (status[b] == 0) {
        T c = T(0);
        for (int i = 0; i < nv; ++i) c += w[(size_t)b * nv + i] * y[(size_t)b * nv + i] * y[(size_t)b * nv + i];
        cost[b] += c;
    } else {
        fails[b] += 1;
    }
}

}  // namespace bcbf

extern "C" {
#define BCBF_ROLLOUT_STATS(T, SUF)                                                                                    \
    int bcbf_rollout_stats_##SUF(const T* cst, const T* y, const int* status, const T* w, const T* gammas, T* min_h,  \
                                 T* cost, int* fails, int Bt, int Kob, int nv, void* stream) {                        \
        if (Bt <= 0) return BCBF_OK;                                                                                  \
        if (!cst || !y || !status || !w || !min_h || !cost || !fails || Kob < 0 || nv < 1 || (Kob > 0 && !gammas))    \
            return BCBF_EINVAL;                                                                                       \
        hipLaunchKernelGGL((bcbf::rollout_stats_kernel<T>), dim3((Bt + 255) / 256), dim3(256), 0, (hipStream_t)stream, \
                           cst, y, status, w, gammas, min_h, cost, fails, Bt, Kob, nv);                               \
        return bcbf::check_launch("rollout_stats");                                                                   \
    }
BCBF_ROLLOUT_STATS(float, f32)
BCBF_ROLLOUT_STATS(double, f64)
#undef BCBF_ROLLOUT_STATS
#define BCBF_UNI(T, SUF)                                                                                              \
    int bcbf_unicycle_constraints_##SUF(const T* x, const T* plan, const T* dot_plan, const T* Kp, T clf_gamma,       \
                                        const T* centers, const T* radii, const T* tw, const T* gammas, T L_mean,     \
                                        T* grad, T* cst, T* fhat, T* ghat, int Bt, int Kob, void* stream) {           \
        if (Bt <= 0) return BCBF_OK;                                                                                  \
        if (!x || !plan || !dot_plan || !Kp || !grad || !cst || !fhat || !ghat) return BCBF_EINVAL;                   \
        if (Kob < 0 || Kob + 1 > BCBF_MAX_CONSTRAINTS || (Kob > 0 && (!centers || !radii || !tw || !gammas)))         \
            return BCBF_EINVAL;                                                                                       \
        hipLaunchKernelGGL((bcbf::unicycle_constraints_kernel<T>), dim3((Bt + 255) / 256), dim3(256), 0,              \
                           (hipStream_t)stream, x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas, L_mean, \
                           grad, cst, fhat, ghat, Bt, Kob);                                                           \
        return bcbf::check_launch("unicycle_constraints");                                                            \
    }                                                                                                                 \
    int bcbf_unicycle_step_##SUF(T* x, const T* u, T dt, T L_true, int Bt, void* stream) {                            \
        if (Bt <= 0) return BCBF_OK;                                                                                  \
        if (!x || !u) return BCBF_EINVAL;                                                                             \
        hipLaunchKernelGGL((bcbf::unicycle_step_kernel<T>), dim3((Bt + 255) / 256), dim3(256), 0,                     \
                           (hipStream_t)stream, x, u, dt, L_true, Bt);                                                \
        return bcbf::check_launch("unicycle_step");                                                                   \
    }
BCBF_UNI(float, f32)
BCBF_UNI(double, f64)
}
