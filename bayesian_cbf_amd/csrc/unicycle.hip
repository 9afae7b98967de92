// Batched unicycle / Ackermann task functions: CLFCartesian, ObstacleCBF, prior dynamics and the
// explicit-Euler plant step (reference: bayes_cbf/unicycle_move_to_pose.py:112-139, 235-282,
// 522-615, 618-696; bayes_cbf/misc.py:317-318).  Elementwise, one lane per instance.
#include "bcbf_common.h"

namespace bcbf {

template <typename T> __device__ inline T normalize_radians(T th) {
    const T pi = T(3.14159265358979323846), two_pi = T(6.28318530717958647692);
    T y = fmod(th + pi, two_pi);         // python % : result has the sign of the divisor
    if (y < T(0)) y += two_pi;
    return y - pi;
}

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
    const T gx = plan[b * 3], gy = plan[b * 3 + 1], gth = plan[b * 3 + 2];
    const int K = 1 + Kob;
    // ---- CLFCartesian (clf_terms :527-534, _grad_clf_terms :564-589, _grad_clf_terms_wrt_goal :536-562)
    const T xd = gx - px, yd = gy - py;
    const T rho2 = xd * xd + yd * yd;
    const T phi = atan2(yd, xd);
    const T alpha = normalize_radians(th - phi), beta = normalize_radians(gth - phi);
    const T k0 = Kp[0], k1 = Kp[1], k2 = Kp[2];
    const T Vx = T(0.5) * k0 * rho2 + k1 * (T(1) - cos(alpha)) + k2 * (T(1) - cos(beta));
    const T sa = sin(alpha), sb = sin(beta);
    T* g0 = grad + (size_t)b * K * 3;
    g0[0] = -k0 * xd - k1 * sa * yd / rho2 - k2 * sb * yd / rho2;
    g0[1] = -k0 * yd + k1 * sa * xd / rho2 + k2 * sb * xd / rho2;
    g0[2] = k1 * sa;
    const T gg0 = k0 * xd + k1 * sa * yd / rho2 + k2 * sb * yd / rho2;
    const T gg1 = k0 * yd - k1 * sa * xd / rho2 - k2 * sb * xd / rho2;
    const T gg2 = k2 * sb;
    cst[(size_t)b * K] = gg0 * dot_plan[b * 3] + gg1 * dot_plan[b * 3 + 1] + gg2 * dot_plan[b * 3 + 2] + clf_gamma * Vx;
    // ---- ObstacleCBF (:624-630, :642-678)
    for (int k = 0; k < Kob; ++k) {
        const T hx = px - centers[((size_t)b * Kob + k) * 2], hy = py - centers[((size_t)b * Kob + k) * 2 + 1];
        const T rad = radii[(size_t)b * Kob + k];
        const T r2 = hx * hx + hy * hy, rn = sqrt(r2);
        const T radial = r2 - rad * rad;
        const T heading = cos(th) * hx / rn + sin(th) * hy / rn;
        const T al = atan2(hy, hx);
        T* gk = g0 + (size_t)(1 + k) * 3;
        gk[0] = tw[0] * T(2) * hx + tw[1] * (sin(al - th) * hy / r2);
        gk[1] = tw[0] * T(2) * hy + tw[1] * (-sin(al - th) * hx / r2);
        gk[2] = tw[1] * (-sin(th - al));
        cst[(size_t)b * K + 1 + k] = gammas[k] * (tw[0] * radial + tw[1] * heading);
    }
    // ---- prior dynamics AckermannDrive(L_mean) (:222-257)
    fhat[b * 3] = fhat[b * 3 + 1] = fhat[b * 3 + 2] = T(0);
    T* G = ghat + (size_t)b * 6;
    G[0] = cos(th); G[1] = T(0);
    G[2] = sin(th); G[3] = T(0);
    G[4] = T(0);    G[5] = T(1) / L_mean;
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

}  // namespace bcbf

extern "C" {
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
