// Unicycle / Ackermann task functions as device code shared by the stand-alone constraint kernel (unicycle.hip) and
// the fused per-step kernel (socp_quad.hip): CLFCartesian, ObstacleCBF, prior dynamics
// (reference: bayes_cbf/unicycle_move_to_pose.py:112-139, 235-282, 522-615, 618-696; bayes_cbf/misc.py:317-318).
#ifndef BCBF_UNICYCLE_TASK_H
#define BCBF_UNICYCLE_TASK_H
#include "bcbf_common.h"

namespace bcbf {

template <typename T> __device__ inline T normalize_radians(T th) {
    const T pi = T(3.14159265358979323846), two_pi = T(6.28318530717958647692);
    T y = fmod(th + pi, two_pi);         // python % : result has the sign of the divisor
    if (y < T(0)) y += two_pi;
    return y - pi;
}

// Row k of the constraint set of one instance: k = 0 the control-Lyapunov condition (gradient of V and the
// constant  grad_goal V' xdot_plan + gamma V), k >= 1 obstacle k-1 (gradient of h and gamma_k h).
template <typename T>
__device__ inline void unicycle_row(int k, T px, T py, T th, const T* __restrict__ plan, const T* __restrict__ dot_plan,
                                    const T* __restrict__ Kp, T clf_gamma, const T* __restrict__ centers,
                                    const T* __restrict__ radii, const T* __restrict__ tw, const T* __restrict__ gammas,
                                    T (&g)[3], T& cst) {
    if (k == 0) {
        // CLFCartesian (clf_terms :527-534, _grad_clf_terms :564-589, _grad_clf_terms_wrt_goal :536-562)
        const T gx = plan[0], gy = plan[1], gth = plan[2];
        const T xd = gx - px, yd = gy - py;
        const T rho2 = xd * xd + yd * yd;
        const T phi = atan2(yd, xd);
        const T alpha = normalize_radians(th - phi), beta = normalize_radians(gth - phi);
        const T k0 = Kp[0], k1 = Kp[1], k2 = Kp[2];
        const T Vx = T(0.5) * k0 * rho2 + k1 * (T(1) - cos(alpha)) + k2 * (T(1) - cos(beta));
        const T sa = sin(alpha), sb = sin(beta);
        g[0] = -k0 * xd - k1 * sa * yd / rho2 - k2 * sb * yd / rho2;
        g[1] = -k0 * yd + k1 * sa * xd / rho2 + k2 * sb * xd / rho2;
        g[2] = k1 * sa;
        const T gg0 = k0 * xd + k1 * sa * yd / rho2 + k2 * sb * yd / rho2;
        const T gg1 = k0 * yd - k1 * sa * xd / rho2 - k2 * sb * xd / rho2;
        const T gg2 = k2 * sb;
        cst = gg0 * dot_plan[0] + gg1 * dot_plan[1] + gg2 * dot_plan[2] + clf_gamma * Vx;
    } else {
        // ObstacleCBF (:624-630, :642-678)
        const int o = k - 1;
        const T hx = px - centers[o * 2], hy = py - centers[o * 2 + 1];
        const T rad = radii[o];
        const T r2 = hx * hx + hy * hy, rn = sqrt(r2);
        const T radial = r2 - rad * rad;
        const T heading = cos(th) * hx / rn + sin(th) * hy / rn;
        const T al = atan2(hy, hx);
        g[0] = tw[0] * T(2) * hx + tw[1] * (sin(al - th) * hy / r2);
        g[1] = tw[0] * T(2) * hy + tw[1] * (-sin(al - th) * hx / r2);
        g[2] = tw[1] * (-sin(th - al));
        cst = gammas[o] * (tw[0] * radial + tw[1] * heading);
    }
}

// The same row from values already in registers (the fused per-step kernel loads every input of an instance up front,
// one exposed memory latency instead of one per use): cen / rad / gam are those of THIS lane's obstacle.
template <typename T>
__device__ inline void unicycle_row_vals(int k, T px, T py, T th, const T (&plan)[3], const T (&dot_plan)[3],
                                         const T (&Kp)[3], T clf_gamma, T cx, T cy, T rad, T tw0, T tw1, T gam,
                                         T (&g)[3], T& cst) {
    if (k == 0) {
        const T xd = plan[0] - px, yd = plan[1] - py;
        const T rho2 = xd * xd + yd * yd;
        const T phi = atan2(yd, xd);
        const T alpha = normalize_radians(th - phi), beta = normalize_radians(plan[2] - phi);
        const T k0 = Kp[0], k1 = Kp[1], k2 = Kp[2];
        const T Vx = T(0.5) * k0 * rho2 + k1 * (T(1) - cos(alpha)) + k2 * (T(1) - cos(beta));
        const T sa = sin(alpha), sb = sin(beta);
        g[0] = -k0 * xd - k1 * sa * yd / rho2 - k2 * sb * yd / rho2;
        g[1] = -k0 * yd + k1 * sa * xd / rho2 + k2 * sb * xd / rho2;
        g[2] = k1 * sa;
        const T gg0 = k0 * xd + k1 * sa * yd / rho2 + k2 * sb * yd / rho2;
        const T gg1 = k0 * yd - k1 * sa * xd / rho2 - k2 * sb * xd / rho2;
        const T gg2 = k2 * sb;
        cst = gg0 * dot_plan[0] + gg1 * dot_plan[1] + gg2 * dot_plan[2] + clf_gamma * Vx;
    } else {
        const T hx = px - cx, hy = py - cy;
        const T r2 = hx * hx + hy * hy, rn = sqrt(r2);
        const T radial = r2 - rad * rad;
        const T heading = cos(th) * hx / rn + sin(th) * hy / rn;
        const T al = atan2(hy, hx);
        g[0] = tw0 * T(2) * hx + tw1 * (sin(al - th) * hy / r2);
        g[1] = tw0 * T(2) * hy + tw1 * (-sin(al - th) * hx / r2);
        g[2] = tw1 * (-sin(th - al));
        cst = gam * (tw0 * radial + tw1 * heading);
    }
}

// prior dynamics AckermannDrive(L_mean) (:222-257): f = 0, g = [[cos th, 0], [sin th, 0], [0, 1/L]]
template <typename T> __device__ inline void ackermann_g(T th, T L, T (&G)[3][2]) {
    G[0][0] = cos(th); G[0][1] = T(0);
    G[1][0] = sin(th); G[1][1] = T(0);
    G[2][0] = T(0);    G[2][1] = T(1) / L;
}

// Everything the fused per-step kernel needs to form the rows itself and to advance the plant afterwards.
template <typename T>
struct UnicycleTask {
    T* x;                                    // [Bt,3] state, advanced in place when dt > 0
    const T *plan, *dot_plan, *Kp, *centers, *radii, *tw, *gammas;
    T clf_gamma, L_mean, dt, L_true;
    T *grad, *cst, *fhat, *ghat;             // workspaces that expose the rows (may be NULL)
    int Kob;
    // Observation of THIS step for the learner (LearnedShiftInvariantDynamics.train / fit, unicycle_move_to_pose.py:326-386),
    // written with the plant step (dt > 0; all NULL = not wanted): row b * obs_ld of
    //   obs_x  [.,3]  the regressor's input: the state BEFORE the step, (0, 0, theta) when shift_invariant (:326-330)
    //   obs_uh [.,3]  (1, u)  -- u = 0 for an instance whose program was not solved (its state is frozen: a unicycle at rest)
    //   obs_y  [.,3]  (x_{t+1} - x_t) / dt  -  (f_mean + g_mean(L_mean) u)(input)   finite difference of the STORED states
    //                 minus the prior-mean dynamics (:364-372)
    // xq_next [Bt,3] (optional): the regressor's input at the NEW state -- the next step's posterior query.
    // advance_plan: the planner's target moves on with the step, plan += dot_plan dt (PiecewiseLinearPlanner.plan(t + 1),
    //               planner.py:54-64: a straight line at constant speed) -- `plan` is then written.
    T *obs_x = nullptr, *obs_uh = nullptr, *obs_y = nullptr, *xq_next = nullptr;
    int obs_ld = 1, shift_invariant = 1, advance_plan = 0;
};

// the observation row of one instance (see UnicycleTask): old state (x0, x1, th), new state (n0, n1, n2), applied u
template <typename T>
__device__ inline void unicycle_observe(const UnicycleTask<T>& task, int b, T x0, T x1, T th, T n0, T n1, T n2, T u0, T u1) {
    const bool si = task.shift_invariant != 0;
    if (task.obs_x != nullptr) {
        const size_t row = (size_t)b * task.obs_ld * 3;
        task.obs_x[row] = si ? T(0) : x0;
        task.obs_x[row + 1] = si ? T(0) : x1;
        task.obs_x[row + 2] = th;
        task.obs_uh[row] = T(1);
        task.obs_uh[row + 1] = u0;
        task.obs_uh[row + 2] = u1;
        T G[3][2];
        ackermann_g<T>(th, task.L_mean, G);                    // (g_mean depends on theta only: the same at the shifted input)
        task.obs_y[row] = (n0 - x0) / task.dt - (G[0][0] * u0 + G[0][1] * u1);
        task.obs_y[row + 1] = (n1 - x1) / task.dt - (G[1][0] * u0 + G[1][1] * u1);
        task.obs_y[row + 2] = (n2 - th) / task.dt - (G[2][0] * u0 + G[2][1] * u1);
    }
    if (task.advance_plan) {
        T* pl = const_cast<T*>(task.plan) + (size_t)b * 3;
        const T* dp = task.dot_plan + (size_t)b * 3;
        pl[0] += dp[0] * task.dt; pl[1] += dp[1] * task.dt; pl[2] += dp[2] * task.dt;
    }
    if (task.xq_next != nullptr) {
        task.xq_next[(size_t)b * 3] = si ? T(0) : n0;
        task.xq_next[(size_t)b * 3 + 1] = si ? T(0) : n1;
        task.xq_next[(size_t)b * 3 + 2] = n2;
    }
}

}  // namespace bcbf
#endif
