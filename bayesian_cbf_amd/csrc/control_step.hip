// One host call per control step: the whole unicycle hot path (task functions -> GP posterior ->
// chance-constraint terms + SOCP (one fused launch) -> plant Euler step) enqueued on one stream.  This is the entry
// point a reference-side `ControllerCLFBayesian.control` binds to (unicycle_move_to_pose.py:926-995);
// it only sequences the kernels of the other translation units.
#include "bcbf_common.h"
#include "unicycle_task.h"

namespace bcbf {
template <typename T>
int launch_unicycle_socp(const T* Mk, const T* Bk, const T* A, const T* sign, const T* w, const T* r, const T* relax_mask,
                         const T* rho, T* cones, int* cstatus, T* y, int* status, int* iters, int Bt, int max_iters,
                         const UnicycleTask<T>& task, void* stream);
}

// Two launches per step: the posterior kernel, then ONE kernel that forms the task rows (CLC + obstacle CBCs) from the
// state, the chance-constraint terms and cones from (M_k, B_k), solves the SOCP and advances the plant.
#define BCBF_CTRL(T, SUF, NAME, QUERY, STEP)                                                                           \
    extern "C" int NAME##SUF(                                                                                          \
        const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2, const T* Bm, const T* M0,     \
        const T* A, T* x, const T* plan, const T* dot_plan, const T* Kp, T clf_gamma, const T* centers,               \
        const T* radii, const T* tw, const T* gammas, T L_mean, const T* w, const T* r, const T* sign,                \
        const T* relax_mask, const T* rho, T* grad, T* cst, T* fhat, T* ghat, T* Mk, T* Bk, T* cones, int* cstatus,   \
        T* y, int* status, int* iters, T dt, T L_true, int Bt, int N, int Kob, int max_iters, int shared_gp,          \
        void* ev_start, void* ev_stop, void* stream) {                                                                 \
        if (Bt <= 0) return BCBF_OK;                                                                                   \
        if (!x || !grad || !cst || !fhat || !ghat || Kob < 0 || Kob + 1 > BCBF_MAX_QUAD_CONSTRAINTS) return BCBF_EINVAL;    \
        hipStream_t st = (hipStream_t)stream;                                                                          \
        if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);                                                  \
        /* Lop == NULL: no learned model in the loop -- (Mk, Bk) are the caller's (fixed-kernel model: 0 and I) */    \
        int rc = !Lop ? BCBF_OK : shared_gp ? QUERY##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, x, nullptr, Mk, Bk, nullptr, \
                                                        1, Bt, N, 3, 2, stream)                                        \
                           : STEP;                                                                                     \
        if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);                                                    \
        if (rc) return rc;                                                                                             \
        bcbf::UnicycleTask<T> task;                                                                                    \
        task.x = x; task.plan = plan; task.dot_plan = dot_plan; task.Kp = Kp; task.centers = centers;                  \
        task.radii = radii; task.tw = tw; task.gammas = gammas; task.clf_gamma = clf_gamma; task.L_mean = L_mean;      \
        task.dt = dt; task.L_true = L_true; task.grad = grad; task.cst = cst; task.fhat = fhat; task.ghat = ghat;      \
        task.Kob = Kob;                                                                                                \
        return bcbf::launch_unicycle_socp<T>(Mk, Bk, A, sign, w, r, relax_mask, rho, cones, cstatus, y, status, iters, \
                                             Bt, max_iters, task, stream);                                             \
    }
#define BCBF_STEP_RBF(SUF) bcbf_posterior_step_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, x, nullptr, Mk, Bk, Bt, N, 3, 2, stream)
#define BCBF_STEP_M52(SUF) bcbf_posterior_query_matern52_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, x, nullptr, Mk, Bk, nullptr, 0, Bt, N, 3, 2, stream)
BCBF_CTRL(float, f32, bcbf_unicycle_control_step_, bcbf_posterior_query_, BCBF_STEP_RBF(f32))
BCBF_CTRL(double, f64, bcbf_unicycle_control_step_, bcbf_posterior_query_, BCBF_STEP_RBF(f64))
// The same step on a model learned with the opt-in Matern-5/2 data kernel (bcbf.h: *_matern52; parity unpinned -- the
// reference has no Matern kernel): the posterior launch evaluates that kernel (one GP per instance: the streaming kernel;
// shared_gp: the matrix-core query), everything behind it is the same launch
BCBF_CTRL(float, f32, bcbf_unicycle_control_step_matern52_, bcbf_posterior_query_matern52_, BCBF_STEP_M52(f32))
BCBF_CTRL(double, f64, bcbf_unicycle_control_step_matern52_, bcbf_posterior_query_matern52_, BCBF_STEP_M52(f64))
#define BCBF_STEP_RM52(SUF) bcbf_posterior_query_rbfm52_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, x, nullptr, Mk, Bk, nullptr, 0, Bt, N, 3, 2, stream)
BCBF_CTRL(float, f32, bcbf_unicycle_control_step_rbfm52_, bcbf_posterior_query_rbfm52_, BCBF_STEP_RM52(f32))
BCBF_CTRL(double, f64, bcbf_unicycle_control_step_rbfm52_, bcbf_posterior_query_rbfm52_, BCBF_STEP_RM52(f64))

// The control step of a loop that LEARNS from itself (LearnedShiftInvariantDynamics, unicycle_move_to_pose.py:326-386: the
// controller hands every visited (x_t, u_t) to `train`; targets are finite differences of the visited states minus the mean
// model, inputs go through `_make_trans_invariant`).  Same two launches as bcbf_unicycle_control_step; the posterior is queried
// at `xq` (NULL: at x -- pass the shift-invariant input (0, 0, theta) of the current state) and the solve / plant launch also
// writes this step's observation row and the next query (UnicycleTask: obs_x, obs_uh, obs_y at row b * obs_ld; xq_next).
// flags: bit 0 = shift-invariant regressor inputs, bit 1 = the planner's target advances with the step (plan += dot_plan dt).
#define BCBF_CTRL_OBS(T, SUF)                                                                                          \
    extern "C" int bcbf_unicycle_control_step_observe_##SUF(                                                           \
        const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2, const T* Bm, const T* M0,     \
        const T* A, T* x, const T* plan, const T* dot_plan, const T* Kp, T clf_gamma, const T* centers,               \
        const T* radii, const T* tw, const T* gammas, T L_mean, const T* w, const T* r, const T* sign,                \
        const T* relax_mask, const T* rho, T* grad, T* cst, T* fhat, T* ghat, T* Mk, T* Bk, T* cones, int* cstatus,   \
        T* y, int* status, int* iters, T dt, T L_true, int Bt, int N, int Kob, int max_iters, int shared_gp,          \
        const T* xq, T* obs_x, T* obs_uh, T* obs_y, int obs_ld, T* xq_next, int flags,                                \
        void* ev_start, void* ev_stop, void* stream) {                                                                 \
        if (Bt <= 0) return BCBF_OK;                                                                                   \
        if (!x || Kob < 0 || Kob + 1 > BCBF_MAX_QUAD_CONSTRAINTS) return BCBF_EINVAL;                                  \
        if ((obs_x || obs_uh || obs_y) && (!obs_x || !obs_uh || !obs_y || obs_ld < 1 || !(dt > T(0)))) return BCBF_EINVAL;   \
        if (grad && (!cst || !fhat || !ghat)) return BCBF_EINVAL;                                                      \
        hipStream_t st = (hipStream_t)stream;                                                                          \
        if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);                                                  \
        const T* q = xq ? xq : x;                                                                                      \
        int rc = !Lop ? BCBF_OK : shared_gp ? bcbf_posterior_query_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, q, nullptr, Mk, Bk, \
                                                                       nullptr, 1, Bt, N, 3, 2, stream)                \
                           : bcbf_posterior_step_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, q, nullptr, Mk, Bk, Bt, N, 3, 2, stream); \
        if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);                                                    \
        if (rc) return rc;                                                                                             \
        bcbf::UnicycleTask<T> task;                                                                                    \
        task.x = x; task.plan = plan; task.dot_plan = dot_plan; task.Kp = Kp; task.centers = centers;                  \
        task.radii = radii; task.tw = tw; task.gammas = gammas; task.clf_gamma = clf_gamma; task.L_mean = L_mean;      \
        task.dt = dt; task.L_true = L_true; task.grad = grad; task.cst = cst; task.fhat = fhat; task.ghat = ghat;      \
        task.Kob = Kob; task.obs_x = obs_x; task.obs_uh = obs_uh; task.obs_y = obs_y; task.obs_ld = obs_ld;            \
        task.xq_next = xq_next; task.shift_invariant = flags & 1; task.advance_plan = (flags >> 1) & 1;                \
        return bcbf::launch_unicycle_socp<T>(Mk, Bk, A, sign, w, r, relax_mask, rho, cones, cstatus, y, status, iters, \
                                             Bt, max_iters, task, stream);                                             \
    }
BCBF_CTRL_OBS(float, f32)
BCBF_CTRL_OBS(double, f64)
