// K8 + K9 + K10 fused, cone-parallel: four adjacent lanes (a "quad") solve one instance of the
// CLF-CBF second-order-cone program; lane k of the quad owns constraint / cone k (K <= 4).
//
//   prologue  lane k forms its chance-constraint terms (bfe,e,V,bfv,v) from (Mk, Bk, A, grad_k, ...)
//             and their cone rows G_k, h_k  (closed form of cbc2_quadratic_terms + cone conversion);
//   loop      the interior-point iteration of coneqp_core.h with the per-cone work (NT scaling, scaled
//             residuals, Jordan algebra, step length) done by the owning lane and the coupling terms
//             (G'z, H = P + sum Gt_k'Gt_k, right-hand sides, gap, step) combined by xor-butterfly
//             reductions inside the quad; the nv x nv reduced KKT system is factored redundantly by
//             all four lanes in fp64.
// Everything a lane needs stays in registers; 64 instances x 4 lanes = 4 waves per 64 instances, so a
// 4096-instance batch is 256 waves instead of 64.
#include "bcbf_common.h"
#include "unicycle_task.h"

namespace bcbf {

template <typename R> struct QTol;
template <> struct QTol<double> {
    static __device__ inline double feas() { return 1e-9; }
    static __device__ inline double gap() { return 1e-9; }
    static __device__ inline double accept() { return 1e-7; }
    static __device__ inline double infeas() { return 1e-8; }
};
__device__ inline float qdv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ inline double qdv(double a, double b) { return a / b; }
__device__ inline float qsq(float a) { return __builtin_amdgcn_sqrtf(a); }
__device__ inline double qsq(double a) { return __builtin_sqrt(a); }

// Quad (4-lane) butterflies with DPP quad_perm moves: VALU-latency cross-lane exchange, no LDS crossbar.
template <int CTRL> __device__ inline float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ inline double dpp_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
constexpr int QP_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int QP_XOR2 = 0x4E;   // quad_perm [2,3,0,1]
__device__ inline float quad_sum(float v) { v += dpp_f<QP_XOR1>(v); v += dpp_f<QP_XOR2>(v); return v; }
__device__ inline double quad_sum(double v) { v += dpp_d<QP_XOR1>(v); v += dpp_d<QP_XOR2>(v); return v; }
__device__ inline float quad_max(float v) { v = fmaxf(v, dpp_f<QP_XOR1>(v)); v = fmaxf(v, dpp_f<QP_XOR2>(v)); return v; }
__device__ inline double quad_max(double v) { v = fmax(v, dpp_d<QP_XOR1>(v)); v = fmax(v, dpp_d<QP_XOR2>(v)); return v; }

// One second-order cone of dimension D held by one lane.
template <typename R, int D>
struct ConeLane {
    R M[D][D], beta2, lam[D], s[D], z[D];

    static __device__ inline R jdet(const R* a) {
        R d = a[0] * a[0];
#pragma unroll
        for (int i = 1; i < D; ++i) d -= a[i] * a[i];
        return d;
    }
    // NT scaling of (s_, z_): returns beta, fills w; lam = W z_
    __device__ inline R nt(const R* s_, const R* z_, R* w) {
        R sz = s_[0] * z_[0];
#pragma unroll
        for (int i = 1; i < D; ++i) sz += s_[i] * z_[i];
        const R sn = qsq(jdet(s_)), zn = qsq(jdet(z_));
        const R isn = qdv(R(1), sn), izn = qdv(R(1), zn);
        const R gamma = qsq((R(1) + sz * isn * izn) * R(0.5));
        const R ig = qdv(R(0.5), gamma);
        w[0] = (s_[0] * isn + z_[0] * izn) * ig;
#pragma unroll
        for (int i = 1; i < D; ++i) w[i] = (s_[i] * isn - z_[i] * izn) * ig;
        const R beta = qsq(sn * izn);
        R w1z = 0;
#pragma unroll
        for (int i = 1; i < D; ++i) w1z += w[i] * z_[i];
        lam[0] = beta * (w[0] * z_[0] + w1z);
        const R cf = z_[0] + qdv(w1z, R(1) + w[0]);
#pragma unroll
        for (int i = 1; i < D; ++i) lam[i] = beta * (z_[i] + cf * w[i]);
        return beta;
    }
    __device__ inline void init_scaling() {
        R w[D];
        const R beta = nt(s, z, w);
        beta2 = beta * beta;
        const R iw0 = qdv(R(1), R(1) + w[0]);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                R v;
                if (a == 0) v = w[c];
                else if (c == 0) v = w[a];
                else v = (a == c ? R(1) : R(0)) + w[a] * w[c] * iw0;
                M[a][c] = beta * v;
            }
    }
    // out = M^-1 v = J M' J v / beta^2
    __device__ inline void minv(const R* v, R* out, R ib2) const {
#pragma unroll
        for (int a = 0; a < D; ++a) {
            R t = 0;
#pragma unroll
            for (int c = 0; c < D; ++c) t += M[c][a] * (c == 0 ? v[c] : -v[c]);
            out[a] = (a == 0 ? t : -t) * ib2;
        }
    }
    // lam o y = x  (in place)
    __device__ inline void sinv(R* x) const {
        R det = lam[0] * lam[0], lx = 0;
#pragma unroll
        for (int i = 1; i < D; ++i) { det -= lam[i] * lam[i]; lx += lam[i] * x[i]; }
        const R y0 = qdv(lam[0] * x[0] - lx, det);
        const R il0 = qdv(R(1), lam[0]);
        x[0] = y0;
#pragma unroll
        for (int i = 1; i < D; ++i) x[i] = (x[i] - y0 * lam[i]) * il0;
    }
    static __device__ inline void sprod(const R* x, const R* y, R* out) {
        R d = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) d += x[i] * y[i];
        out[0] = d;
#pragma unroll
        for (int i = 1; i < D; ++i) out[i] = x[0] * y[i] + y[0] * x[i];
    }
    // -min eigenvalue of P(lam^-1/2) x
    __device__ inline R scaled_max_step(const R* x) const {
        const R nrm = qsq(jdet(lam)), inrm = qdv(R(1), nrm);
        const R lb0 = lam[0] * inrm;
        R lx = 0;
#pragma unroll
        for (int i = 1; i < D; ++i) lx += lam[i] * inrm * x[i];
        const R y0 = (lb0 * x[0] - lx) * inrm;
        const R coef = -x[0] + qdv(lx, R(1) + lb0);
        R nn = 0;
#pragma unroll
        for (int i = 1; i < D; ++i) { const R ya = (x[i] + coef * lam[i] * inrm) * inrm; nn += ya * ya; }
        return qsq(nn) - y0;
    }
    __device__ inline void advance(const R* dst, const R* dzt, R step, R ib2) {
        // unscaled iterates with the old scaling, then the new NT point in scaled coordinates
#pragma unroll
        for (int a = 0; a < D; ++a) {
            R vs = 0, vz = 0;
#pragma unroll
            for (int c = 0; c < D; ++c) { vs += M[a][c] * dst[c]; vz += M[a][c] * (c == 0 ? dzt[c] : -dzt[c]); }
            s[a] += step * vs;
            z[a] += step * (a == 0 ? vz : -vz) * ib2;
        }
        R st[D], zt[D], w[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { st[i] = lam[i] + step * dst[i]; zt[i] = lam[i] + step * dzt[i]; }
        const R beta = nt(st, zt, w);
        beta2 *= beta * beta;
        const R iw0 = qdv(R(1), R(1) + w[0]);
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const R m0 = M[a][0];
            R mw = 0;
#pragma unroll
            for (int c = 1; c < D; ++c) mw += M[a][c] * w[c];
            const R cf = m0 + mw * iw0;
            M[a][0] = beta * (m0 * w[0] + mw);
#pragma unroll
            for (int c = 1; c < D; ++c) M[a][c] = beta * (M[a][c] + cf * w[c]);
        }
    }
};

// 1/sqrt(d) in double: hardware estimate + two Newton steps (a divide or a sqrt in fp64 is a ~15-instruction dependent
// chain, and the reduced KKT system is factored and solved twice per iteration)
__device__ inline double rsqrt_d(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}
// Cholesky H = L L' in place; the diagonal holds 1 / L_jj so that the solves multiply.
template <int NV>
__device__ inline bool chol3(double (*H)[NV]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        double d = H[j][j];
#pragma unroll
        for (int k = 0; k < NV; ++k) if (k < j) d -= H[j][k] * H[j][k];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double il = rsqrt_d(d);
        H[j][j] = il;
#pragma unroll
        for (int i = 0; i < NV; ++i) if (i > j) {
            double v = H[i][j];
#pragma unroll
            for (int k = 0; k < NV; ++k) if (k < j) v -= H[i][k] * H[j][k];
            H[i][j] = v * il;
        }
    }
    return ok;
}
template <int NV>
__device__ inline void chol3_solve(const double (*H)[NV], double* b) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < NV; ++k) if (k < i) v -= H[i][k] * b[k];
        b[i] = v * H[i][i];
    }
#pragma unroll
    for (int ii = 0; ii < NV; ++ii) {
        const int i = NV - 1 - ii;
        double v = b[i];
#pragma unroll
        for (int k = 0; k < NV; ++k) if (k > i) v -= H[k][i] * b[k];
        b[i] = v * H[i][i];
    }
}

// FROM_TERMS: build the cone rows from the GP posterior (cbc_terms fused); else read packed cones.
// UNI (with FROM_TERMS, n = 3, m = 2): the unicycle step in one launch -- each lane also forms its own task row
// (CLC / obstacle k-1: unicycle_task.h) from the state instead of reading grad/cst/fhat/ghat, and lane 0 advances the
// plant with the solution (explicit Euler, unicycle_move_to_pose.py:277-282).
template <typename T, int M_, bool FROM_TERMS, bool UNI>
__global__ void __launch_bounds__(64)
socp_quad_kernel(const T* __restrict__ w, const T* __restrict__ r, const T* __restrict__ cones_in,
                 const T* __restrict__ relax_mask, const T* __restrict__ rho,
                 // FROM_TERMS inputs
                 const T* __restrict__ Mk, const T* __restrict__ Bk, const T* __restrict__ A,
                 const T* __restrict__ grad, const T* __restrict__ cst, const T* __restrict__ sign,
                 const T* __restrict__ fhat, const T* __restrict__ ghat, int n,
                 T* __restrict__ terms_out, T* __restrict__ cones_out, int* __restrict__ cstatus,
                 T* __restrict__ y, int* __restrict__ status, int* __restrict__ iters, int Bt, int K, int max_iters,
                 UnicycleTask<T> task) {
    // The interior-point iterates are fp64 for BOTH entry-point precisions (T is the I/O type): with fp32 iterates the
    // rounding floor of the scaled residuals left ~0.1 % of near-degenerate programs 2e-3 away from the optimum, above
    // the 1e-3 the fp32 path has to hold.  The kernel is latency bound on one wave per CU either way (47 -> 65 us per
    // 4096 programs) and runs beside the posterior stream of the other half batch in the pipelined step.
    using R = double;
#ifdef BCBF_SOCP_SETPRIO
    __builtin_amdgcn_s_setprio(BCBF_SOCP_SETPRIO);     // tuning knob: issue priority over co-resident streaming waves
#endif
    constexpr int NV = M_ + 1, D = M_ + 2, C = M_ + 1;
    constexpr int Q = (M_ + 1) * M_ + (M_ + 1) + M_ + 1;
    constexpr int TW = M_ + 1 + M_ * M_ + M_ + 1;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gid >> 2, k = gid & 3;
    const bool inst_ok = b < Bt;
    const int bb = inst_ok ? b : Bt - 1;          // out-of-range quads shadow the last instance (no stores)
    const bool active = k < K;
    const int kk = active ? k : 0;

    // ---- cone rows  G_k [D][NV], h_k [D]
    R G[D][NV], h[D];
    const R rh = (R)rho[bb];
    int bad = 0;
    {
        R cA[C][M_], cb[C], cc[M_], cd;
        if (FROM_TERMS) {
            double gd[BCBF_MAX_STATE_DIM];
            double a_h = 0.0, e = 0.0;
            double bfe[M_], Asq[C][C], L[C][C];
            if constexpr (UNI) {
                // the unicycle step (n = 3, m = 2): every input of this lane is loaded first, unconditionally, so the
                // prologue pays ONE memory latency (it used to pay one per branch / use: ~20 round trips)
                static_assert(M_ == 2, "unicycle: two controls");
                T xs[3], pl[3], dp[3], kp[3], Mkl[9], Bkl[9], Al[9];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    xs[d] = task.x[(size_t)bb * 3 + d];
                    pl[d] = task.plan[(size_t)bb * 3 + d];
                    dp[d] = task.dot_plan[(size_t)bb * 3 + d];
                    kp[d] = task.Kp[d];
                }
#pragma unroll
                for (int a = 0; a < 9; ++a) {
                    Mkl[a] = Mk[(size_t)bb * 9 + a];
                    Bkl[a] = Bk[(size_t)bb * 9 + a];
                    Al[a] = A[(size_t)bb * 9 + a];
                }
                const double sg = (double)sign[kk];
                T cx = T(0), cy = T(0), rad = T(1), gam = T(0), tw0 = T(0), tw1 = T(0);
                if (task.Kob > 0) {                              // uniform
                    const int o = min(max(kk - 1, 0), task.Kob - 1);
                    cx = task.centers[((size_t)bb * task.Kob + o) * 2];
                    cy = task.centers[((size_t)bb * task.Kob + o) * 2 + 1];
                    rad = task.radii[(size_t)bb * task.Kob + o];
                    gam = task.gammas[o];
                    tw0 = task.tw[0];
                    tw1 = task.tw[1];
                }
                T urow[3], ucst = T(0), uG[3][2];          // this lane's task row and the prior input matrix
                unicycle_row_vals<T>(kk, xs[0], xs[1], xs[2], pl, dp, kp, task.clf_gamma, cx, cy, rad, tw0, tw1, gam, urow, ucst);
                ackermann_g<T>(xs[2], task.L_mean, uG);
#pragma unroll
                for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) gd[d] = d < 3 ? (double)urow[d < 3 ? d : 0] : 0.0;
                if (inst_ok && active && task.grad != nullptr) {
                    for (int d = 0; d < 3; ++d) task.grad[((size_t)b * K + k) * 3 + d] = urow[d];
                    task.cst[(size_t)b * K + k] = ucst;
                    if (k == 0)
                        for (int d = 0; d < 3; ++d) {
                            task.fhat[(size_t)b * 3 + d] = T(0);
                            task.ghat[((size_t)b * 3 + d) * 2] = uG[d][0];
                            task.ghat[((size_t)b * 3 + d) * 2 + 1] = uG[d][1];
                        }
                }
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    double t = 0.0;
#pragma unroll
                    for (int e2 = 0; e2 < 3; ++e2) t += (double)Al[d * 3 + e2] * gd[e2];
                    a_h += gd[d] * t;
                }
                e = (double)ucst;
#pragma unroll
                for (int d = 0; d < 3; ++d) e += gd[d] * (double)Mkl[d * 3];
                e *= sg;
#pragma unroll
                for (int i = 0; i < M_; ++i) {
                    double s_ = 0.0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) s_ += ((double)uG[d][i] + (double)Mkl[d * 3 + 1 + i]) * gd[d];
                    bfe[i] = sg * s_;
                }
#pragma unroll
                for (int a = 0; a < C; ++a)
#pragma unroll
                    for (int c = 0; c < C; ++c) { Asq[a][c] = a_h * (double)Bkl[a * 3 + c]; L[a][c] = 0.0; }
            } else {
                const T* Mkb = Mk + (size_t)bb * n * C;
                const T* Bkb = Bk + (size_t)bb * C * C;
                const T* Ab = A + (size_t)bb * n * n;
                const double sg = (double)sign[kk];
                const T* g = grad + ((size_t)bb * K + kk) * n;
#pragma unroll
                for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) gd[d] = d < n ? (double)g[d] : 0.0;
                for (int d = 0; d < n; ++d) {
                    double t = 0.0;
                    for (int e2 = 0; e2 < n; ++e2) t += (double)Ab[d * n + e2] * gd[e2];
                    a_h += gd[d] * t;
                }
                e = (double)cst[(size_t)bb * K + kk];
                for (int d = 0; d < n; ++d) e += gd[d] * ((double)fhat[(size_t)bb * n + d] + (double)Mkb[d * C]);
                e *= sg;
#pragma unroll
                for (int i = 0; i < M_; ++i) {
                    double s_ = 0.0;
                    for (int d = 0; d < n; ++d)
                        s_ += ((double)ghat[((size_t)bb * n + d) * M_ + i] + (double)Mkb[d * C + 1 + i]) * gd[d];
                    bfe[i] = sg * s_;
                }
#pragma unroll
                for (int a = 0; a < C; ++a)
#pragma unroll
                    for (int c = 0; c < C; ++c) { Asq[a][c] = a_h * (double)Bkb[a * C + c]; L[a][c] = 0.0; }
            }
            if (terms_out && inst_ok && active) {
                T* t = terms_out + ((size_t)b * K + k) * TW;
                for (int i = 0; i < M_; ++i) t[i] = (T)bfe[i];
                t[M_] = (T)e;
                for (int i = 0; i < M_; ++i) for (int j = 0; j < M_; ++j) t[M_ + 1 + i * M_ + j] = (T)Asq[1 + i][1 + j];
                for (int i = 0; i < M_; ++i) t[M_ + 1 + M_ * M_ + i] = (T)(2.0 * Asq[1 + i][0]);
                t[M_ + 1 + M_ * M_ + M_] = (T)Asq[0][0];
            }
#pragma unroll
            for (int j = 0; j < C; ++j) {
                double d = Asq[j][j];
#pragma unroll
                for (int q = 0; q < C; ++q) if (q < j) d -= L[j][q] * L[j][q];
                if (!(d > 0.0)) { bad = BCBF_SOCP_BADCONE; d = 1.0; }
                const double ljj = __builtin_sqrt(d);
                L[j][j] = ljj;
#pragma unroll
                for (int i = 0; i < C; ++i) if (i > j) {
                    double s_ = Asq[i][j];
#pragma unroll
                    for (int q = 0; q < C; ++q) if (q < j) s_ -= L[i][q] * L[j][q];
                    L[i][j] = s_ / ljj;
                }
            }
#pragma unroll
            for (int a = 0; a < C; ++a) {
                cb[a] = (R)L[0][a];
#pragma unroll
                for (int i = 0; i < M_; ++i) cA[a][i] = (R)L[1 + i][a];
            }
#pragma unroll
            for (int i = 0; i < M_; ++i) cc[i] = (R)bfe[i];
            cd = (R)e;
            if (cones_out && inst_ok && active) {
                T* cn = cones_out + ((size_t)b * K + k) * Q;
                for (int a = 0; a < C; ++a) for (int i = 0; i < M_; ++i) cn[a * M_ + i] = (T)cA[a][i];
                for (int a = 0; a < C; ++a) cn[C * M_ + a] = (T)cb[a];
                for (int i = 0; i < M_; ++i) cn[C * M_ + C + i] = (T)cc[i];
                cn[C * M_ + C + M_] = (T)cd;
            }
            if (cstatus && inst_ok && active) cstatus[(size_t)b * K + k] = bad;
        } else {
            const T* cn = cones_in + ((size_t)bb * K + kk) * Q;
#pragma unroll
            for (int a = 0; a < C; ++a) {
                cb[a] = (R)cn[C * M_ + a];
#pragma unroll
                for (int i = 0; i < M_; ++i) cA[a][i] = (R)cn[a * M_ + i];
            }
#pragma unroll
            for (int i = 0; i < M_; ++i) cc[i] = (R)cn[C * M_ + C + i];
            cd = (R)cn[C * M_ + C + M_];
        }
#pragma unroll
        for (int i = 0; i < M_; ++i) G[0][i] = -cc[i];
        G[0][M_] = -(R)relax_mask[kk];
        h[0] = cd;
#pragma unroll
        for (int a = 0; a < C; ++a) {
#pragma unroll
            for (int i = 0; i < M_; ++i) G[1 + a][i] = -rh * cA[a][i];
            G[1 + a][M_] = R(0);
            h[1 + a] = rh * cb[a];
        }
    }
    if (!active) {       // idle lanes: a cone that contributes nothing
#pragma unroll
        for (int a = 0; a < D; ++a) {
            h[a] = a == 0 ? R(1) : R(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) G[a][i] = R(0);
        }
    }
    const bool act = active;
    const bool any_bad = quad_max((R)bad) > R(0);

    // ---- objective (replicated): P = 2 diag(w), q = [-2 w r, 0]
    R Pd[NV], qv[NV], x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const R wi = (R)w[(size_t)bb * NV + i];
        Pd[i] = R(2) * wi;
        qv[i] = i < M_ ? -R(2) * wi * (R)r[(size_t)bb * M_ + i] : R(0);
    }
    R resx0 = 0, hh = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) resx0 += qv[i] * qv[i];
#pragma unroll
    for (int a = 0; a < D; ++a) hh += h[a] * h[a];
    hh = active ? hh : R(0);
    resx0 = fmax(R(1), qsq(resx0));
    const R resz0 = fmax(R(1), qsq(quad_sum(hh)));

    // ---- initial point: (P + G'G) x = G'h - q,  z = G x - h,  s = -z, shifted into the cone
    ConeLane<R, D> cone;
    double H[NV][NV], rhs[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double ri = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) ri += (double)G[a][i] * (double)h[a];
        rhs[i] = quad_sum(act ? ri : 0.0) - (double)qv[i];
#pragma unroll
        for (int j = 0; j < NV; ++j) if (j <= i) {
            double v = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) v += (double)G[a][i] * (double)G[a][j];
            H[i][j] = quad_sum(act ? v : 0.0) + (i == j ? (double)Pd[i] : 0.0);
        }
    }
    bool okc = chol3<NV>(H);
    chol3_solve<NV>(H, rhs);
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = (R)rhs[i];
    {
        R nrm2 = 0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            R v = -h[a];
#pragma unroll
            for (int i = 0; i < NV; ++i) v += G[a][i] * x[i];
            cone.z[a] = v;
            cone.s[a] = -v;
            nrm2 += v * v;
        }
        const R nrm = fmax(qsq(quad_sum(active ? nrm2 : R(0))), R(1));
        R ns = 0;
#pragma unroll
        for (int a = 1; a < D; ++a) ns += cone.s[a] * cone.s[a];
        ns = qsq(ns);
        const R ts = quad_max(active ? ns - cone.s[0] : R(-1e30));
        const R tz = quad_max(active ? ns - cone.z[0] : R(-1e30));     // |z1| = |s1|
        if (ts >= R(-1e-8) * nrm) cone.s[0] += R(1) + ts;
        if (tz >= R(-1e-8) * nrm) cone.z[0] += R(1) + tz;
        if (!active) {
#pragma unroll
            for (int a = 0; a < D; ++a) { cone.s[a] = a == 0 ? R(1) : R(0); cone.z[a] = cone.s[a]; }
        }
    }
    cone.init_scaling();

    R xbest[NV], best = R(1e30);
#pragma unroll
    for (int i = 0; i < NV; ++i) xbest[i] = x[i];
    int stall = 0, st_code = BCBF_SOCP_MAXITER, it = 0;
    const R ideg = qdv(R(1), (R)K);
    if (!okc) st_code = BCBF_SOCP_DIVERGED;

    for (it = 0; it <= max_iters && okc; ++it) {
        const R ib2 = qdv(R(1), cone.beta2);
        // residuals
        R gz[NV], rz[D], rzt[D];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            R t = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) t += G[a][i] * cone.z[a];
            gz[i] = quad_sum(active ? t : R(0));
        }
        R f0 = 0, resx = 0, rz2 = 0, gp = 0;
        R rx[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const R px = Pd[i] * x[i];
            f0 += x[i] * (R(0.5) * px + qv[i]);
            rx[i] = px + qv[i] + gz[i];
            resx += rx[i] * rx[i];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
            R v = cone.s[a] - h[a];
#pragma unroll
            for (int i = 0; i < NV; ++i) v += G[a][i] * x[i];
            rz[a] = v;
            rz2 += v * v;
            gp += cone.lam[a] * cone.lam[a];
        }
        cone.minv(rz, rzt, ib2);
        R lrz = 0;
#pragma unroll
        for (int a = 0; a < D; ++a) lrz += cone.lam[a] * rzt[a];
        resx = qsq(resx);
        const R resz = qsq(quad_sum(active ? rz2 : R(0)));
        const R gap = quad_sum(active ? gp : R(0));
        lrz = quad_sum(active ? lrz : R(0));
        const R pcost = f0, dcost = f0 + lrz - gap;
        R relgap = R(1e30);
        if (pcost < R(0)) relgap = qdv(gap, -pcost);
        else if (dcost > R(0)) relgap = qdv(gap, dcost);
        const R pres = qdv(resz, resz0), dres = qdv(resx, resx0);
        if (pres <= QTol<R>::feas() && dres <= QTol<R>::feas() && (gap <= QTol<R>::gap() || relgap <= QTol<R>::gap())) {
            st_code = BCBF_SOCP_OPTIMAL;
            break;
        }
        {
            const R merit = fmax(fmax(pres, dres), fmin(gap, relgap));
            stall = merit < R(0.9) * best ? 0 : stall + 1;
            if (merit < best) {
                best = merit;
#pragma unroll
                for (int i = 0; i < NV; ++i) xbest[i] = x[i];
            }
            if (stall >= 3 && best <= QTol<R>::accept()) break;
        }
        {
            // Certificate of primal infeasibility (z in K*, G'z = 0, h'z < 0; cvxopt's conelp test pinfres): on an
            // infeasible program z runs off along such a ray, so the instance can leave long before the iteration cap
            R hz = 0, gzn = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) hz += h[a] * cone.z[a];
            hz = quad_sum(active ? hz : R(0));
#pragma unroll
            for (int i = 0; i < NV; ++i) gzn += gz[i] * gz[i];
            if (hz < R(0) && qsq(gzn) <= QTol<R>::infeas() * resx0 * (-hz)) { st_code = BCBF_SOCP_DIVERGED; break; }
        }
        R xmax = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) xmax = fmax(xmax, fabs(x[i]));
        if (!(gap + resx + resz < R(1e30)) || !(xmax < R(1e12))) { st_code = BCBF_SOCP_DIVERGED; break; }
        if (it == max_iters) break;

        // Gt = M^-1 G (own cone), H = P + sum_k Gt'Gt  (fp64)
        R Gt[D][NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            R col[D], out[D];
#pragma unroll
            for (int a = 0; a < D; ++a) col[a] = G[a][i];
            cone.minv(col, out, ib2);
#pragma unroll
            for (int a = 0; a < D; ++a) Gt[a][i] = out[a];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) if (j <= i) {
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < D; ++a) v += (double)Gt[a][i] * (double)Gt[a][j];
                H[i][j] = quad_sum(act ? v : 0.0) + (i == j ? (double)Pd[i] : 0.0);
            }
        if (!chol3<NV>(H)) { st_code = BCBF_SOCP_DIVERGED; break; }

        R lsq[D], corr[D], dx[NV], dst[D], dzt[D];
        ConeLane<R, D>::sprod(cone.lam, cone.lam, lsq);
#pragma unroll
        for (int a = 0; a < D; ++a) corr[a] = R(0);
        const R mu = gap * ideg;
        R sigma = R(0), step = R(1);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            R c[D];
#pragma unroll
            for (int a = 0; a < D; ++a) c[a] = -lsq[a] - corr[a];
            c[0] += sigma * mu;
            cone.sinv(c);
            double rh3[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                double t = 0.0;
#pragma unroll
                for (int a = 0; a < D; ++a) t += (double)Gt[a][i] * (double)(rzt[a] + c[a]);
                rh3[i] = -(double)rx[i] - quad_sum(act ? t : 0.0);
            }
            chol3_solve<NV>(H, rh3);
#pragma unroll
            for (int i = 0; i < NV; ++i) dx[i] = (R)rh3[i];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                R t = rzt[a];
#pragma unroll
                for (int i = 0; i < NV; ++i) t += Gt[a][i] * dx[i];
                dzt[a] = t + c[a];
                dst[a] = -t;
            }
            R dsdz = 0;
            if (pass == 0) {
                ConeLane<R, D>::sprod(dst, dzt, corr);
#pragma unroll
                for (int a = 0; a < D; ++a) dsdz += dst[a] * dzt[a];
                dsdz = quad_sum(active ? dsdz : R(0));
            }
            const R tloc = active ? fmax(cone.scaled_max_step(dst), cone.scaled_max_step(dzt)) : R(-1e30);
            const R tm = fmax(R(0), quad_max(tloc));
            if (tm == R(0)) step = R(1);
            else step = pass == 0 ? fmin(R(1), qdv(R(1), tm)) : fmin(R(1), qdv(R(0.99), tm));
            if (pass == 0) {
                const R sg = fmin(R(1), fmax(R(0), R(1) - step + qdv(dsdz, gap) * step * step));
                sigma = sg * sg * sg;
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] += step * dx[i];
        if (active) cone.advance(dst, dzt, step, ib2);
    }
    if (st_code != BCBF_SOCP_OPTIMAL && best <= QTol<R>::accept()) {
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = xbest[i];
        st_code = BCBF_SOCP_OPTIMAL;
    }
    if (any_bad) st_code = BCBF_SOCP_BADCONE;
    if (inst_ok && k == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) y[(size_t)b * NV + i] = (T)x[i];
        status[b] = st_code;
        if (iters) iters[b] = it;
        if (UNI && task.dt > T(0)) {             // plant step with y = [u0, u1, relax] as stored (rounded to T)
            T* xs = task.x + (size_t)b * 3;
            const T x0 = xs[0], x1 = xs[1], th = xs[2];
            T n0 = x0, n1 = x1, n2 = th, u0 = T(0), u1 = T(0);
            if (st_code == BCBF_SOCP_OPTIMAL) {  // An instance whose program was not solved keeps its state: the reference raises
                u0 = (T)x[0]; u1 = (T)x[1];      // there (unicycle_move_to_pose.py:954-964), a batch freezes the instance and
                n0 = x0 + cos(th) * u0 * task.dt;                 // reports it in status[]
                n1 = x1 + sin(th) * u0 * task.dt;
                n2 = th + u1 / task.L_true * task.dt;
                xs[0] = n0; xs[1] = n1; xs[2] = n2;
            }
            unicycle_observe<T>(task, b, x0, x1, th, n0, n1, n2, u0, u1);
        }
    }
}

template <typename T, bool FROM_TERMS>
static int launch_quad(const T* w, const T* r, const T* cones_in, const T* relax_mask, const T* rho, const T* Mk,
                       const T* Bk, const T* A, const T* grad, const T* cst, const T* sign, const T* fhat,
                       const T* ghat, int n, T* terms_out, T* cones_out, int* cstatus, T* y, int* status, int* iters,
                       int Bt, int K, int m, int max_iters, void* stream, const UnicycleTask<T>* task = nullptr) {
    if (Bt <= 0) return BCBF_OK;
    if (!w || !r || !relax_mask || !rho || !y || !status) return BCBF_EINVAL;
    if (K < 1 || K > 4 || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    if (FROM_TERMS && task != nullptr) {
        if (!Mk || !Bk || !A || !sign || n != 3 || m != 2 || K != 1 + task->Kob || !task->x || !task->plan ||
            !task->dot_plan || !task->Kp || (task->Kob > 0 && (!task->centers || !task->radii || !task->tw || !task->gammas)))
            return BCBF_EINVAL;
        if (task->grad != nullptr && (!task->cst || !task->fhat || !task->ghat)) return BCBF_EINVAL;
    } else if (FROM_TERMS) {
        if (!Mk || !Bk || !A || !grad || !cst || !sign || !fhat || !ghat || n < 1 || n > BCBF_MAX_STATE_DIM) return BCBF_EINVAL;
    } else if (!cones_in) return BCBF_EINVAL;
    if (max_iters <= 0) max_iters = 100;
    const int threads = 64;     // one wave per workgroup: the 4*Bt lanes spread over as many CUs as possible
    const long lanes = (long)Bt * 4;
    dim3 grid((unsigned)((lanes + threads - 1) / threads)), block(threads);
    hipStream_t st = (hipStream_t)stream;
    if (FROM_TERMS && task != nullptr) {
        hipLaunchKernelGGL((socp_quad_kernel<T, 2, FROM_TERMS, FROM_TERMS>), grid, block, 0, st, w, r, cones_in, relax_mask,
                           rho, Mk, Bk, A, grad, cst, sign, fhat, ghat, n, terms_out, cones_out, cstatus, y, status, iters,
                           Bt, K, max_iters, *task);
        return check_launch("unicycle_socp");
    }
#define BCBF_Q(MM) hipLaunchKernelGGL((socp_quad_kernel<T, MM, FROM_TERMS, false>), grid, block, 0, st, w, r, cones_in, relax_mask, rho, Mk, Bk, A, grad, cst, sign, fhat, ghat, n, terms_out, cones_out, cstatus, y, status, iters, Bt, K, max_iters, UnicycleTask<T>{})
    switch (m) {
        case 1: BCBF_Q(1); break;
        case 2: BCBF_Q(2); break;
        case 3: BCBF_Q(3); break;
        default: return BCBF_EINVAL;
    }
#undef BCBF_Q
    return check_launch("socp_quad");
}

// internal entry of the fused unicycle step (control_step.hip)
template <typename T>
int launch_unicycle_socp(const T* Mk, const T* Bk, const T* A, const T* sign, const T* w, const T* r, const T* relax_mask,
                         const T* rho, T* cones, int* cstatus, T* y, int* status, int* iters, int Bt, int max_iters,
                         const UnicycleTask<T>& task, void* stream) {
    return launch_quad<T, true>(w, r, nullptr, relax_mask, rho, Mk, Bk, A, nullptr, nullptr, sign, nullptr, nullptr, 3,
                                nullptr, cones, cstatus, y, status, iters, Bt, 1 + task.Kob, 2, max_iters, stream, &task);
}
template int launch_unicycle_socp<float>(const float*, const float*, const float*, const float*, const float*, const float*,
                                         const float*, const float*, float*, int*, float*, int*, int*, int, int,
                                         const UnicycleTask<float>&, void*);
template int launch_unicycle_socp<double>(const double*, const double*, const double*, const double*, const double*,
                                          const double*, const double*, const double*, double*, int*, double*, int*, int*,
                                          int, int, const UnicycleTask<double>&, void*);

}  // namespace bcbf

extern "C" {
int bcbf_socp_f32(const float* w, const float* r, const float* cones, const float* relax_mask, const float* rho,
                  float* y, int* status, int* iters, int Bt, int K, int m, int max_iters, void* stream) {
    return bcbf::launch_quad<float, false>(w, r, cones, relax_mask, rho, nullptr, nullptr, nullptr, nullptr, nullptr,
                                           nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, y, status, iters,
                                           Bt, K, m, max_iters, stream);
}
int bcbf_socp_f64(const double* w, const double* r, const double* cones, const double* relax_mask, const double* rho,
                  double* y, int* status, int* iters, int Bt, int K, int m, int max_iters, void* stream) {
    return bcbf::launch_quad<double, false>(w, r, cones, relax_mask, rho, nullptr, nullptr, nullptr, nullptr, nullptr,
                                            nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, y, status, iters,
                                            Bt, K, m, max_iters, stream);
}
int bcbf_cbc_socp_f32(const float* Mk, const float* Bk, const float* A, const float* grad, const float* cst,
                      const float* sign, const float* fhat, const float* ghat, const float* w, const float* r,
                      const float* relax_mask, const float* rho, float* terms, float* cones, int* cstatus,
                      float* y, int* status, int* iters, int Bt, int K, int n, int m, int max_iters, void* stream) {
    return bcbf::launch_quad<float, true>(w, r, nullptr, relax_mask, rho, Mk, Bk, A, grad, cst, sign, fhat, ghat, n,
                                          terms, cones, cstatus, y, status, iters, Bt, K, m, max_iters, stream);
}
int bcbf_cbc_socp_f64(const double* Mk, const double* Bk, const double* A, const double* grad, const double* cst,
                      const double* sign, const double* fhat, const double* ghat, const double* w, const double* r,
                      const double* relax_mask, const double* rho, double* terms, double* cones, int* cstatus,
                      double* y, int* status, int* iters, int Bt, int K, int n, int m, int max_iters, void* stream) {
    return bcbf::launch_quad<double, true>(w, r, nullptr, relax_mask, rho, Mk, Bk, A, grad, cst, sign, fhat, ghat, n,
                                           terms, cones, cstatus, y, status, iters, Bt, K, m, max_iters, stream);
}
}
