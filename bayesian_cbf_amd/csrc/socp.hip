// K9 + K10: batched primal-dual interior-point solver for the per-step cone program, one lane per
// instance, everything in registers (fp64).  Algorithm: cvxopt `coneqp` (Vandenberghe 2010) --
// Mehrotra predictor-corrector, Nesterov-Todd scaling, step 0.99, sigma = (1-step)^3 -- with the
// iterates kept in scaled coordinates and the scaling of each second-order cone kept as an
// accumulated product M = W_1 W_2 ... (J-orthogonal up to beta: M J M' = beta^2 J, so
// M^-1 = J M' J / beta^2 needs no storage).  The KKT system reduces to an nv x nv SPD solve.
#include "bcbf_common.h"

namespace bcbf {

constexpr double IPM_STEP = 0.99;
constexpr double IPM_ABSTOL = 1e-9, IPM_RELTOL = 1e-9, IPM_FEASTOL = 1e-9;

// Generic, size-templated solver.  FIXED = true: every bound is the template constant (loops unroll,
// arrays live in registers).  FIXED = false: runtime sizes up to the template maxima.
template <int MAXNV, int MAXL, int MAXNQ, int MAXD, bool FIXED>
struct ConeQP {
    static constexpr int MAXK = MAXL + MAXNQ * MAXD;
    int nv, l, nq, qd[MAXNQ], qo[MAXNQ], K;

    __device__ inline void dims_fixed() {
        nv = MAXNV; l = MAXL; nq = MAXNQ; K = MAXK;
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) { qd[k] = MAXD; qo[k] = MAXL + k * MAXD; }
    }
    __device__ inline void dims_runtime(int nv_, int l_, const int* qd_, int nq_) {
        nv = nv_; l = l_; nq = nq_;
        int off = l_;
        for (int k = 0; k < MAXNQ; ++k) { qd[k] = k < nq_ ? qd_[k] : 0; qo[k] = off; off += qd[k]; }
        K = off;
    }
    __device__ inline int NV() const { return FIXED ? MAXNV : nv; }
    __device__ inline int L() const { return FIXED ? MAXL : l; }
    __device__ inline int NQ() const { return FIXED ? MAXNQ : nq; }
    __device__ inline int QD(int k) const { return FIXED ? MAXD : qd[k]; }
    __device__ inline int QO(int k) const { return FIXED ? MAXL + k * MAXD : qo[k]; }
    __device__ inline int KT() const { return FIXED ? MAXK : K; }

    // -min eigenvalue over all cones
    __device__ inline double max_step(const double* x) const {
        double t = -1e300;
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) t = fmax(t, -x[i]);
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            double nn = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) nn += x[o + a] * x[o + a];
            t = fmax(t, sqrt(nn) - x[o]);
        }
        return t;
    }
    __device__ inline void add_e(double* x, double a) const {
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) x[i] += a;
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) x[QO(k)] += a;
    }
    // out = x o y (Jordan product)
    __device__ inline void sprod(const double* x, const double* y, double* out) const {
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) out[i] = x[i] * y[i];
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            double d = 0;
#pragma unroll
            for (int a = 0; a < MAXD; ++a) if (a < QD(k)) d += x[o + a] * y[o + a];
            out[o] = d;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) out[o + a] = x[o] * y[o + a] + y[o] * x[o + a];
        }
    }
    // solve lam o y = x, in place on x
    __device__ inline void sinv(const double* lam, double* x) const {
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) x[i] /= lam[i];
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            double det = lam[o] * lam[o], lx = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) { det -= lam[o + a] * lam[o + a]; lx += lam[o + a] * x[o + a]; }
            const double y0 = (lam[o] * x[o] - lx) / det;
            x[o] = y0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) x[o + a] = (x[o + a] - y0 * lam[o + a]) / lam[o];
        }
    }
    // x := P(lam^-1/2) x, then return max_step(x)  (largest t with lam + x/t ... see oracle/socp.py:_scale2)
    __device__ inline double scaled_max_step(const double* lam, const double* x) const {
        double t = -1e300;
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) t = fmax(t, -x[i] / lam[i]);
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            double det = lam[o] * lam[o];
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) det -= lam[o + a] * lam[o + a];
            const double nrm = sqrt(det), inrm = 1.0 / nrm;
            const double lb0 = lam[o] * inrm;
            double lx = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) lx += lam[o + a] * inrm * x[o + a];
            const double y0 = (lb0 * x[o] - lx) * inrm;
            const double coef = (-x[o] + lx / (1.0 + lb0));
            double nn = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) {
                const double ya = (x[o + a] + coef * lam[o + a] * inrm) * inrm;
                nn += ya * ya;
            }
            t = fmax(t, sqrt(nn) - y0);
        }
        return t;
    }
    // NT scaling of one second-order cone block: returns beta, fills w (w'Jw = 1); lam_out = W z
    __device__ inline double nt_block(const double* s, const double* z, int d, double* w, double* lam_out) const {
        double sj = s[0] * s[0], zj = z[0] * z[0], sz = s[0] * z[0];
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) { sj -= s[a] * s[a]; zj -= z[a] * z[a]; sz += s[a] * z[a]; }
        const double sn = sqrt(sj), zn = sqrt(zj);
        const double gamma = sqrt((1.0 + sz / (sn * zn)) * 0.5);
        const double ig = 1.0 / (2.0 * gamma);
        w[0] = (s[0] / sn + z[0] / zn) * ig;
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) w[a] = (s[a] / sn - z[a] / zn) * ig;
        const double beta = sqrt(sn / zn);
        // lam = beta * Wbar z,  Wbar = [[w0, w1'],[w1, I + w1 w1'/(1+w0)]]
        double w1z = 0;
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) w1z += w[a] * z[a];
        lam_out[0] = beta * (w[0] * z[0] + w1z);
        const double cf = z[0] + w1z / (1.0 + w[0]);
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) lam_out[a] = beta * (z[a] + cf * w[a]);
        return beta;
    }

    // Solve.  P (symmetric, row-major MAXNV stride), q, G[K][MAXNV stride], h.  Returns status.
    __device__ int solve(const double (*P)[MAXNV], const double* q, const double (*G)[MAXNV], const double* h,
                         double* x, int max_iters, int* iters_out) {
        double lam[MAXK], M[MAXNQ][MAXD][MAXD], beta2[MAXNQ], dl[MAXL > 0 ? MAXL : 1];
        double s[MAXK], z[MAXK];
        double resx0 = 0, resz0 = 0;
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) if (i < NV()) resx0 += q[i] * q[i];
#pragma unroll
        for (int a = 0; a < MAXK; ++a) if (a < KT()) resz0 += h[a] * h[a];
        resx0 = fmax(1.0, sqrt(resx0));
        resz0 = fmax(1.0, sqrt(resz0));

        // ---- initial point: (P + G'G) x = G'h - q, z = G x - h, s = -z, shifted into the cone
        double H[MAXNV][MAXNV], rhs[MAXNV];
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
            double r = -q[i];
#pragma unroll
            for (int a = 0; a < MAXK; ++a) if (a < KT()) r += G[a][i] * h[a];
            rhs[i] = r;
#pragma unroll
            for (int j = 0; j < MAXNV; ++j) if (j <= i) {
                double v = P[i][j];
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) v += G[a][i] * G[a][j];
                H[i][j] = v;
            }
        }
        if (!chol_solve(H, rhs)) { *iters_out = 0; return BCBF_SOCP_DIVERGED; }
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) if (i < NV()) x[i] = rhs[i];
        double nrm = 0;
#pragma unroll
        for (int a = 0; a < MAXK; ++a) if (a < KT()) {
            double v = -h[a];
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) v += G[a][i] * x[i];
            z[a] = v; s[a] = -v; nrm += v * v;
        }
        nrm = fmax(sqrt(nrm), 1.0);
        double ts = max_step(s);
        if (ts >= -1e-8 * nrm) add_e(s, 1.0 + ts);
        double tz = max_step(z);
        if (tz >= -1e-8 * nrm) add_e(z, 1.0 + tz);
        // ---- initial scaling
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) { dl[i] = sqrt(s[i] / z[i]); lam[i] = sqrt(s[i] * z[i]); }
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k), d = QD(k);
            double w[MAXD];
            const double beta = nt_block(&s[o], &z[o], d, w, &lam[o]);
            beta2[k] = beta * beta;
#pragma unroll
            for (int a = 0; a < MAXD; ++a)
#pragma unroll
                for (int c = 0; c < MAXD; ++c) if (a < d && c < d) {
                    double v;
                    if (a == 0) v = w[c];
                    else if (c == 0) v = w[a];
                    else v = (a == c ? 1.0 : 0.0) + w[a] * w[c] / (1.0 + w[0]);
                    M[k][a][c] = beta * v;
                }
        }

        int status = BCBF_SOCP_MAXITER;
        int it = 0;
        for (it = 0; it <= max_iters; ++it) {
            // unscaled s = M lam, z = M^-T lam = J M J lam / beta^2
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) { s[i] = dl[i] * lam[i]; z[i] = lam[i] / dl[i]; }
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
                    double vs = 0, vz = 0;
#pragma unroll
                    for (int c = 0; c < MAXD; ++c) if (c < d) {
                        vs += M[k][a][c] * lam[o + c];
                        vz += M[k][a][c] * (c == 0 ? lam[o + c] : -lam[o + c]);
                    }
                    s[o + a] = vs;
                    z[o + a] = (a == 0 ? vz : -vz) / beta2[k];
                }
            }
            double rx[MAXNV], rz[MAXK], rzt[MAXK];
            double f0 = 0, resx = 0, resz = 0, gap = 0;
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
                double px = 0, gz = 0;
#pragma unroll
                for (int j = 0; j < MAXNV; ++j) if (j < NV()) px += P[i][j] * x[j];
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) gz += G[a][i] * z[a];
                f0 += x[i] * (0.5 * px + q[i]);
                rx[i] = px + q[i] + gz;
                resx += rx[i] * rx[i];
            }
#pragma unroll
            for (int a = 0; a < MAXK; ++a) if (a < KT()) {
                double v = s[a] - h[a];
#pragma unroll
                for (int i = 0; i < MAXNV; ++i) if (i < NV()) v += G[a][i] * x[i];
                rz[a] = v;
                resz += v * v;
                gap += lam[a] * lam[a];
            }
            resx = sqrt(resx); resz = sqrt(resz);
            // rzt = M^-1 rz
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) rzt[i] = rz[i] / dl[i];
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
                    double v = 0;
#pragma unroll
                    for (int c = 0; c < MAXD; ++c) if (c < d) v += M[k][c][a] * (c == 0 ? rz[o + c] : -rz[o + c]);
                    rzt[o + a] = (a == 0 ? v : -v) / beta2[k];
                }
            }
            double lrz = 0;
#pragma unroll
            for (int a = 0; a < MAXK; ++a) if (a < KT()) lrz += lam[a] * rzt[a];
            const double pcost = f0, dcost = f0 + lrz - gap;
            double relgap = 1e300;
            if (pcost < 0.0) relgap = gap / -pcost;
            else if (dcost > 0.0) relgap = gap / dcost;
            const double pres = resz / resz0, dres = resx / resx0;
            if (pres <= IPM_FEASTOL && dres <= IPM_FEASTOL && (gap <= IPM_ABSTOL || relgap <= IPM_RELTOL)) {
                status = BCBF_SOCP_OPTIMAL;
                break;
            }
            double xmax = 0;
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) xmax = fmax(xmax, fabs(x[i]));
            if (!(gap + resx + resz < 1e300) || !(xmax < 1e12)) { status = BCBF_SOCP_DIVERGED; break; }
            if (it == max_iters) break;

            // ---- Gt = M^-1 G,  H = P + Gt'Gt
            double Gt[MAXK][MAXNV];
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) {
#pragma unroll
                for (int v = 0; v < MAXNV; ++v) if (v < NV()) Gt[i][v] = G[i][v] / dl[i];
            }
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
#pragma unroll
                    for (int v = 0; v < MAXNV; ++v) if (v < NV()) {
                        double acc = 0;
#pragma unroll
                        for (int c = 0; c < MAXD; ++c) if (c < d) acc += M[k][c][a] * (c == 0 ? G[o + c][v] : -G[o + c][v]);
                        Gt[o + a][v] = (a == 0 ? acc : -acc) / beta2[k];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
#pragma unroll
                for (int j = 0; j < MAXNV; ++j) if (j <= i) {
                    double v = P[i][j];
#pragma unroll
                    for (int a = 0; a < MAXK; ++a) if (a < KT()) v += Gt[a][i] * Gt[a][j];
                    H[i][j] = v;
                }
            }
            if (!chol_factor(H)) { status = BCBF_SOCP_DIVERGED; break; }

            double lsq[MAXK], dsdz_o[MAXK], dx[MAXNV], dst[MAXK], dzt[MAXK];
            sprod(lam, lam, lsq);
            int deg = L() + NQ();
            const double mu = gap / deg;
            double sigma = 0.0, step = 1.0, dsdz = 0.0;
#pragma unroll
            for (int a = 0; a < MAXK; ++a) dsdz_o[a] = 0.0;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                double c[MAXK];
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) c[a] = -lsq[a] - dsdz_o[a];
                add_e(c, sigma * mu);
                sinv(lam, c);
                // (P + Gt'Gt) dx = -rx + Gt'(-rzt - c)
#pragma unroll
                for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
                    double r = -rx[i];
#pragma unroll
                    for (int a = 0; a < MAXK; ++a) if (a < KT()) r -= Gt[a][i] * (rzt[a] + c[a]);
                    dx[i] = r;
                }
                chol_backsolve(H, dx);
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) {
                    double t = rzt[a];
#pragma unroll
                    for (int i = 0; i < MAXNV; ++i) if (i < NV()) t += Gt[a][i] * dx[i];
                    dzt[a] = t + c[a];
                    dst[a] = -t;
                }
                if (pass == 0) {
                    sprod(dst, dzt, dsdz_o);
                    dsdz = 0;
#pragma unroll
                    for (int a = 0; a < MAXK; ++a) if (a < KT()) dsdz += dst[a] * dzt[a];
                }
                const double t1 = scaled_max_step(lam, dst), t2 = scaled_max_step(lam, dzt);
                const double tm = fmax(0.0, fmax(t1, t2));
                if (tm == 0.0) step = 1.0;
                else step = pass == 0 ? fmin(1.0, 1.0 / tm) : fmin(1.0, IPM_STEP / tm);
                if (pass == 0) {
                    double sg = fmin(1.0, fmax(0.0, 1.0 - step + dsdz / gap * step * step));
                    sigma = sg * sg * sg;
                }
            }
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) x[i] += step * dx[i];
            // ---- update scaled iterates and the accumulated scaling
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) {
                const double st = lam[i] + step * dst[i], zt = lam[i] + step * dzt[i];
                dl[i] *= sqrt(st / zt);
                lam[i] = sqrt(st * zt);
            }
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
                double st[MAXD], zt[MAXD], w[MAXD];
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) { st[a] = lam[o + a] + step * dst[o + a]; zt[a] = lam[o + a] + step * dzt[o + a]; }
                const double beta = nt_block(st, zt, d, w, &lam[o]);
                beta2[k] *= beta * beta;
                // M <- M * (beta * Wbar(w)):   row a of M times Wbar
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
                    double m0 = M[k][a][0], mw = 0;
#pragma unroll
                    for (int c = 1; c < MAXD; ++c) if (c < d) mw += M[k][a][c] * w[c];
                    const double n0 = m0 * w[0] + mw;
                    const double cf = m0 + mw / (1.0 + w[0]);
                    M[k][a][0] = beta * n0;
#pragma unroll
                    for (int c = 1; c < MAXD; ++c) if (c < d) M[k][a][c] = beta * (M[k][a][c] + cf * w[c]);
                }
            }
        }
        *iters_out = it;
        return status;
    }

    // dense SPD helpers on the lower triangle of H (NV x NV)
    __device__ inline bool chol_factor(double (*H)[MAXNV]) const {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MAXNV; ++j) if (j < NV()) {
            double d = H[j][j];
#pragma unroll
            for (int k = 0; k < MAXNV; ++k) if (k < j) d -= H[j][k] * H[j][k];
            if (!(d > 0.0)) { ok = false; d = 1.0; }
            const double ljj = sqrt(d);
            H[j][j] = ljj;
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i > j && i < NV()) {
                double v = H[i][j];
#pragma unroll
                for (int k = 0; k < MAXNV; ++k) if (k < j) v -= H[i][k] * H[j][k];
                H[i][j] = v / ljj;
            }
        }
        return ok;
    }
    __device__ inline void chol_backsolve(const double (*H)[MAXNV], double* b) const {
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
            double v = b[i];
#pragma unroll
            for (int k = 0; k < MAXNV; ++k) if (k < i) v -= H[i][k] * b[k];
            b[i] = v / H[i][i];
        }
#pragma unroll
        for (int ii = 0; ii < MAXNV; ++ii) {
            const int i = MAXNV - 1 - ii;
            if (i < NV()) {
                double v = b[i];
#pragma unroll
                for (int k = 0; k < MAXNV; ++k) if (k > i && k < NV()) v -= H[k][i] * b[k];
                b[i] = v / H[i][i];
            }
        }
    }
    __device__ inline bool chol_solve(double (*H)[MAXNV], double* b) const {
        const bool ok = chol_factor(H);
        chol_backsolve(H, b);
        return ok;
    }
};

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
    using Solver = ConeQP<NV, 0, KC, D, true>;
    Solver S;
    S.dims_fixed();
    double P[NV][NV], q[NV], G[KC * D][NV], h[KC * D], x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int j = 0; j < NV; ++j) P[i][j] = 0.0;
        P[i][i] = 2.0 * (double)w[(size_t)b * NV + i];
        q[i] = i < M_ ? -2.0 * (double)w[(size_t)b * NV + i] * (double)r[(size_t)b * M_ + i] : 0.0;
    }
    const double rh = (double)rho[b];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const T* cn = cones + ((size_t)b * KC + k) * Q;
        const T* cA = cn;                       // [(M+1)][M]
        const T* cb = cn + (M_ + 1) * M_;       // [M+1]
        const T* cc = cb + (M_ + 1);            // [M]
        const T cd = cc[M_];
#pragma unroll
        for (int i = 0; i < M_; ++i) G[k * D][i] = -(double)cc[i];
        G[k * D][M_] = -(double)relax_mask[k];
        h[k * D] = (double)cd;
#pragma unroll
        for (int a = 0; a < M_ + 1; ++a) {
#pragma unroll
            for (int i = 0; i < M_; ++i) G[k * D + 1 + a][i] = -rh * (double)cA[a * M_ + i];
            G[k * D + 1 + a][M_] = 0.0;
            h[k * D + 1 + a] = rh * (double)cb[a];
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
    using Solver = ConeQP<GNV, GL, GNQ, GD, false>;
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
