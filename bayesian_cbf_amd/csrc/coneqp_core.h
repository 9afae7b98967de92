// Size-templated primal-dual interior-point solver for small cone QPs (see socp.hip).
// Plain C++ (no HIP types) so that the same code can be compiled for the host when debugging.
#pragma once
#include <math.h>
#ifndef BCBF_HD
#define BCBF_HD __host__ __device__
#endif
#ifndef BCBF_SOCP_OPTIMAL
#define BCBF_SOCP_OPTIMAL 0
#define BCBF_SOCP_MAXITER 1
#define BCBF_SOCP_DIVERGED 2
#define BCBF_SOCP_BADCONE 3
#endif

namespace bcbf {

template <typename R> struct Tol;
template <> struct Tol<double> {
    static BCBF_HD inline double feas() { return 1e-9; }
    static BCBF_HD inline double abs_() { return 1e-9; }
    static BCBF_HD inline double rel() { return 1e-9; }
    static BCBF_HD inline double accept() { return 1e-7; }
};
template <> struct Tol<float> {     // fp32 iterates: residuals bottom out near 1e-6 relative
    static BCBF_HD inline float feas() { return 2e-5f; }
    static BCBF_HD inline float abs_() { return 1e-6f; }
    static BCBF_HD inline float rel() { return 1e-6f; }
    static BCBF_HD inline float accept() { return 2e-4f; }   // best-iterate fallback (see solve)
};

// Generic, size-templated solver.  FIXED = true: every bound is the template constant (loops unroll,
// arrays live in registers).  FIXED = false: runtime sizes up to the template maxima.
template <typename R, int MAXNV, int MAXL, int MAXNQ, int MAXD, bool FIXED>
struct ConeQP {
    // a/b and sqrt: single-instruction forms for fp32 (v_rcp_f32 / v_sqrt_f32, ~1 ulp), IEEE for fp64
#if defined(__HIP_DEVICE_COMPILE__)
    static BCBF_HD inline float dv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static BCBF_HD inline float sq(float a) { return __builtin_amdgcn_sqrtf(a); }
#else
    static BCBF_HD inline float dv(float a, float b) { return a / b; }
    static BCBF_HD inline float sq(float a) { return __builtin_sqrtf(a); }
#endif
    static BCBF_HD inline double dv(double a, double b) { return a / b; }
    static BCBF_HD inline double sq(double a) { return __builtin_sqrt(a); }
    static constexpr int MAXK = MAXL + MAXNQ * MAXD;
    int nv, l, nq, qd[MAXNQ], qo[MAXNQ], K;

    BCBF_HD inline void dims_fixed() {
        nv = MAXNV; l = MAXL; nq = MAXNQ; K = MAXK;
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) { qd[k] = MAXD; qo[k] = MAXL + k * MAXD; }
    }
    BCBF_HD inline void dims_runtime(int nv_, int l_, const int* qd_, int nq_) {
        nv = nv_; l = l_; nq = nq_;
        int off = l_;
        for (int k = 0; k < MAXNQ; ++k) { qd[k] = k < nq_ ? qd_[k] : 0; qo[k] = off; off += qd[k]; }
        K = off;
    }
    BCBF_HD inline int NV() const { return FIXED ? MAXNV : nv; }
    BCBF_HD inline int L() const { return FIXED ? MAXL : l; }
    BCBF_HD inline int NQ() const { return FIXED ? MAXNQ : nq; }
    BCBF_HD inline int QD(int k) const { return FIXED ? MAXD : qd[k]; }
    BCBF_HD inline int QO(int k) const { return FIXED ? MAXL + k * MAXD : qo[k]; }
    BCBF_HD inline int KT() const { return FIXED ? MAXK : K; }

    // -min eigenvalue over all cones
    BCBF_HD inline R max_step(const R* x) const {
        R t = R(-1e30);
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) t = fmax(t, -x[i]);
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            R nn = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) nn += x[o + a] * x[o + a];
            t = fmax(t, sq(nn) - x[o]);
        }
        return t;
    }
    BCBF_HD inline void add_e(R* x, R a) const {
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) x[i] += a;
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) x[QO(k)] += a;
    }
    // out = x o y (Jordan product)
    BCBF_HD inline void sprod(const R* x, const R* y, R* out) const {
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) out[i] = x[i] * y[i];
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            R d = 0;
#pragma unroll
            for (int a = 0; a < MAXD; ++a) if (a < QD(k)) d += x[o + a] * y[o + a];
            out[o] = d;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) out[o + a] = x[o] * y[o + a] + y[o] * x[o + a];
        }
    }
    // solve lam o y = x, in place on x
    BCBF_HD inline void sinv(const R* lam, R* x) const {
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) x[i] = dv(x[i], lam[i]);
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            R det = lam[o] * lam[o], lx = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) { det -= lam[o + a] * lam[o + a]; lx += lam[o + a] * x[o + a]; }
            const R y0 = dv(lam[o] * x[o] - lx, det);
            const R il0 = dv(R(1.0), lam[o]);
            x[o] = y0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) x[o + a] = (x[o + a] - y0 * lam[o + a]) * il0;
        }
    }
    // x := P(lam^-1/2) x, then return max_step(x)  (largest t with lam + x/t ... see oracle/socp.py:_scale2)
    BCBF_HD inline R scaled_max_step(const R* lam, const R* x) const {
        R t = R(-1e30);
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) t = fmax(t, -dv(x[i], lam[i]));
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k);
            R det = lam[o] * lam[o];
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) det -= lam[o + a] * lam[o + a];
            const R nrm = sq(det), inrm = dv(R(1.0), nrm);
            const R lb0 = lam[o] * inrm;
            R lx = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) lx += lam[o + a] * inrm * x[o + a];
            const R y0 = (lb0 * x[o] - lx) * inrm;
            const R coef = (-x[o] + dv(lx, R(1.0) + lb0));
            R nn = 0;
#pragma unroll
            for (int a = 1; a < MAXD; ++a) if (a < QD(k)) {
                const R ya = (x[o + a] + coef * lam[o + a] * inrm) * inrm;
                nn += ya * ya;
            }
            t = fmax(t, sq(nn) - y0);
        }
        return t;
    }
    // NT scaling of one second-order cone block: returns beta, fills w (w'Jw = 1); lam_out = W z
    BCBF_HD inline R nt_block(const R* s, const R* z, int d, R* w, R* lam_out) const {
        R sj = s[0] * s[0], zj = z[0] * z[0], sz = s[0] * z[0];
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) { sj -= s[a] * s[a]; zj -= z[a] * z[a]; sz += s[a] * z[a]; }
        const R sn = sq(sj), zn = sq(zj);
        const R isn = dv(R(1.0), sn), izn = dv(R(1.0), zn);
        const R gamma = sq((R(1.0) + sz * isn * izn) * R(0.5));
        const R ig = dv(R(0.5), gamma);
        w[0] = (s[0] * isn + z[0] * izn) * ig;
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) w[a] = (s[a] * isn - z[a] * izn) * ig;
        const R beta = sq(sn * izn);
        // lam = beta * Wbar z,  Wbar = [[w0, w1'],[w1, I + w1 w1'/(1+w0)]]
        R w1z = 0;
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) w1z += w[a] * z[a];
        lam_out[0] = beta * (w[0] * z[0] + w1z);
        const R cf = z[0] + dv(w1z, R(1.0) + w[0]);
#pragma unroll
        for (int a = 1; a < MAXD; ++a) if (a < d) lam_out[a] = beta * (z[a] + cf * w[a]);
        return beta;
    }

    // Solve.  P (symmetric, row-major MAXNV stride), q, G[K][MAXNV stride], h.  Returns status.
    BCBF_HD int solve(const R (*P)[MAXNV], const R* q, const R (*G)[MAXNV], const R* h,
                         R* x, int max_iters, int* iters_out) {
        R lam[MAXK], M[MAXNQ][MAXD][MAXD], beta2[MAXNQ], dl[MAXL > 0 ? MAXL : 1];
        R s[MAXK], z[MAXK];
        R resx0 = 0, resz0 = 0;
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) if (i < NV()) resx0 += q[i] * q[i];
#pragma unroll
        for (int a = 0; a < MAXK; ++a) if (a < KT()) resz0 += h[a] * h[a];
        resx0 = fmax(R(1.0), sq(resx0));
        resz0 = fmax(R(1.0), sq(resz0));

        // ---- initial point: (P + G'G) x = G'h - q, z = G x - h, s = -z, shifted into the cone
        // The reduced KKT matrix H = P + Gt'Gt squares the conditioning of the scaled constraints:
        // it is accumulated, factored and solved in fp64 even when the iterates are fp32 (nv x nv work).
        double H[MAXNV][MAXNV], rhs[MAXNV];
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
            double r = -(double)q[i];
#pragma unroll
            for (int a = 0; a < MAXK; ++a) if (a < KT()) r += (double)G[a][i] * (double)h[a];
            rhs[i] = r;
#pragma unroll
            for (int j = 0; j < MAXNV; ++j) if (j <= i) {
                double v = (double)P[i][j];
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) v += (double)G[a][i] * (double)G[a][j];
                H[i][j] = v;
            }
        }
        if (!chol_solve(H, rhs)) { *iters_out = 0; return BCBF_SOCP_DIVERGED; }
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) if (i < NV()) x[i] = (R)rhs[i];
        R nrm = 0;
#pragma unroll
        for (int a = 0; a < MAXK; ++a) if (a < KT()) {
            R v = -h[a];
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) v += G[a][i] * x[i];
            z[a] = v; s[a] = -v; nrm += v * v;
        }
        nrm = fmax(sq(nrm), R(1.0));
        R ts = max_step(s);
        if (ts >= R(-1e-8) * nrm) add_e(s, R(1.0) + ts);
        R tz = max_step(z);
        if (tz >= R(-1e-8) * nrm) add_e(z, R(1.0) + tz);
        // ---- initial scaling
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < L()) { dl[i] = sq(dv(s[i], z[i])); lam[i] = sq(s[i] * z[i]); }
#pragma unroll
        for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
            const int o = QO(k), d = QD(k);
            R w[MAXD];
            const R beta = nt_block(&s[o], &z[o], d, w, &lam[o]);
            beta2[k] = beta * beta;
            const R iw0 = dv(R(1.0), R(1.0) + w[0]);
#pragma unroll
            for (int a = 0; a < MAXD; ++a)
#pragma unroll
                for (int c = 0; c < MAXD; ++c) if (a < d && c < d) {
                    R v;
                    if (a == 0) v = w[c];
                    else if (c == 0) v = w[a];
                    else v = (a == c ? R(1.0) : R(0.0)) + w[a] * w[c] * iw0;
                    M[k][a][c] = beta * v;
                }
        }

        // Best iterate so far by merit = max(pres, dres, min(gap, relgap)): in fp32 the scaling loses
        // accuracy once the gap falls below ~1e-7, so the last iterate is not always the best one.
        R xbest[MAXNV], best = R(1e30);
        int stall = 0;
#pragma unroll
        for (int i = 0; i < MAXNV; ++i) xbest[i] = R(0.0);
        int status = BCBF_SOCP_MAXITER;
        int it = 0;
        for (it = 0; it <= max_iters; ++it) {
            // s, z (unscaled) are carried as accumulated iterates (updated with the step below): used
            // for the residuals only.  Recomputing them as M lam / M^-T lam would inject the
            // conditioning of M into the residuals (fatal in fp32 near convergence).
            R ib2[MAXNQ];
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) ib2[k] = dv(R(1.0), beta2[k]);
            R rx[MAXNV], rz[MAXK], rzt[MAXK];
            R f0 = 0, resx = 0, resz = 0, gap = 0;
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
                R px = 0, gz = 0;
#pragma unroll
                for (int j = 0; j < MAXNV; ++j) if (j < NV()) px += P[i][j] * x[j];
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) gz += G[a][i] * z[a];
                f0 += x[i] * (R(0.5) * px + q[i]);
                rx[i] = px + q[i] + gz;
                resx += rx[i] * rx[i];
            }
#pragma unroll
            for (int a = 0; a < MAXK; ++a) if (a < KT()) {
                R v = s[a] - h[a];
#pragma unroll
                for (int i = 0; i < MAXNV; ++i) if (i < NV()) v += G[a][i] * x[i];
                rz[a] = v;
                resz += v * v;
                gap += lam[a] * lam[a];
            }
            resx = sq(resx); resz = sq(resz);
            // rzt = M^-1 rz
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) rzt[i] = dv(rz[i], dl[i]);
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
                    R v = 0;
#pragma unroll
                    for (int c = 0; c < MAXD; ++c) if (c < d) v += M[k][c][a] * (c == 0 ? rz[o + c] : -rz[o + c]);
                    rzt[o + a] = (a == 0 ? v : -v) * ib2[k];
                }
            }
            R lrz = 0;
#pragma unroll
            for (int a = 0; a < MAXK; ++a) if (a < KT()) lrz += lam[a] * rzt[a];
            const R pcost = f0, dcost = f0 + lrz - gap;
            R relgap = R(1e30);
            if (pcost < R(0.0)) relgap = dv(gap, -pcost);
            else if (dcost > R(0.0)) relgap = dv(gap, dcost);
            const R pres = dv(resz, resz0), dres = dv(resx, resx0);
            if (pres <= Tol<R>::feas() && dres <= Tol<R>::feas() && (gap <= Tol<R>::abs_() || relgap <= Tol<R>::rel())) {
                status = BCBF_SOCP_OPTIMAL;
                break;
            }
            {
                const R merit = fmax(fmax(pres, dres), fmin(gap, relgap));
                stall = merit < R(0.9) * best ? 0 : stall + 1;
                if (merit < best) {
                    best = merit;
#pragma unroll
                    for (int i = 0; i < MAXNV; ++i) if (i < NV()) xbest[i] = x[i];
                }
                // rounding floor reached: no progress for 3 iterations on an acceptable iterate
                if (stall >= 3 && best <= Tol<R>::accept()) break;
            }
            R xmax = 0;
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) xmax = fmax(xmax, fabs(x[i]));
            if (!(gap + resx + resz < R(1e30)) || !(xmax < R(1e12))) { status = BCBF_SOCP_DIVERGED; break; }
            if (it == max_iters) break;

            // ---- Gt = M^-1 G,  H = P + Gt'Gt
            R Gt[MAXK][MAXNV];
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) {
#pragma unroll
                for (int v = 0; v < MAXNV; ++v) if (v < NV()) Gt[i][v] = dv(G[i][v], dl[i]);
            }
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
#pragma unroll
                    for (int v = 0; v < MAXNV; ++v) if (v < NV()) {
                        R acc = 0;
#pragma unroll
                        for (int c = 0; c < MAXD; ++c) if (c < d) acc += M[k][c][a] * (c == 0 ? G[o + c][v] : -G[o + c][v]);
                        Gt[o + a][v] = (a == 0 ? acc : -acc) * ib2[k];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
#pragma unroll
                for (int j = 0; j < MAXNV; ++j) if (j <= i) {
                    double v = (double)P[i][j];
#pragma unroll
                    for (int a = 0; a < MAXK; ++a) if (a < KT()) v += (double)Gt[a][i] * (double)Gt[a][j];
                    H[i][j] = v;
                }
            }
            if (!chol_factor(H)) { status = BCBF_SOCP_DIVERGED; break; }

            R lsq[MAXK], dsdz_o[MAXK], dx[MAXNV], dst[MAXK], dzt[MAXK];
            double dxd[MAXNV];
            sprod(lam, lam, lsq);
            int deg = L() + NQ();
            const R mu = dv(gap, (R)deg);
            R sigma = R(0.0), step = R(1.0), dsdz = R(0.0);
#pragma unroll
            for (int a = 0; a < MAXK; ++a) dsdz_o[a] = R(0.0);
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                R c[MAXK];
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) c[a] = -lsq[a] - dsdz_o[a];
                add_e(c, sigma * mu);
                sinv(lam, c);
                // (P + Gt'Gt) dx = -rx + Gt'(-rzt - c)
#pragma unroll
                for (int i = 0; i < MAXNV; ++i) if (i < NV()) {
                    double r = -(double)rx[i];
#pragma unroll
                    for (int a = 0; a < MAXK; ++a) if (a < KT()) r -= (double)Gt[a][i] * (double)(rzt[a] + c[a]);
                    dxd[i] = r;
                }
                chol_backsolve(H, dxd);
#pragma unroll
                for (int i = 0; i < MAXNV; ++i) if (i < NV()) dx[i] = (R)dxd[i];
#pragma unroll
                for (int a = 0; a < MAXK; ++a) if (a < KT()) {
                    R t = rzt[a];
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
                const R t1 = scaled_max_step(lam, dst), t2 = scaled_max_step(lam, dzt);
                const R tm = fmax(R(0.0), fmax(t1, t2));
                if (tm == R(0.0)) step = R(1.0);
                else step = pass == 0 ? fmin(R(1.0), dv(R(1.0), tm)) : fmin(R(1.0), dv(R(0.99), tm));
                if (pass == 0) {
                    R sg = fmin(R(1.0), fmax(R(0.0), R(1.0) - step + dv(dsdz, gap) * step * step));
                    sigma = sg * sg * sg;
                }
            }
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) x[i] += step * dx[i];
            // ---- unscaled iterates (old scaling): s += step * M dst,  z += step * J M J dzt / beta^2
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) { s[i] += step * dl[i] * dst[i]; z[i] += step * dv(dzt[i], dl[i]); }
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
                    R vs = 0, vz = 0;
#pragma unroll
                    for (int c = 0; c < MAXD; ++c) if (c < d) {
                        vs += M[k][a][c] * dst[o + c];
                        vz += M[k][a][c] * (c == 0 ? dzt[o + c] : -dzt[o + c]);
                    }
                    s[o + a] += step * vs;
                    z[o + a] += step * (a == 0 ? vz : -vz) * ib2[k];
                }
            }
            // ---- update scaled iterates and the accumulated scaling
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < L()) {
                const R st = lam[i] + step * dst[i], zt = lam[i] + step * dzt[i];
                dl[i] *= sq(dv(st, zt));
                lam[i] = sq(st * zt);
            }
#pragma unroll
            for (int k = 0; k < MAXNQ; ++k) if (k < NQ()) {
                const int o = QO(k), d = QD(k);
                R st[MAXD], zt[MAXD], w[MAXD];
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) { st[a] = lam[o + a] + step * dst[o + a]; zt[a] = lam[o + a] + step * dzt[o + a]; }
                const R beta = nt_block(st, zt, d, w, &lam[o]);
                beta2[k] *= beta * beta;
                const R iw0 = dv(R(1.0), R(1.0) + w[0]);
                // M <- M * (beta * Wbar(w)):   row a of M times Wbar
#pragma unroll
                for (int a = 0; a < MAXD; ++a) if (a < d) {
                    R m0 = M[k][a][0], mw = 0;
#pragma unroll
                    for (int c = 1; c < MAXD; ++c) if (c < d) mw += M[k][a][c] * w[c];
                    const R n0 = m0 * w[0] + mw;
                    const R cf = m0 + mw * iw0;
                    M[k][a][0] = beta * n0;
#pragma unroll
                    for (int c = 1; c < MAXD; ++c) if (c < d) M[k][a][c] = beta * (M[k][a][c] + cf * w[c]);
                }
            }
        }
        if (status != BCBF_SOCP_OPTIMAL && best <= Tol<R>::accept()) {
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i < NV()) x[i] = xbest[i];
            status = BCBF_SOCP_OPTIMAL;
        }
        *iters_out = it;
        return status;
    }

    // dense SPD helpers on the lower triangle of H (NV x NV)
    BCBF_HD inline bool chol_factor(double (*H)[MAXNV]) const {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MAXNV; ++j) if (j < NV()) {
            double d = H[j][j];
#pragma unroll
            for (int k = 0; k < MAXNV; ++k) if (k < j) d -= H[j][k] * H[j][k];
            if (!(d > 0.0)) { ok = false; d = 1.0; }
            const double ljj = sq(d);
            const double iljj = 1.0 / ljj;
            H[j][j] = ljj;
#pragma unroll
            for (int i = 0; i < MAXNV; ++i) if (i > j && i < NV()) {
                double v = H[i][j];
#pragma unroll
                for (int k = 0; k < MAXNV; ++k) if (k < j) v -= H[i][k] * H[j][k];
                H[i][j] = v * iljj;
            }
        }
        return ok;
    }
    BCBF_HD inline void chol_backsolve(const double (*H)[MAXNV], double* b) const {
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
    BCBF_HD inline bool chol_solve(double (*H)[MAXNV], double* b) const {
        const bool ok = chol_factor(H);
        chol_backsolve(H, b);
        return ok;
    }
};


}  // namespace bcbf
