// Real nonsymmetric eigen-decomposition of an n x n matrix, n <= 4, by the algorithm LAPACK's xGEEV runs for such sizes
// -- and the reference's clean-up of a kernel Hessian that is built on it.
//
// Why this exists.  bayes_cbf/gp_algebra.py:384-392 (GradientGP.knl at x' == x) eigen-decomposes the Hessian with the
// GENERAL solver (`torch.eig` = xGEEV), asserts every eigenvalue > -2e-3, zeroes those in (-2e-3, 0) and rebuilds
//        H <- eigenvectors.T @ diag(evalz) @ eigenvectors            (V' L V, not V L V')
// With V the matrix whose COLUMNS are the eigenvectors this is not the spectral projection: entry (i,j) is
// sum_k V[k][i] l_k V[k][j] -- it pairs eigenvalue k with ROW k of V -- so the result depends on the ORDER in which
// the solver returns the eigenvalues and on the SIGN it happens to give each eigenvector (flipping column j flips the
// sign of row / column j of the result).  To give the reference's numbers the device therefore has to walk xGEEV's own
// path, not any eigen-solver's:
//   xGEBAL('B')  permutation that isolates eigenvalues (zero rows / columns: a barrier gradient with a zero component),
//                then power-of-two row / column scaling;
//   xGEHD2 + xORG2R  Householder reduction to Hessenberg form, Q formed explicitly;
//   xLAHQR  double-shift QR on the active block (one real shift used twice when the trailing 2x2 has real eigenvalues,
//           Ahues-Tisseur deflation test, exceptional shifts after 10 / 20 sweeps), xLANV2 standardisation of a 2x2;
//   xTREVC  back-substitution for the eigenvectors of the triangular factor (x_k = 1), multiplied by the Schur vectors,
//           scaled by the largest component;  xGEBAK;  columns normalised to unit 2-norm.
// For n <= 2 (the reference's only rel-degree-2 system, the pendulum) there are no QR sweeps and the outcome is a closed
// form of the entries: every LAPACK build agrees, and so does this file.  For an active block of 3 or 4 the NUMBER of
// sweeps before a deflation depends on rounding-level residues of the sub-diagonal; one sweep more or less flips the
// sign of two Schur vectors.  Measured on 6000 random symmetric matrices with one eigenvalue in (-1e-4, 0): MKL (torch)
// and OpenBLAS 0.3.29 (numpy) disagree with each other in 1.5 % (n=3) / 2.2 % (n=4) of them, this file with either in
// the same share -- the reference's own output is not reproducible across LAPACK builds there.
//
// Host + device code (plain C++, double precision, no library call): compiled into the kernels and, by the CPU tests,
// into a host shared object that is compared with numpy's / torch's xGEEV.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define BCBF_HD __host__ __device__
#else
#define BCBF_HD
#endif

namespace bcbf {
namespace geev {

constexpr int NMAX = 4;
constexpr double ULP = 2.220446049250313e-16;      // dlamch('P') = eps * base
constexpr double SAFMIN = 2.2250738585072014e-308;

BCBF_HD inline double sgn(double a, double b) { return (b >= 0.0 && !(b == 0.0 && signbit(b))) ? fabs(a) : -fabs(a); }   // Fortran SIGN

// Householder reflector H = I - tau [1; v] [1; v]' with H [alpha; x] = [beta; 0]  (xLARFG; n counts alpha)
BCBF_HD inline void larfg(int n, double& alpha, double* x, double& tau) {
    tau = 0.0;
    if (n <= 1) return;
    double xn = 0.0;
    for (int i = 0; i < n - 1; ++i) xn += x[i] * x[i];
    xn = sqrt(xn);
    if (xn == 0.0) return;
    const double beta = -sgn(hypot(alpha, xn), alpha);
    tau = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    for (int i = 0; i < n - 1; ++i) x[i] *= sc;
    alpha = beta;
}

// Schur factorisation of a real 2x2 in standardised form (xLANV2); only the rotation and the new entries are needed
BCBF_HD inline void lanv2(double& a, double& b, double& c, double& d, double& rt1i, double& cs, double& sn) {
    const double multpl = 4.0;
    if (c == 0.0) {
        cs = 1.0; sn = 0.0;
    } else if (b == 0.0) {
        cs = 0.0; sn = 1.0;
        const double t = d; d = a; a = t;
        b = -c; c = 0.0;
    } else if ((a - d) == 0.0 && sgn(1.0, b) != sgn(1.0, c)) {
        cs = 1.0; sn = 0.0;
    } else {
        double temp = a - d;
        double p = 0.5 * temp;
        const double bcmax = fmax(fabs(b), fabs(c));
        const double bcmis = fmin(fabs(b), fabs(c)) * sgn(1.0, b) * sgn(1.0, c);
        const double scale = fmax(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= multpl * ULP) {                       // real eigenvalues
            z = p + sgn(sqrt(scale) * sqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            const double tau = hypot(c, z);
            cs = z / tau; sn = c / tau;
            b = b - c; c = 0.0;
        } else {                                       // complex or nearly equal real eigenvalues: make the diagonal equal
            const double sigma = b + c;
            double tau = hypot(sigma, temp);
            cs = sqrt(0.5 * (1.0 + fabs(sigma) / tau));
            sn = -(p / (tau * cs)) * sgn(1.0, sigma);
            const double aa = a * cs + b * sn, bb = -a * sn + b * cs, cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn; b = bb * cs + dd * sn; c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
            temp = 0.5 * (a + d);
            a = temp; d = temp;
            if (c != 0.0) {
                if (b != 0.0) {
                    if (sgn(1.0, b) == sgn(1.0, c)) {  // real eigenvalues after all: reduce to upper triangular form
                        const double sab = sqrt(fabs(b)), sac = sqrt(fabs(c));
                        p = sgn(sab * sac, c);
                        tau = 1.0 / sqrt(fabs(b + c));
                        a = temp + p; d = temp - p;
                        b = b - c; c = 0.0;
                        const double cs1 = sab * tau, sn1 = sac * tau;
                        temp = cs * cs1 - sn * sn1;
                        sn = cs * sn1 + sn * cs1;
                        cs = temp;
                    }
                } else {
                    b = -c; c = 0.0;
                    temp = cs; cs = -sn; sn = temp;
                }
            }
        }
    }
    rt1i = (c == 0.0) ? 0.0 : sqrt(fabs(b)) * sqrt(fabs(c));
}

// Eigenvalues wr[] (all real, else the return value is 2) and unit-norm right eigenvectors as the COLUMNS of V, in the
// order and with the signs xGEEV(jobvr='V') produces.  A is destroyed.  Return: 0 ok, 1 no convergence, 2 a complex pair.
BCBF_HD inline int geev_real(int n, double A[NMAX][NMAX], double wr[NMAX], double V[NMAX][NMAX]) {
    double scale[NMAX];
    int perm[NMAX];
    for (int i = 0; i < NMAX; ++i) { scale[i] = 1.0; perm[i] = i; wr[i] = 0.0; }
    // ---- xGEBAL('B'): isolate eigenvalues by permutation (rows from the bottom, then columns from the top)
    int k = 0, l = n - 1;
    bool alone = false;
    for (bool again = true; again && !alone;) {
        again = false;
        for (int i = l; i >= 0 && !alone; --i) {
            bool can = true;
            for (int j = 0; j <= l; ++j)
                if (i != j && A[i][j] != 0.0) { can = false; break; }
            if (!can) continue;
            perm[l] = i;
            if (i != l) {
                for (int r = 0; r <= l; ++r) { const double t = A[r][i]; A[r][i] = A[r][l]; A[r][l] = t; }
                for (int c = k; c < n; ++c) { const double t = A[i][c]; A[i][c] = A[l][c]; A[l][c] = t; }
            }
            again = true;
            if (l == 0) { alone = true; break; }
            --l;
        }
    }
    if (!alone) {
        for (bool again = true; again;) {
            again = false;
            for (int j = k; j <= l; ++j) {
                bool can = true;
                for (int i = k; i <= l; ++i)
                    if (i != j && A[i][j] != 0.0) { can = false; break; }
                if (!can) continue;
                perm[k] = j;
                if (j != k) {
                    for (int r = 0; r <= l; ++r) { const double t = A[r][j]; A[r][j] = A[r][k]; A[r][k] = t; }
                    for (int c = k; c < n; ++c) { const double t = A[j][c]; A[j][c] = A[k][c]; A[k][c] = t; }
                }
                again = true;
                ++k;
            }
        }
        // ... and balance the norms of the rows / columns of the active block by powers of two
        const double sclfac = 2.0, factor = 0.95;
        const double sfmin1 = SAFMIN / ULP, sfmax1 = 1.0 / sfmin1, sfmin2 = sfmin1 * sclfac, sfmax2 = 1.0 / sfmin2;
        int passes = 0;
        for (bool again = true; again && passes < 64; ++passes) {     // (bounded: no input may spin a device thread forever)
            again = false;
            for (int i = k; i <= l; ++i) {
                double c = 0.0, r = 0.0, ca = 0.0, ra = 0.0;
                for (int q = k; q <= l; ++q) { c += A[q][i] * A[q][i]; r += A[i][q] * A[i][q]; }
                c = sqrt(c); r = sqrt(r);
                for (int q = 0; q <= l; ++q) ca = fmax(ca, fabs(A[q][i]));
                for (int q = k; q < n; ++q) ra = fmax(ra, fabs(A[i][q]));
                if (c == 0.0 || r == 0.0) continue;
                double g = r / sclfac, f = 1.0;
                const double s = c + r;
                while (c < g && fmax(f, fmax(c, ca)) < sfmax2 && fmin(r, fmin(g, ra)) > sfmin2) {
                    f *= sclfac; c *= sclfac; ca *= sclfac; r /= sclfac; g /= sclfac; ra /= sclfac;
                }
                g = c / sclfac;
                while (g >= r && fmax(r, ra) < sfmax2 && fmin(fmin(f, c), fmin(g, ca)) > sfmin2) {
                    f /= sclfac; c /= sclfac; g /= sclfac; ca /= sclfac; r *= sclfac; ra *= sclfac;
                }
                if ((c + r) >= factor * s) continue;
                if (f < 1.0 && scale[i] < 1.0 && f * scale[i] <= sfmin1) continue;
                if (f > 1.0 && scale[i] > 1.0 && scale[i] >= sfmax1 / f) continue;
                scale[i] *= f;
                again = true;
                for (int q = k; q < n; ++q) A[i][q] /= f;
                for (int q = 0; q <= l; ++q) A[q][i] *= f;
            }
        }
    } else {
        k = 0; l = 0;
    }
    const int ilo = k, ihi = l;
    // ---- xGEHD2: Hessenberg form of the active block; xORG2R: Z = H(ilo) ... H(ihi-1)
    double Z[NMAX][NMAX];
    for (int i = 0; i < NMAX; ++i) for (int j = 0; j < NMAX; ++j) Z[i][j] = i == j ? 1.0 : 0.0;
    double taus[NMAX], vs[NMAX][NMAX];
    for (int i = ilo; i < ihi; ++i) {
        double alpha = A[i + 1][i], x[NMAX];
        const int nr = ihi - i;                            // length incl. alpha
        for (int q = 0; q < nr - 1; ++q) x[q] = A[i + 2 + q][i];
        double tau;
        larfg(nr, alpha, x, tau);
        double v[NMAX];
        v[0] = 1.0;
        for (int q = 0; q < nr - 1; ++q) v[1 + q] = x[q];
        A[i + 1][i] = alpha;
        for (int q = i + 2; q <= ihi; ++q) A[q][i] = 0.0;
        for (int r = 0; r <= ihi; ++r) {                   // from the right: A[0:ihi+1, i+1:ihi+1]
            double w = 0.0;
            for (int q = 0; q < nr; ++q) w += A[r][i + 1 + q] * v[q];
            for (int q = 0; q < nr; ++q) A[r][i + 1 + q] -= tau * w * v[q];
        }
        for (int c = i + 1; c < n; ++c) {                  // from the left: A[i+1:ihi+1, i+1:n]
            double w = 0.0;
            for (int q = 0; q < nr; ++q) w += v[q] * A[i + 1 + q][c];
            for (int q = 0; q < nr; ++q) A[i + 1 + q][c] -= tau * v[q] * w;
        }
        taus[i] = tau;
        for (int q = 0; q < nr; ++q) vs[i][q] = v[q];
    }
    for (int i = ihi - 1; i >= ilo; --i) {
        const int nr = ihi - i;
        for (int c = 0; c < n; ++c) {
            double w = 0.0;
            for (int q = 0; q < nr; ++q) w += vs[i][q] * Z[i + 1 + q][c];
            for (int q = 0; q < nr; ++q) Z[i + 1 + q][c] -= taus[i] * vs[i][q] * w;
        }
    }
    // ---- xLAHQR on rows / columns ilo..ihi (the whole Schur form T and the Schur vectors Z are updated)
    for (int i = 0; i < n; ++i) if (i < ilo || i > ihi) wr[i] = A[i][i];
    if (ilo == ihi) {
        wr[ilo] = A[ilo][ilo];
    } else {
        for (int j = ilo; j <= ihi - 3; ++j) { A[j + 2][j] = 0.0; A[j + 3][j] = 0.0; }
        if (ilo <= ihi - 2) A[ihi][ihi - 2] = 0.0;
        const int nh = ihi - ilo + 1;
        const double smlnum = SAFMIN * ((double)nh / ULP);
        const int i1 = 0, i2 = n - 1;
        const int itmax = 30 * (nh > 10 ? nh : 10);
        int kdefl = 0;
        int i = ihi;
        while (i >= ilo) {
            int L = ilo;
            bool split = false;
            for (int its = 0; its <= itmax; ++its) {
                int kk = i;
                for (; kk > L; --kk) {                     // a negligible sub-diagonal entry?
                    if (fabs(A[kk][kk - 1]) <= smlnum) break;
                    double tst = fabs(A[kk - 1][kk - 1]) + fabs(A[kk][kk]);
                    if (tst == 0.0) {
                        if (kk - 2 >= ilo) tst += fabs(A[kk - 1][kk - 2]);
                        if (kk + 1 <= ihi) tst += fabs(A[kk + 1][kk]);
                    }
                    if (fabs(A[kk][kk - 1]) <= ULP * tst) {
                        const double ab = fmax(fabs(A[kk][kk - 1]), fabs(A[kk - 1][kk]));
                        const double ba = fmin(fabs(A[kk][kk - 1]), fabs(A[kk - 1][kk]));
                        const double aa = fmax(fabs(A[kk][kk]), fabs(A[kk - 1][kk - 1] - A[kk][kk]));
                        const double bb = fmin(fabs(A[kk][kk]), fabs(A[kk - 1][kk - 1] - A[kk][kk]));
                        const double s = aa + ab;
                        if (ba * (ab / s) <= fmax(smlnum, ULP * (bb * (aa / s)))) break;
                    }
                }
                L = kk;
                if (L > ilo) A[L][L - 1] = 0.0;
                if (L >= i - 1) { split = true; break; }
                ++kdefl;
                double h11, h21, h12, h22;
                if (kdefl % 20 == 0) {                     // exceptional shifts
                    const double s = fabs(A[i][i - 1]) + fabs(A[i - 1][i - 2]);
                    h11 = 0.75 * s + A[i][i]; h12 = -0.4375 * s; h21 = s; h22 = h11;
                } else if (kdefl % 10 == 0) {
                    const double s = fabs(A[L + 1][L]) + fabs(A[L + 2][L + 1]);
                    h11 = 0.75 * s + A[L][L]; h12 = -0.4375 * s; h21 = s; h22 = h11;
                } else {
                    h11 = A[i - 1][i - 1]; h21 = A[i][i - 1]; h12 = A[i - 1][i]; h22 = A[i][i];
                }
                double rt1r, rt1i, rt2r, rt2i;
                {
                    const double s = fabs(h11) + fabs(h12) + fabs(h21) + fabs(h22);
                    if (s == 0.0) {
                        rt1r = rt1i = rt2r = rt2i = 0.0;
                    } else {
                        h11 /= s; h21 /= s; h12 /= s; h22 /= s;
                        const double tr = (h11 + h22) / 2.0;
                        const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                        const double rtdisc = sqrt(fabs(det));
                        if (det >= 0.0) {                  // complex conjugate shifts
                            rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i;
                        } else {                           // real shifts: the one nearer h22, twice
                            rt1r = tr + rtdisc; rt2r = tr - rtdisc;
                            if (fabs(rt1r - h22) <= fabs(rt2r - h22)) { rt1r = rt1r * s; rt2r = rt1r; }
                            else { rt2r = rt2r * s; rt1r = rt2r; }
                            rt1i = rt2i = 0.0;
                        }
                    }
                }
                int m = i - 2;
                double v[3] = {0.0, 0.0, 0.0};
                for (;; --m) {                             // two consecutive small sub-diagonal entries?
                    double h21s = fabs(A[m + 1][m]);
                    double s = fabs(A[m][m] - rt2r) + fabs(rt2i) + h21s;
                    h21s = A[m + 1][m] / s;
                    v[0] = h21s * A[m][m + 1] + (A[m][m] - rt1r) * ((A[m][m] - rt2r) / s) - rt1i * (rt2i / s);
                    v[1] = h21s * (A[m][m] + A[m + 1][m + 1] - rt1r - rt2r);
                    v[2] = h21s * A[m + 2][m + 1];
                    s = fabs(v[0]) + fabs(v[1]) + fabs(v[2]);
                    v[0] /= s; v[1] /= s; v[2] /= s;
                    if (m == L) break;
                    if (fabs(A[m][m - 1]) * (fabs(v[1]) + fabs(v[2])) <=
                        ULP * fabs(v[0]) * (fabs(A[m - 1][m - 1]) + fabs(A[m][m]) + fabs(A[m + 1][m + 1])))
                        break;
                }
                for (int q = m; q <= i - 1; ++q) {         // the double-shift sweep: chase the bulge
                    const int nr = (i - q + 1) < 3 ? (i - q + 1) : 3;
                    if (q > m) { for (int r = 0; r < 3; ++r) v[r] = r < nr ? A[q + r][q - 1] : 0.0; }
                    double t1;
                    larfg(nr, v[0], v + 1, t1);
                    if (q > m) {
                        A[q][q - 1] = v[0];
                        A[q + 1][q - 1] = 0.0;
                        if (q < i - 1) A[q + 2][q - 1] = 0.0;
                    } else if (m > L) {
                        A[q][q - 1] = A[q][q - 1] * (1.0 - t1);
                    }
                    const double v2 = v[1], t2 = t1 * v2;
                    if (nr == 3) {
                        const double v3 = v[2], t3 = t1 * v3;
                        for (int j = q; j <= i2; ++j) {
                            const double sm = A[q][j] + v2 * A[q + 1][j] + v3 * A[q + 2][j];
                            A[q][j] -= sm * t1; A[q + 1][j] -= sm * t2; A[q + 2][j] -= sm * t3;
                        }
                        const int jmax = (q + 3) < i ? (q + 3) : i;
                        for (int j = i1; j <= jmax; ++j) {
                            const double sm = A[j][q] + v2 * A[j][q + 1] + v3 * A[j][q + 2];
                            A[j][q] -= sm * t1; A[j][q + 1] -= sm * t2; A[j][q + 2] -= sm * t3;
                        }
                        for (int j = 0; j < n; ++j) {
                            const double sm = Z[j][q] + v2 * Z[j][q + 1] + v3 * Z[j][q + 2];
                            Z[j][q] -= sm * t1; Z[j][q + 1] -= sm * t2; Z[j][q + 2] -= sm * t3;
                        }
                    } else {
                        for (int j = q; j <= i2; ++j) {
                            const double sm = A[q][j] + v2 * A[q + 1][j];
                            A[q][j] -= sm * t1; A[q + 1][j] -= sm * t2;
                        }
                        for (int j = i1; j <= i; ++j) {
                            const double sm = A[j][q] + v2 * A[j][q + 1];
                            A[j][q] -= sm * t1; A[j][q + 1] -= sm * t2;
                        }
                        for (int j = 0; j < n; ++j) {
                            const double sm = Z[j][q] + v2 * Z[j][q + 1];
                            Z[j][q] -= sm * t1; Z[j][q + 1] -= sm * t2;
                        }
                    }
                }
            }
            if (!split) return 1;
            if (L == i) {
                wr[i] = A[i][i];
            } else {                                       // a 2x2 block split off: standardise it
                double rt1i, cs, sn;
                lanv2(A[i - 1][i - 1], A[i - 1][i], A[i][i - 1], A[i][i], rt1i, cs, sn);
                if (rt1i != 0.0) return 2;
                wr[i - 1] = A[i - 1][i - 1]; wr[i] = A[i][i];
                for (int j = i + 1; j <= i2; ++j) {
                    const double x_ = A[i - 1][j], y_ = A[i][j];
                    A[i - 1][j] = cs * x_ + sn * y_; A[i][j] = cs * y_ - sn * x_;
                }
                for (int j = i1; j < i - 1; ++j) {
                    const double x_ = A[j][i - 1], y_ = A[j][i];
                    A[j][i - 1] = cs * x_ + sn * y_; A[j][i] = cs * y_ - sn * x_;
                }
                for (int j = 0; j < n; ++j) {
                    const double x_ = Z[j][i - 1], y_ = Z[j][i];
                    Z[j][i - 1] = cs * x_ + sn * y_; Z[j][i] = cs * y_ - sn * x_;
                }
            }
            kdefl = 0;
            i = L - 1;
        }
    }
    // ---- xTREVC (all eigenvalues real): eigenvectors of the triangular T by back-substitution, times the Schur vectors
    const double smlnum_t = SAFMIN * ((double)n / ULP);
    for (int ki = n - 1; ki >= 0; --ki) {
        const double w = A[ki][ki];
        const double smin = fmax(ULP * fabs(w), smlnum_t);
        double x[NMAX];
        for (int q = 0; q < ki; ++q) x[q] = -A[q][ki];
        x[ki] = 1.0;
        for (int j = ki - 1; j >= 0; --j) {
            double den = A[j][j] - w;
            if (fabs(den) < smin) den = smin;
            const double xx = x[j] / den;
            x[j] = xx;
            for (int q = 0; q < j; ++q) x[q] -= xx * A[q][j];
        }
        double emax = 0.0;
        double vcol[NMAX];
        for (int r = 0; r < n; ++r) {
            double t = 0.0;
            for (int q = 0; q <= ki; ++q) t += Z[r][q] * x[q];
            vcol[r] = t;
            emax = fmax(emax, fabs(t));
        }
        for (int r = 0; r < n; ++r) V[r][ki] = vcol[r] / emax;
    }
    // ---- xGEBAK: undo the scaling, then the permutation; unit 2-norm columns
    if (ilo != ihi)
        for (int r = ilo; r <= ihi; ++r) for (int c = 0; c < n; ++c) V[r][c] *= scale[r];
    for (int ii = ilo - 1; ii >= 0; --ii) {
        const int kx = perm[ii];
        if (kx != ii) for (int c = 0; c < n; ++c) { const double t = V[ii][c]; V[ii][c] = V[kx][c]; V[kx][c] = t; }
    }
    for (int ii = ihi + 1; ii < n; ++ii) {
        const int kx = perm[ii];
        if (kx != ii) for (int c = 0; c < n; ++c) { const double t = V[ii][c]; V[ii][c] = V[kx][c]; V[kx][c] = t; }
    }
    for (int c = 0; c < n; ++c) {
        double s = 0.0;
        for (int r = 0; r < n; ++r) s += V[r][c] * V[r][c];
        s = 1.0 / sqrt(s);
        for (int r = 0; r < n; ++r) V[r][c] *= s;
    }
    return 0;
}

// gp_algebra.py:384-392 on H (n x n, in place).  Return value (the kernels' status word):
//   0 nothing to do (every eigenvalue >= 0), 1 an eigenvalue <= -eps (the reference's assert fails),
//   4 the branch fired: H <- V' diag(max-zeroed eigenvalues) V with xGEEV's V,
//   6 the branch fired but the general solver saw a complex pair / did not converge (a nearly repeated eigenvalue of a
//     slightly non-symmetric H): H <- spectral projection of the symmetric part (the fallback below).
// mode 0 = reference formula, 1 = spectral projection V max(L,0) V' of the symmetric part (the round-1..3 behaviour).
BCBF_HD inline void project_psd(int n, double H[NMAX][NMAX], double w[NMAX], double Vv[NMAX][NMAX]);

BCBF_HD inline int clean_hessian(int n, double H[NMAX][NMAX], double eps, int mode) {
    double w[NMAX], Vs[NMAX][NMAX], S[NMAX][NMAX];
    // a non-finite entry: the reference's eigenvalues are NaN there and its `assert (evalz > -EPS).all()` fails (status 1, H
    // untouched).  Decided HERE: xGEBAL's balancing loop does not terminate on NaN input (every comparison is false; newer
    // LAPACK releases test DISNAN at that spot for the same reason)
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (!(fabs(H[i][j]) <= 1.7976931348623157e308)) return 1;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[i][j] = 0.5 * (H[i][j] + H[j][i]);
    {   // The common case first: a symmetric part that is positive definite by a wide margin has no eigenvalue < 0 and the
        // matrix stays as it is (status 0) -- decided by an unpivoted Cholesky whose pivots all exceed 1e-10 of the trace
        // (a negative eigenvalue forces a pivot <= 0 up to rounding, 1e-16 of the trace: the margin cannot be bridged), without
        // the twelve Jacobi sweeps below.  Anything closer to singular than that takes the full path.
        double L[NMAX][NMAX], tr = 0.0;
        for (int i = 0; i < n; ++i) tr += fabs(S[i][i]);
        bool pd = tr > 0.0;
        for (int j = 0; j < n && pd; ++j) {
            double d = S[j][j];
            for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
            if (!(d > 1e-10 * tr)) { pd = false; break; }
            const double r = 1.0 / sqrt(d);
            L[j][j] = d * r;
            for (int i = j + 1; i < n; ++i) {
                double t = S[i][j];
                for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
                L[i][j] = t * r;
            }
        }
        if (pd) return 0;
    }
    // Which branch runs is decided as the reference decides it (gp_algebra.py:385-387): from the real parts of xGEEV's
    // eigenvalues of H ITSELF -- `assert (evalz > -EPS).all()`, then `if (evalz < 0).any()` -- not from the symmetric part
    // (for an H that is non-symmetric by rounding the two can differ next to 0 and next to -EPS).
    if (mode == 0) {
        double Acp[NMAX][NMAX], wr[NMAX], V[NMAX][NMAX];
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Acp[i][j] = H[i][j];
        if (geev_real(n, Acp, wr, V) == 0) {
            bool any = false, bad = false;
            for (int k = 0; k < n; ++k) { if (wr[k] <= -eps) bad = true; if (wr[k] < 0.0) { any = true; wr[k] = 0.0; } }
            if (bad) return 1;
            if (!any) return 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double t = 0.0;
                    for (int k = 0; k < n; ++k) t += V[k][i] * wr[k] * V[k][j];
                    H[i][j] = t;
                }
            return 4;
        }
    }
    // mode 1 (spectral projection), or the general solver saw a complex pair / did not converge: the symmetric part decides
    project_psd(n, S, w, Vs);
    bool neg = false, bad = false;
    for (int i = 0; i < n; ++i) { if (w[i] <= -eps) bad = true; if (w[i] < 0.0) neg = true; }
    if (bad) return 1;
    if (!neg) return 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += Vs[i][k] * fmax(w[k], 0.0) * Vs[j][k];
            H[i][j] = t;
        }
    return mode == 0 ? 6 : 4;
}

// cyclic Jacobi on a symmetric S (destroyed): eigenvalues w, eigenvectors as columns of Vv
BCBF_HD inline void project_psd(int n, double S[NMAX][NMAX], double w[NMAX], double Vv[NMAX][NMAX]) {
    for (int i = 0; i < NMAX; ++i) for (int j = 0; j < NMAX; ++j) Vv[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                if (fabs(S[p][q]) < 1e-300) continue;
                const double th = 0.5 * (S[q][q] - S[p][p]) / S[p][q];
                const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < n; ++k) { const double a_ = S[k][p], b_ = S[k][q]; S[k][p] = cs * a_ - sn * b_; S[k][q] = sn * a_ + cs * b_; }
                for (int k = 0; k < n; ++k) { const double a_ = S[p][k], b_ = S[q][k]; S[p][k] = cs * a_ - sn * b_; S[q][k] = sn * a_ + cs * b_; }
                for (int k = 0; k < n; ++k) { const double a_ = Vv[k][p], b_ = Vv[k][q]; Vv[k][p] = cs * a_ - sn * b_; Vv[k][q] = sn * a_ + cs * b_; }
            }
    for (int i = 0; i < n; ++i) w[i] = S[i][i];
}

}  // namespace geev
}  // namespace bcbf
