// K8 (rel-degree 1) + K9: chance-constraint terms of a control barrier / Lyapunov condition and
// their second-order-cone form.  One lane per (instance, constraint); closed form of what the
// reference obtains with autograd through its gp_algebra expression.
#include "bcbf_common.h"
#include "geev_small.h"

namespace bcbf {

template <typename T>
__global__ void cbc_terms_kernel(const T* __restrict__ Mk, const T* __restrict__ Bk, const T* __restrict__ A,
                                 const T* __restrict__ grad, const T* __restrict__ cst, const T* __restrict__ sign,
                                 const T* __restrict__ fhat, const T* __restrict__ ghat, T* __restrict__ terms,
                                 T* __restrict__ cones, int* __restrict__ cstatus, int Bt, int K, int n, int m) {
    constexpr int MM = BCBF_MAX_CTRL_DIM, NN = BCBF_MAX_STATE_DIM;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bt * K) return;
    const int b = idx / K, k = idx - b * K;
    const int C = m + 1;
    const T* Mkb = Mk + (size_t)b * n * C;
    const T* Bkb = Bk + (size_t)b * C * C;
    const T* Ab = A + (size_t)b * n * n;
    const T* g = grad + ((size_t)b * K + k) * n;
    const T* fh = fhat + (size_t)b * n;
    const T* gh = ghat + (size_t)b * n * m;
    const double sg = (double)sign[k];
    double gd[NN];
    for (int d = 0; d < NN; ++d) gd[d] = d < n ? (double)g[d] : 0.0;
    double a_h = 0.0;
    for (int d = 0; d < n; ++d) {
        double t = 0.0;
        for (int e2 = 0; e2 < n; ++e2) t += (double)Ab[d * n + e2] * gd[e2];
        a_h += gd[d] * t;
    }
    double bfe[MM], bfv[MM], V[MM][MM];
    double e = (double)cst[(size_t)b * K + k];
    for (int d = 0; d < n; ++d) e += gd[d] * ((double)fh[d] + (double)Mkb[d * C]);
    e *= sg;
    for (int i = 0; i < MM; ++i) {
        double s = 0.0;
        if (i < m) for (int d = 0; d < n; ++d) s += ((double)gh[d * m + i] + (double)Mkb[d * C + 1 + i]) * gd[d];
        bfe[i] = sg * s;
        bfv[i] = i < m ? 2.0 * a_h * (double)Bkb[(1 + i) * C] : 0.0;
        for (int j = 0; j < MM; ++j) V[i][j] = (i < m && j < m) ? a_h * (double)Bkb[(1 + i) * C + 1 + j] : 0.0;
    }
    const double v = a_h * (double)Bkb[0];
    if (terms) {
        T* t = terms + ((size_t)b * K + k) * (m + 1 + m * m + m + 1);
        for (int i = 0; i < m; ++i) t[i] = (T)bfe[i];
        t[m] = (T)e;
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) t[m + 1 + i * m + j] = (T)V[i][j];
        for (int i = 0; i < m; ++i) t[m + 1 + m * m + i] = (T)bfv[i];
        t[m + 1 + m * m + m] = (T)v;
    }
    // cone form: Asq = [[v, bfv'/2],[bfv/2, V]] = L L'
    double L[MM + 1][MM + 1], Asq[MM + 1][MM + 1];
    for (int a = 0; a <= MM; ++a) for (int c = 0; c <= MM; ++c) { L[a][c] = 0.0; Asq[a][c] = 0.0; }
    Asq[0][0] = v;
    for (int i = 0; i < m; ++i) {
        Asq[0][1 + i] = Asq[1 + i][0] = 0.5 * bfv[i];
        for (int j = 0; j < m; ++j) Asq[1 + i][1 + j] = V[i][j];
    }
    int bad = 0;
    for (int j = 0; j < C; ++j) {
        double d = Asq[j][j];
        for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
        if (!(d > 0.0)) { bad = BCBF_SOCP_BADCONE; d = 1.0; }
        const double ljj = sqrt(d);
        L[j][j] = ljj;
        for (int i = j + 1; i < C; ++i) {
            double s = Asq[i][j];
            for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
            L[i][j] = s / ljj;
        }
    }
    if (cstatus) cstatus[(size_t)b * K + k] = bad;
    if (cones) {
        T* cn = cones + ((size_t)b * K + k) * ((m + 1) * m + (m + 1) + m + 1);
        for (int a = 0; a < C; ++a) for (int i = 0; i < m; ++i) cn[a * m + i] = (T)L[1 + i][a];   // A = L'[:,1:]
        T* cb = cn + C * m;
        for (int a = 0; a < C; ++a) cb[a] = (T)L[0][a];                                           // b = L'[:,0]
        T* cc = cb + C;
        for (int i = 0; i < m; ++i) cc[i] = (T)bfe[i];
        cc[m] = (T)e;
    }
}

template <typename T>
static int launch_cbc_terms(const T* Mk, const T* Bk, const T* A, const T* grad, const T* cst, const T* sign,
                            const T* fhat, const T* ghat, T* terms, T* cones, int* cstatus,
                            int Bt, int K, int n, int m, void* stream) {
    if (Bt <= 0 || K <= 0) return BCBF_OK;
    if (!Mk || !Bk || !A || !grad || !cst || !sign || !fhat || !ghat) return BCBF_EINVAL;
    if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM || K > BCBF_MAX_CONSTRAINTS) return BCBF_EINVAL;
    const int total = Bt * K;
    hipLaunchKernelGGL((cbc_terms_kernel<T>), dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, Mk, Bk, A,
                       grad, cst, sign, fhat, ghat, terms, cones, cstatus, Bt, K, n, m);
    return check_launch("cbc_terms");
}


// --------------------------------------------------------------------------------------------
// K8, rel-degree 2: CBC2 = grad(L_f h)'(f + g u) + ka0 h + ka1 L_f h as a GP in u, from the posterior jets
// (closed form of cbc2_gp + cbc2_quadratic_terms, cbc2.py:7-33, gp_algebra.py:133-168, 319-402;
// SURVEY.md A.4).  One lane per instance, fp64 internally.
//   out[Bt, m + 1 + m*m + m + 1 + 2] = (mean_A[m], mean_b, Q[m,m], p[m], r, mean(u0), var(u0))
//   status: 0 ok, 1 = kernel Hessian has an eigenvalue <= -2e-3 (the reference asserts, gp_algebra.py:386),
//           4 = the clean-up branch of gp_algebra.py:387-392 ran (eigenvalues in (-2e-3, 0) zeroed), 6 = it ran through
//           the projection because the general eigen-solver saw a complex pair (geev_small.h).
template <typename T>
__global__ void cbc2_terms_kernel(const T* __restrict__ Mk, const T* __restrict__ Bk, const T* __restrict__ G,
                                  const T* __restrict__ Mj, const T* __restrict__ A, const T* __restrict__ Bm,
                                  const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ hval,
                                  const T* __restrict__ gh_, const T* __restrict__ Hh_, const T* __restrict__ kalpha,
                                  const T* __restrict__ u0_, T* __restrict__ out, int* __restrict__ status,
                                  int Bt, int n, int m, int hessian_mode, int kernel_kind) {
    constexpr int NN = 4, MM = BCBF_MAX_CTRL_DIM, CC = MM + 1;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    const int C = m + 1, CT = C * (1 + n);
    const T* Gb = G + (size_t)b * CT * CT;
    const T* Mjb = Mj + (size_t)b * n * CT;
    auto Gat = [&](int blk_a, int a, int blk_c, int c) { return (double)Gb[(blk_a * C + a) * CT + blk_c * C + c]; };
    double Bkd[CC][CC], Ad[NN][NN], gh[NN], Hh[NN][NN], m0[NN], Mt[NN][MM], u0[MM], a0[CC];
    for (int a = 0; a < CC; ++a) for (int c = 0; c < CC; ++c) Bkd[a][c] = (a < C && c < C) ? (double)Bk[((size_t)b * C + a) * C + c] : 0.0;
    for (int i = 0; i < NN; ++i) {
        gh[i] = i < n ? (double)gh_[(size_t)b * n + i] : 0.0;
        m0[i] = i < n ? (double)Mk[((size_t)b * n + i) * C] : 0.0;
        for (int j = 0; j < NN; ++j) {
            Ad[i][j] = (i < n && j < n) ? (double)A[((size_t)b * n + i) * n + j] : 0.0;
            Hh[i][j] = (i < n && j < n) ? (double)Hh_[((size_t)b * n + i) * n + j] : 0.0;
        }
        for (int j = 0; j < MM; ++j) Mt[i][j] = (i < n && j < m) ? (double)Mk[((size_t)b * n + i) * C + 1 + j] : 0.0;
    }
    a0[0] = 1.0;
    for (int j = 0; j < MM; ++j) { u0[j] = j < m ? (double)u0_[(size_t)b * m + j] : 0.0; a0[1 + j] = u0[j]; }
    const double ka0 = (double)kalpha[0], ka1 = (double)kalpha[1], h = (double)hval[b];
    const double s2 = (double)s2p[b], B00 = (double)Bm[(size_t)b * C * C];
    double Agh[NN], HAg[NN], g[NN];
    double phi0 = 0.0;
    for (int i = 0; i < NN; ++i) { double t = 0; for (int j = 0; j < NN; ++j) t += Ad[i][j] * gh[j]; Agh[i] = t; }
    for (int i = 0; i < NN; ++i) phi0 += gh[i] * Agh[i];
    for (int i = 0; i < NN; ++i) { double t = 0; for (int j = 0; j < NN; ++j) t += Hh[i][j] * Agh[j]; HAg[i] = t; }
    // g = Hh' m0 + [gh' dMk_i e0]_i,   dMk_i e0 = Mj[:, (1+i) C]
    for (int i = 0; i < NN; ++i) {
        double t = 0;
        for (int j = 0; j < NN; ++j) t += Hh[j][i] * m0[j];
        if (i < n) for (int j = 0; j < n; ++j) t += gh[j] * (double)Mjb[j * CT + (1 + i) * C];
        g[i] = i < n ? t : 0.0;
    }
    // total derivative of s(z,a;z,a') : -(a'(G10_i + G10_i')a'),  G10_i = dW_i'W = G[(1+i) blk, 0 blk]
    auto dsdz = [&](const double* a, const double* ap, int i) {
        double t = 0;
        for (int p_ = 0; p_ < C; ++p_) for (int q = 0; q < C; ++q) t += a[p_] * (Gat(1 + i, p_, 0, q) + Gat(1 + i, q, 0, p_)) * ap[q];
        return -t;
    };
    double e0[CC];
    for (int a = 0; a < CC; ++a) e0[a] = a == 0 ? 1.0 : 0.0;
    const double s00 = Bkd[0][0];
    const double kxx = kernel_kxx(kernel_kind);      // d2 k / dx_d dx'_d at x' = x in units of s2 / ell_d^2: RBF 1, Matern-5/2 5/3, product 8/3
    double s_i[NN], H[NN][NN];
    for (int i = 0; i < NN; ++i) s_i[i] = i < n ? -Gat(1 + i, 0, 0, 0) : 0.0;
    for (int i = 0; i < NN; ++i)
        for (int j = 0; j < NN; ++j) {
            double hah = 0;
            for (int p_ = 0; p_ < NN; ++p_) for (int q = 0; q < NN; ++q) hah += Hh[i][p_] * Ad[p_][q] * Hh[q][j];
            double sij = 0;
            if (i < n && j < n) {
                const double li = (double)ell[(size_t)b * n + i];
                sij = (i == j ? kxx * s2 / (li * li) * B00 : 0.0) - Gat(1 + i, 0, 1 + j, 0);
            }
            H[i][j] = hah * s00 + HAg[i] * s_i[j] + s_i[i] * HAg[j] + phi0 * sij;
        }
    // gp_algebra.py:384-392: assert no eigenvalue <= -2e-3, rebuild H with the ones in (-2e-3, 0) zeroed -- by the
    // reference's own formula on xGEEV's eigenvectors (geev_small.h), or the spectral projection (hessian_mode 1)
    static_assert(NN == geev::NMAX, "geev_small.h is written for n <= 4");
    const int st = geev::clean_hessian(n, H, 2e-3, hessian_mode);
    // C(a) = (d/dz [A gh(z) s(z,a;z,e0)])',  J[k][i] = (A Hh)[k][i] s(a,e0) + Agh[k] dsdz_i(a,e0)
    double Cu0[NN][NN], C0[NN][NN];
    for (int which = 0; which < 2; ++which) {
        const double* a = which == 0 ? a0 : e0;
        double sa0 = 0;
        for (int p_ = 0; p_ < C; ++p_) sa0 += a[p_] * Bkd[p_][0];
        for (int i = 0; i < NN; ++i) {
            const double ds = i < n ? dsdz(a, e0, i) : 0.0;
            for (int k = 0; k < NN; ++k) {
                double ahh = 0;
                for (int q = 0; q < NN; ++q) ahh += Ad[k][q] * Hh[q][i];
                const double Jki = ahh * sa0 + Agh[k] * ds;
                if (which == 0) Cu0[i][k] = Jki; else C0[i][k] = Jki;
            }
        }
    }
    double dq[NN];
    for (int i = 0; i < NN; ++i) dq[i] = i < n ? dsdz(e0, e0, i) * phi0 + s00 * 2.0 * HAg[i] : 0.0;
    double gAg = 0, gAgh = 0, trC = 0, ell1 = 0;
    for (int i = 0; i < NN; ++i) { double t = 0; for (int j = 0; j < NN; ++j) t += Ad[i][j] * g[j]; gAg += g[i] * t; gAgh += g[i] * Agh[i]; trC += Cu0[i][i]; ell1 += gh[i] * m0[i]; }
    // helper vectors
    double Hsm0[NN], Ctg[NN], C0gh[NN];       // (H+H')m0, Cu0' g, C0 gh
    for (int i = 0; i < NN; ++i) {
        double t1 = 0, t2 = 0, t3 = 0;
        for (int j = 0; j < NN; ++j) { t1 += (H[i][j] + H[j][i]) * m0[j]; t2 += Cu0[j][i] * g[j]; t3 += C0[i][j] * gh[j]; }
        Hsm0[i] = t1; Ctg[i] = t2; C0gh[i] = t3;
    }
    T* o = out + (size_t)b * (m + 1 + m * m + m + 1 + 2);
    double meanA[MM], Q[MM][MM], pp[MM];
    double gm0 = 0, m0Hm0 = 0, m0Ctg = 0, m0dq = 0, m0C0gh = 0;
    for (int i = 0; i < NN; ++i) { gm0 += g[i] * m0[i]; m0Ctg += m0[i] * Ctg[i]; m0dq += m0[i] * dq[i]; m0C0gh += m0[i] * C0gh[i];
                                   for (int j = 0; j < NN; ++j) m0Hm0 += m0[i] * H[i][j] * m0[j]; }
    for (int a = 0; a < MM; ++a) {
        double ma = 0, p1 = 0, p3 = 0, p5 = 0, p6 = 0;
        for (int i = 0; i < NN; ++i) { ma += Mt[i][a] * g[i]; p1 += Mt[i][a] * Hsm0[i]; p3 += Mt[i][a] * Ctg[i]; p5 += Mt[i][a] * dq[i]; p6 += Mt[i][a] * C0gh[i]; }
        meanA[a] = ma;
        pp[a] = p1 + 2.0 * gAg * Bkd[1 + a][0] + 2.0 * p3 + ka1 * (2.0 * gAgh * Bkd[1 + a][0] + p5 + p6);
        for (int c = 0; c < MM; ++c) {
            double q_ = 0;
            for (int i = 0; i < NN; ++i) for (int j = 0; j < NN; ++j) q_ += Mt[i][a] * (H[i][j] + H[j][i]) * Mt[j][c];
            Q[a][c] = 0.5 * q_ + gAg * Bkd[1 + a][1 + c];
        }
    }
    const double mean_b = gm0 + trC + ka0 * h + ka1 * ell1;
    const double rr = 2.0 * trC * trC + m0Hm0 + gAg * Bkd[0][0] + 2.0 * m0Ctg + ka1 * ka1 * phi0 * s00
                      + ka1 * (2.0 * gAgh * Bkd[0][0] + m0dq + m0C0gh);
    double mean = mean_b, var = rr;
    for (int a = 0; a < m; ++a) { mean += meanA[a] * u0[a]; var += pp[a] * u0[a]; for (int c = 0; c < m; ++c) var += u0[a] * Q[a][c] * u0[c]; }
    int k = 0;
    for (int a = 0; a < m; ++a) o[k++] = (T)meanA[a];
    o[k++] = (T)mean_b;
    for (int a = 0; a < m; ++a) for (int c = 0; c < m; ++c) o[k++] = (T)Q[a][c];
    for (int a = 0; a < m; ++a) o[k++] = (T)pp[a];
    o[k++] = (T)rr;
    o[k++] = (T)mean;
    o[k++] = (T)var;
    if (status) status[b] = st;
}

// The clean-up alone on a batch of n x n matrices (row-major, n <= 4): what GradientGP.knl(x, x) does to its Hessian
// (gp_algebra.py:384-392).  The facade's single-state paths and the tests call it; cbc2_terms_kernel has it inline.
template <typename T>
__global__ void clean_hessian_kernel(const T* __restrict__ Hin, T* __restrict__ Hout, int* __restrict__ status, int Bt,
                                     int n, double eps, int mode) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bt) return;
    double H[geev::NMAX][geev::NMAX];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H[i][j] = (double)Hin[((size_t)b * n + i) * n + j];
    const int st = geev::clean_hessian(n, H, eps, mode);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Hout[((size_t)b * n + i) * n + j] = (T)H[i][j];
    if (status) status[b] = st;
}

template <typename T>
static int launch_clean_hessian(const T* Hin, T* Hout, int* status, int Bt, int n, double eps, int mode, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Hin || !Hout || n < 1 || n > geev::NMAX || (mode != 0 && mode != 1) || !(eps > 0.0)) return BCBF_EINVAL;
    hipLaunchKernelGGL((clean_hessian_kernel<T>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, Hin, Hout, status,
                       Bt, n, eps, mode);
    return check_launch("clean_hessian");
}

template <typename T>
static int launch_cbc2_terms(const T* Mk, const T* Bk, const T* G, const T* Mj, const T* A, const T* Bm, const T* ell,
                             const T* s2, const T* h, const T* gh, const T* Hh, const T* kalpha, const T* u0, T* out,
                             int* status, int Bt, int n, int m, int hessian_mode, int kernel_kind, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if ((hessian_mode != 0 && hessian_mode != 1) || kernel_kind < 0 || kernel_kind >= BCBF_KINDS) return BCBF_EINVAL;
    if (!Mk || !Bk || !G || !Mj || !A || !Bm || !ell || !s2 || !h || !gh || !Hh || !kalpha || !u0 || !out) return BCBF_EINVAL;
    if (n < 1 || n > 4 || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    hipLaunchKernelGGL((cbc2_terms_kernel<T>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, Mk, Bk, G, Mj, A,
                       Bm, ell, s2, h, gh, Hh, kalpha, u0, out, status, Bt, n, m, hessian_mode, kernel_kind);
    return check_launch("cbc2_terms");
}

}  // namespace bcbf

extern "C" {
int bcbf_cbc_terms_f32(const float* Mk, const float* Bk, const float* A, const float* grad, const float* cst,
                       const float* sign, const float* fhat, const float* ghat,
                       float* terms, float* cones, int* cstatus, int Bt, int K, int n, int m, void* stream) {
    return bcbf::launch_cbc_terms<float>(Mk, Bk, A, grad, cst, sign, fhat, ghat, terms, cones, cstatus, Bt, K, n, m, stream);
}
int bcbf_cbc_terms_f64(const double* Mk, const double* Bk, const double* A, const double* grad, const double* cst,
                       const double* sign, const double* fhat, const double* ghat,
                       double* terms, double* cones, int* cstatus, int Bt, int K, int n, int m, void* stream) {
    return bcbf::launch_cbc_terms<double>(Mk, Bk, A, grad, cst, sign, fhat, ghat, terms, cones, cstatus, Bt, K, n, m, stream);
}
int bcbf_clean_hessian_f32(const float* Hin, float* Hout, int* status, int Bt, int n, double eps, int mode, void* stream) {
    return bcbf::launch_clean_hessian<float>(Hin, Hout, status, Bt, n, eps, mode, stream);
}
int bcbf_clean_hessian_f64(const double* Hin, double* Hout, int* status, int Bt, int n, double eps, int mode, void* stream) {
    return bcbf::launch_clean_hessian<double>(Hin, Hout, status, Bt, n, eps, mode, stream);
}
int bcbf_cbc2_terms_f32(const float* Mk, const float* Bk, const float* G, const float* Mj, const float* A,
                        const float* Bm, const float* ell, const float* s2, const float* h, const float* gh,
                        const float* Hh, const float* kalpha, const float* u0, float* out, int* status,
                        int Bt, int n, int m, int hessian_mode, int kernel_kind, void* stream) {
    return bcbf::launch_cbc2_terms<float>(Mk, Bk, G, Mj, A, Bm, ell, s2, h, gh, Hh, kalpha, u0, out, status, Bt, n, m, hessian_mode, kernel_kind, stream);
}
int bcbf_cbc2_terms_f64(const double* Mk, const double* Bk, const double* G, const double* Mj, const double* A,
                        const double* Bm, const double* ell, const double* s2, const double* h, const double* gh,
                        const double* Hh, const double* kalpha, const double* u0, double* out, int* status,
                        int Bt, int n, int m, int hessian_mode, int kernel_kind, void* stream) {
    return bcbf::launch_cbc2_terms<double>(Mk, Bk, G, Mj, A, Bm, ell, s2, h, gh, Hh, kalpha, u0, out, status, Bt, n, m, hessian_mode, kernel_kind, stream);
}
}
