// K8 (rel-degree 1) + K9: chance-constraint terms of a control barrier / Lyapunov condition and
// their second-order-cone form.  One lane per (instance, constraint); closed form of what the
// reference obtains with autograd through its gp_algebra expression.
#include "bcbf_common.h"

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
}
