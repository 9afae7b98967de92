// K9 for the generic (pendulum-style) controllers of bayes_cbf/controllers.py: rows of the cone program that
// SOCPController / QPController hand to their optimiser, over the variable y = [y_1 (epigraph); rho (relaxation); u]
// (extravars = 2, SOCPController.control :569-591) or y = [rho; u] (extravars = 1, QPController.control :638-662),
// written directly in the solver's layout  G y + s = h,  s in R_+^l x Q^{d_1} x ...  (optimizers.py:6-39:
// a cone |A y + b| <= c'y + d becomes Gq = [-c'; -A], hq = [d; b]).
// One lane per (instance, constraint) plus one lane per instance for the objective cone; fp64 throughout
// (the rows feed bcbf_coneqp_f64).
//   kind 0  "stability":  convert_cbc_terms_to_socp_terms (:423-482)  Asq = L L' (retry with + 1e-3 I),
//                         A[:, ev:] = L'[:, 1:], b = L'[:, 0], c = [.., 1 (at ev-1), bfe], d = e
//   kind 1  "safety":     _socp_safety (:502-540)  A[:, ev:] = L[:, 1:], b = L[:, 0] with the LOWER factor L
//                         (not L': the reference's own choice, kept), scaled by `factor`; if the factorisation
//                         meets a non-positive pivot: L = sqrt(max(Lambda, 0)) V' from the symmetric
//                         eigendecomposition (ascending); c = [0.., bfe], d = e
//   kind 2  "linear":     QPController._qp_stability (:614-629) keeps only (c, d) of kind 0:  0 <= c'y + d
//   objective             _socp_objective (:396-420)  R = [[0, sqrt(lambda), 0], [0, 0, sqrt(Q) I]],
//                         h = [0; -sqrt(Q) u_ref], a = e_0, b = 0
#include "bcbf_common.h"

namespace bcbf {

constexpr int CCK = BCBF_MAX_CONSTRAINTS;
struct ConeSpec {
    int kind[CCK];
    int row[CCK];        // first row of this constraint in G / h
    double factor[CCK];
    int obj_row;         // first row of the objective cone, or -1
};

template <typename T>
__global__ void __launch_bounds__(64)
controller_cones_kernel(const T* __restrict__ terms, const T* __restrict__ u_ref, ConeSpec sp, double ctrl_reg,
                        double relax_weight, int ev, double* __restrict__ G, double* __restrict__ h,
                        int* __restrict__ cstatus, int Bt, int K, int m, int Kt) {
    constexpr int MM = BCBF_MAX_CTRL_DIM, CC = MM + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bt * (K + 1)) return;
    const int b = idx / (K + 1), k = idx - b * (K + 1);
    const int nv = ev + m, C = m + 1;
    double* Gb = G + (size_t)b * Kt * nv;
    double* hb = h + (size_t)b * Kt;
    if (k == K) {                      // objective cone, C + 1 rows
        if (sp.obj_row < 0) return;
        const int r0 = sp.obj_row;
        const double sq = sqrt(ctrl_reg), sl = sqrt(relax_weight);
        for (int r = 0; r < C + 1; ++r) for (int j = 0; j < nv; ++j) Gb[(r0 + r) * nv + j] = 0.0;
        Gb[r0 * nv + 0] = -1.0;                          // -a', a = e_yidx, yidx = 0
        hb[r0] = 0.0;
        Gb[(r0 + 1) * nv + 1] = -sl;                     // -R[0, 1]
        hb[r0 + 1] = 0.0;
        for (int i = 0; i < m; ++i) {
            Gb[(r0 + 2 + i) * nv + ev + i] = -sq;
            hb[r0 + 2 + i] = -sq * (double)u_ref[(size_t)b * m + i];
        }
        return;
    }
    const T* t = terms + ((size_t)b * K + k) * (m + 1 + m * m + m + 1);
    double bfe[MM], e, Asq[CC][CC], L[CC][CC];
    for (int a = 0; a < CC; ++a) for (int c = 0; c < CC; ++c) { Asq[a][c] = 0.0; L[a][c] = 0.0; }
    for (int i = 0; i < MM; ++i) bfe[i] = i < m ? (double)t[i] : 0.0;
    e = (double)t[m];
    Asq[0][0] = (double)t[m + 1 + m * m + m];
    for (int i = 0; i < m; ++i) {
        Asq[0][1 + i] = Asq[1 + i][0] = 0.5 * (double)t[m + 1 + m * m + i];
        for (int j = 0; j < m; ++j) Asq[1 + i][1 + j] = (double)t[m + 1 + i * m + j];
    }
    const int kind = sp.kind[k], r0 = sp.row[k];
    int st = 0;
    if (kind == 2) {
        for (int j = 0; j < nv; ++j) Gb[r0 * nv + j] = 0.0;
        Gb[r0 * nv + ev - 1] = -1.0;
        for (int i = 0; i < m; ++i) Gb[r0 * nv + ev + i] = -bfe[i];
        hb[r0] = e;
        if (cstatus) cstatus[(size_t)b * K + k] = 0;
        return;
    }
    auto chol = [&](double shift) {
        bool ok = true;
        for (int a = 0; a < CC; ++a) for (int c = 0; c < CC; ++c) L[a][c] = 0.0;
        for (int j = 0; j < C; ++j) {
            double d = Asq[j][j] + shift;
            for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
            if (!(d > 0.0)) { ok = false; d = 1.0; }
            const double ljj = sqrt(d);
            L[j][j] = ljj;
            for (int i = j + 1; i < C; ++i) {
                double s = Asq[i][j];
                for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
                L[i][j] = s / ljj;
            }
        }
        return ok;
    };
    bool ok = chol(0.0);
    double factor = 1.0;
    bool transpose = true;               // kind 0 reads L'
    if (kind == 0) {
        if (!ok) ok = chol(1e-3);
        if (!ok) st = BCBF_SOCP_BADCONE;
    } else {
        factor = sp.factor[k];
        transpose = false;               // kind 1 reads L itself
        if (!ok) {                       // L = sqrt(max(Lambda, 0)) V', eigenvalues ascending (torch.symeig)
            double S[CC][CC], V[CC][CC];
            for (int i = 0; i < CC; ++i) for (int j = 0; j < CC; ++j) { S[i][j] = Asq[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
            for (int sweep = 0; sweep < 16; ++sweep)
                for (int p = 0; p < C; ++p)
                    for (int q = p + 1; q < C; ++q) {
                        if (fabs(S[p][q]) < 1e-300) continue;
                        const double th = 0.5 * (S[q][q] - S[p][p]) / S[p][q];
                        const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                        const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                        for (int i = 0; i < C; ++i) { const double a_ = S[i][p], b_ = S[i][q]; S[i][p] = cs * a_ - sn * b_; S[i][q] = sn * a_ + cs * b_; }
                        for (int i = 0; i < C; ++i) { const double a_ = S[p][i], b_ = S[q][i]; S[p][i] = cs * a_ - sn * b_; S[q][i] = sn * a_ + cs * b_; }
                        for (int i = 0; i < C; ++i) { const double a_ = V[i][p], b_ = V[i][q]; V[i][p] = cs * a_ - sn * b_; V[i][q] = sn * a_ + cs * b_; }
                    }
            int ord[CC];
            for (int i = 0; i < CC; ++i) ord[i] = i;
            for (int i = 0; i < C; ++i)                                   // selection sort, ascending
                for (int j = i + 1; j < C; ++j)
                    if (S[ord[j]][ord[j]] < S[ord[i]][ord[i]]) { const int o = ord[i]; ord[i] = ord[j]; ord[j] = o; }
            for (int a = 0; a < C; ++a) {
                const double sa = sqrt(fmax(S[ord[a]][ord[a]], 0.0));
                for (int c = 0; c < C; ++c) L[a][c] = sa * V[c][ord[a]];
            }
        }
    }
    // rows: [-c'; -A], [d; b]
    for (int r = 0; r < C + 1; ++r) for (int j = 0; j < nv; ++j) Gb[(r0 + r) * nv + j] = 0.0;
    if (kind == 0) Gb[r0 * nv + ev - 1] = -1.0;
    for (int i = 0; i < m; ++i) Gb[r0 * nv + ev + i] = -bfe[i];
    hb[r0] = e;
    for (int a = 0; a < C; ++a) {
        for (int i = 0; i < m; ++i) Gb[(r0 + 1 + a) * nv + ev + i] = -factor * (transpose ? L[1 + i][a] : L[a][1 + i]);
        hb[r0 + 1 + a] = factor * (transpose ? L[0][a] : L[a][0]);
    }
    if (cstatus) cstatus[(size_t)b * K + k] = st;
}

template <typename T>
static int launch_controller_cones(const T* terms, const T* u_ref, const int* kind, const double* factor,
                                   double ctrl_reg, double relax_weight, int extravars, int objective, double* G,
                                   double* h, int* cstatus, int Bt, int K, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!G || !h || K < 0 || K > CCK || (K > 0 && (!terms || !kind)) || (objective && !u_ref)) return BCBF_EINVAL;
    if (m < 1 || m > BCBF_MAX_CTRL_DIM || extravars < 1 || extravars > 2 || (objective && extravars != 2)) return BCBF_EINVAL;
    if (K == 0 && !objective) return BCBF_EINVAL;
    ConeSpec sp;
    int row = 0;
    for (int k = 0; k < CCK; ++k) { sp.kind[k] = 0; sp.row[k] = 0; sp.factor[k] = 1.0; }
    for (int k = 0; k < K; ++k) {
        if (kind[k] < 0 || kind[k] > 2) return BCBF_EINVAL;
        sp.kind[k] = kind[k];
        sp.factor[k] = factor ? factor[k] : 1.0;
    }
    for (int k = 0; k < K; ++k) if (kind[k] == 2) sp.row[k] = row++;          // linear rows lead (cvxopt order)
    sp.obj_row = -1;
    if (objective) { sp.obj_row = row; row += m + 2; }
    for (int k = 0; k < K; ++k) if (kind[k] != 2) { sp.row[k] = row; row += m + 2; }
    const int total = Bt * (K + 1);
    hipLaunchKernelGGL((controller_cones_kernel<T>), dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, terms,
                       u_ref, sp, ctrl_reg, relax_weight, extravars, G, h, cstatus, Bt, K, m, row);
    return check_launch("controller_cones");
}

}  // namespace bcbf

extern "C" {
int bcbf_controller_cones_rows(const int* kind, int K, int m, int objective) {
    if (K < 0 || K > bcbf::CCK || (K > 0 && !kind) || m < 1) return BCBF_EINVAL;
    int rows = objective ? m + 2 : 0;
    for (int k = 0; k < K; ++k) {
        if (kind[k] < 0 || kind[k] > 2) return BCBF_EINVAL;       // 0 stability cone, 1 safety cone, 2 linear row
        rows += kind[k] == 2 ? 1 : m + 2;
    }
    return rows;
}
int bcbf_controller_cones_f32(const float* terms, const float* u_ref, const int* kind, const double* factor,
                              double ctrl_reg, double relax_weight, int extravars, int objective, double* G, double* h,
                              int* cstatus, int Bt, int K, int m, void* stream) {
    return bcbf::launch_controller_cones<float>(terms, u_ref, kind, factor, ctrl_reg, relax_weight, extravars, objective,
                                                G, h, cstatus, Bt, K, m, stream);
}
int bcbf_controller_cones_f64(const double* terms, const double* u_ref, const int* kind, const double* factor,
                              double ctrl_reg, double relax_weight, int extravars, int objective, double* G, double* h,
                              int* cstatus, int Bt, int K, int m, void* stream) {
    return bcbf::launch_controller_cones<double>(terms, u_ref, kind, factor, ctrl_reg, relax_weight, extravars,
                                                 objective, G, h, cstatus, Bt, K, m, stream);
}
}
