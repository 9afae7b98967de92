// Gram of the whitened cross-covariances on the shared-GP prediction path, on the matrix cores, and the prediction call in ONE
// entry point.
//
//   G[b, p, c, d] = sum_k W[b, k, c] Wp[p, k, d]        W [b, Np, C], Wp [bp, Np, C]  (W = L^-1 Phi: bcbf_posterior_query, want W)
//
// is `v.t() @ vp` of ControlAffineRegressor.custom_predict (control_affine_model.py:586 of the reference) for C = 1 and
// `kb_star' Bdagger` of ControlAffineRegressorExact._custom_predict_matrix (:1079-1088) for C = 1 + m -- a [bC x Np] . [Np x b'C]
// product that rounds 1-5 left to a library GEMM.  One wave per 32 x 32 output tile; both MFMA operands are read straight from
// W (row (q, c) of the operand = the strided column c of query q's slab; the slabs are cache resident: 400 queries x 512 points
// x 2 columns = 1.6 MB in fp32), v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 as in syrk.hip.  W == Wp: only the tiles on and
// below the diagonal are computed, the mirror tile is written from the same accumulators.
//
// bcbf_predict_fullmat: query (shared model) -> Gram -> bcbf_predict_assemble on one stream, one host call: what
// ControlAffineRegressorExact.custom_predict_fullmat (:963-980) needs behind a cached factor.
#include "bcbf_common.h"

namespace bcbf {

constexpr int GR_DEPTH = 8;

// tile index t -> (I, J): the full R x S grid, or J <= I when symmetric
__device__ inline void gram_tile(int t, int tilesS, bool sym, int& I, int& J) {
    if (sym) {
        I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        J = t - I * (I + 1) / 2;
    } else {
        I = t / tilesS;
        J = t - I * tilesS;
    }
}

__global__ void __launch_bounds__(64)
gram_kernel_f32(const float* __restrict__ W, const float* __restrict__ Wp, float* __restrict__ G, int b, int bp, int Np, int C,
                int tilesS, int sym) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    int I, J;
    gram_tile(blockIdx.x, tilesS, sym != 0, I, J);
    const int R = b * C, S = bp * C;
    const int lane = threadIdx.x, col = lane & 31, kk = lane >> 5;
    const int r = I * 32 + col, s = J * 32 + col;                    // this lane's operand rows: (query, column) = (r / C, r % C)
    const bool vr = r < R, vs = s < S;
    const float* a_ = W + (vr ? (size_t)(r / C) * Np * C + (r % C) : 0);
    const float* b_ = Wp + (vs ? (size_t)(s / C) * Np * C + (s % C) : 0);
    f32x16 acc = {0};
    for (int k0 = 0; k0 < Np; k0 += 2 * GR_DEPTH) {
        float av[GR_DEPTH], bv[GR_DEPTH];
#pragma unroll
        for (int q = 0; q < GR_DEPTH; ++q) {
            const int k = k0 + 2 * q + kk;
            av[q] = (vr && k < Np) ? a_[(size_t)k * C] : 0.0f;
            bv[q] = (vs && k < Np) ? b_[(size_t)k * C] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < GR_DEPTH; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    }
    // accumulator: lane (column j = lane % 32, group g = lane / 32), register q -> row i = 8 (q / 4) + 4 g + q % 4
    const int sp = s / C, sd = s % C;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int gi = I * 32 + 8 * (q >> 2) + 4 * kk + (q & 3);
        if (gi < R && vs) {
            const int rq = gi / C, rc = gi % C;
            G[(((size_t)rq * bp + sp) * C + rc) * C + sd] = acc[q];
            if (sym && I != J) G[(((size_t)sp * bp + rq) * C + sd) * C + rc] = acc[q];
        }
    }
}

__global__ void __launch_bounds__(64)
gram_kernel_f64(const double* __restrict__ W, const double* __restrict__ Wp, double* __restrict__ G, int b, int bp, int Np, int C,
                int tilesS, int sym) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    int I, J;
    gram_tile(blockIdx.x, tilesS, sym != 0, I, J);
    const int R = b * C, S = bp * C;
    const int lane = threadIdx.x, col = lane & 15, kk = lane >> 4;   // v_mfma_f64_16x16x4: lane = (row / column, k of 4)
    const double *a_[2], *b_[2];
    bool vr[2], vs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = I * 32 + 16 * h + col, s = J * 32 + 16 * h + col;
        vr[h] = r < R; vs[h] = s < S;
        a_[h] = W + (vr[h] ? (size_t)(r / C) * Np * C + (r % C) : 0);
        b_[h] = Wp + (vs[h] ? (size_t)(s / C) * Np * C + (s % C) : 0);
    }
    f64x4 acc[2][2] = {};
    for (int k0 = 0; k0 < Np; k0 += 4 * (GR_DEPTH / 2)) {
        double av[GR_DEPTH / 2][2], bv[GR_DEPTH / 2][2];
#pragma unroll
        for (int q = 0; q < GR_DEPTH / 2; ++q) {
            const int k = k0 + 4 * q + kk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                av[q][h] = (vr[h] && k < Np) ? a_[h][(size_t)k * C] : 0.0;
                bv[q][h] = (vs[h] && k < Np) ? b_[h][(size_t)k * C] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < GR_DEPTH / 2; ++q)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int hj = 0; hj < 2; ++hj)
                    acc[hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][hi], bv[q][hj], acc[hi][hj], 0, 0, 0);
    }
    // accumulator of one 16 x 16 tile: lane (column j = lane % 16, group g = lane / 16), register q -> row i = g + 4 q
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int hj = 0; hj < 2; ++hj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gi = I * 32 + 16 * hi + kk + 4 * q, gj = J * 32 + 16 * hj + col;
                if (gi < R && gj < S) {
                    const int rq = gi / C, rc = gi % C, sp = gj / C, sd = gj % C;
                    G[(((size_t)rq * bp + sp) * C + rc) * C + sd] = acc[hi][hj][q];
                    if (sym && I != J) G[(((size_t)sp * bp + rq) * C + sd) * C + rc] = acc[hi][hj][q];
                }
            }
}

static int gram_grid(int b, int bp, int C, bool sym, int& tilesS) {
    const int tr = (b * C + 31) / 32;
    tilesS = (bp * C + 31) / 32;
    return sym ? tr * (tr + 1) / 2 : tr * tilesS;
}
static bool gram_args_ok(const void* W, const void* Wp, const void* G, int b, int bp, int Np, int C) {
    return W && Wp && G && G != W && G != Wp && Np >= 1 && C >= 1 && C <= BCBF_MAX_TASK_DIM &&
           (long long)b * C < (1ll << 24) && (long long)bp * C < (1ll << 24);
}

}  // namespace bcbf

extern "C" int bcbf_gram_f32(const float* W, const float* Wp, float* G, int b, int bp, int Np, int C, void* stream) {
    if (b <= 0 || bp <= 0) return BCBF_OK;
    if (!bcbf::gram_args_ok(W, Wp, G, b, bp, Np, C)) return BCBF_EINVAL;
    const bool sym = W == Wp && b == bp;
    int tilesS;
    const int tiles = bcbf::gram_grid(b, bp, C, sym, tilesS);
    hipLaunchKernelGGL(bcbf::gram_kernel_f32, dim3(tiles), dim3(64), 0, (hipStream_t)stream, W, Wp, G, b, bp, Np, C, tilesS, sym ? 1 : 0);
    return bcbf::check_launch("bcbf_gram");
}
extern "C" int bcbf_gram_f64(const double* W, const double* Wp, double* G, int b, int bp, int Np, int C, void* stream) {
    if (b <= 0 || bp <= 0) return BCBF_OK;
    if (!bcbf::gram_args_ok(W, Wp, G, b, bp, Np, C)) return BCBF_EINVAL;
    const bool sym = W == Wp && b == bp;
    int tilesS;
    const int tiles = bcbf::gram_grid(b, bp, C, sym, tilesS);
    hipLaunchKernelGGL(bcbf::gram_kernel_f64, dim3(tiles), dim3(64), 0, (hipStream_t)stream, W, Wp, G, b, bp, Np, C, tilesS, sym ? 1 : 0);
    return bcbf::check_launch("bcbf_gram");
}

// query -> Gram -> assembly, one host call (custom_predict_fullmat behind a cached factor)
#define BCBF_PREDICT_FULLMAT(T, SUF)                                                                                              \
    extern "C" int bcbf_predict_fullmat_##SUF(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2,     \
                                              const T* Bm, const T* M0, const T* A, const T* Xq, const T* jitter, T* Mk, T* Bk,   \
                                              T* W, T* G, T* BkXX, T* Kron, int b, int N, int n, int m, int kernel_kind,          \
                                              void* stream) {                                                                     \
        if (b <= 0) return BCBF_OK;                                                                                               \
        if (!Lop || !Vw || !X || !UHB || !ell || !s2 || !Bm || !M0 || !Xq || !Mk || !Bk || !W || !G || (!BkXX && !Kron) ||        \
            (Kron && !A))                                                                                                         \
            return BCBF_EINVAL;                                                                                                   \
        int rc;                                                                                                                   \
        if (kernel_kind == 0) rc = bcbf_posterior_query_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, Xq, nullptr, Mk, Bk, W, 1, b, N, n, m, stream); \
        else if (kernel_kind == 1) rc = bcbf_posterior_query_matern52_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, Xq, nullptr, Mk, Bk, W, 1, b, N, n, m, stream); \
        else if (kernel_kind == 2) rc = bcbf_posterior_query_rbfm52_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, Xq, nullptr, Mk, Bk, W, 1, b, N, n, m, stream); \
        else return BCBF_EINVAL;                                                                                                  \
        if (rc) return rc;                                                                                                        \
        rc = bcbf_gram_##SUF(W, W, G, b, b, bcbf::round_up(N, bcbf::NB), m + 1, stream);                                          \
        if (rc) return rc;                                                                                                        \
        return bcbf_predict_assemble_##SUF(G, Xq, Xq, ell, s2, Bm, A, jitter, BkXX, Kron, b, b, n, m, kernel_kind, stream);       \
    }
BCBF_PREDICT_FULLMAT(float, f32)
BCBF_PREDICT_FULLMAT(double, f64)
