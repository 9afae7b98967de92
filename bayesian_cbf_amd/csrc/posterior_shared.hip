// Regime S: many queries against ONE GP (the reference's own batched API -- custom_predict with b test
// points, control_affine_model.py:536, 1051; Monte-Carlo rollouts of a fixed learned model).
//
// L (<= 1 MB) stays in L2 / Infinity Cache and the work is a triangular solve with b*(1+m) right-hand
// sides: compute bound, so it runs on the matrix cores.  One wave = 8 queries = 32 right-hand-side columns
// (4 per query, the 4th zero for m = 2): blocked forward substitution with v_mfma_f32_32x32x2_f32,
//     acc_I  = Phi_I - sum_{K<I} L_IK W_K          A = -L_IK (one dword per lane from the packed operator,
//                                                   coalesced),  B = W_K rows from LDS (the wave's W slab)
//     W_I    = inv(L_II) acc_I                      A = stored inverse,  B = accumulator registers of acc_I
// and the per-query Gram W'W / mean Vw'W are accumulated from the accumulator registers with quad
// broadcasts (the 4 columns of a query sit in 4 adjacent lanes).  fp32 only; N <= 1024.
#include "bcbf_common.h"

namespace bcbf {

using f32x16s = __attribute__((__vector_size__(16 * sizeof(float)))) float;

__device__ inline int acc_row_s(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
template <int CTRL> __device__ inline float dpp_q(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int C>
__global__ void __launch_bounds__(64)
posterior_shared_kernel(const float* __restrict__ Lop, const float* __restrict__ Vw, const float* __restrict__ X,
                        const float* __restrict__ UHB, const float* __restrict__ ell, const float* __restrict__ s2p,
                        const float* __restrict__ Bm, const float* __restrict__ M0, const float* __restrict__ xq,
                        const float* __restrict__ jitter2, float* __restrict__ Mk, float* __restrict__ Bk,
                        float* __restrict__ Wout, int nq, int N, int Np, int n) {
    constexpr int V = 4, QW = 8;                      // queries per wave
    extern __shared__ float Wl[];                     // [Np][32]: the wave's W slab, row-major
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    const int ql = li >> 2, c = li & 3;               // query slot in the wave, component
    const int q = blockIdx.x * QW + ql;
    const bool qok = q < nq, cok = c < C;
    const int qq = qok ? q : nq - 1;
    const float* __restrict__ lop = Lop;

    float xqr[BCBF_MAX_STATE_DIM], iell[BCBF_MAX_STATE_DIM];
#pragma unroll
    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) {
        xqr[d] = d < n ? xq[(size_t)qq * n + d] : 0.f;
        iell[d] = d < n ? 1.f / ell[d] : 0.f;
    }
    const float s2 = s2p[0];
    float gram[C], mk[BCBF_MAX_STATE_DIM];
#pragma unroll
    for (int a = 0; a < C; ++a) gram[a] = 0.f;
#pragma unroll
    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) mk[d] = 0.f;

    const int nblk = Np / NB;
    for (int I = 0; I < nblk; ++I) {
        const int row0 = I * NB;
        // ---- Phi tile: acc[r] = k(X_row, x_q) UHB[row][c], row = row0 + rho(r) + 4h.  The exp is computed once
        //      per (query,row) by the lane whose component equals r & 3 and broadcast inside the quad.
        f32x16s acc;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int rmine = 4 * r4 + c;                                   // the register this lane evaluates
            const int row_m = row0 + acc_row_s(rmine, lh);
            float kmine = 0.f;
            if (row_m < N) {
                float d2 = 0.f;
#pragma unroll
                for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d)
                    if (d < n) { const float z = (X[(size_t)row_m * n + d] - xqr[d]) * iell[d]; d2 += z * z; }
                kmine = s2 * expf(-0.5f * d2);
            }
            const float k0 = dpp_q<0x00>(kmine), k1 = dpp_q<0x55>(kmine), k2 = dpp_q<0xAA>(kmine), k3 = dpp_q<0xFF>(kmine);
            const float kk[4] = {k0, k1, k2, k3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * r4 + j, row = row0 + acc_row_s(r, lh);
                acc[r] = (cok && row < N) ? kk[j] * UHB[(size_t)row * C + c] : 0.f;
            }
        }
        // ---- acc -= L_IK W_K for K < I, then W_I = inv(L_II) acc  (K == I, fresh accumulator)
        for (int K = 0; K < I; ++K) {
            const float* wk = Wl + (size_t)K * NB * 32;
#pragma unroll 4
            for (int s_ = 0; s_ < NB / 2; ++s_) {
                const float a = -lop[lop_base<V>(K * NB + 2 * s_ + lh, Np) + row0 + li];
                const float bq = wk[(2 * s_ + lh) * 32 + li];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
            }
        }
        f32x16s w = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = lop[lop_base<V>(row0 + acc_row_s(r, lh), Np) + row0 + li];     // inv(L_II)[li][rho(r)+4h]
            w = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[r], w, 0, 0, 0);
        }
        // ---- publish W_I to the slab, accumulate the per-query Gram row and mean column
        float* wi = Wl + (size_t)I * NB * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = acc_row_s(r, lh), row = row0 + rr;
            const float v = w[r];
            wi[rr * 32 + li] = v;
            if (Wout != nullptr && qok && cok) Wout[((size_t)q * Np + row) * C + c] = v;
            const float v0 = dpp_q<0x00>(v), v1 = dpp_q<0x55>(v), v2 = dpp_q<0xAA>(v), v3 = dpp_q<0xFF>(v);
            const float vb[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int a = 0; a < C; ++a) gram[a] += v * vb[a];               // lane c: G[c][a]
            if (row < N) {
#pragma unroll
                for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d)
                    if (d < n) mk[d] += Vw[(size_t)row * n + d] * v;        // lane c: (Vw'W)[d][c]
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): the slab rows are visible to this wave's next reads
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the two lane halves hold different rows: combine, then write Mk[q][d][c], Bk[q][c][a]
#pragma unroll
    for (int a = 0; a < C; ++a) gram[a] += __shfl_xor(gram[a], 32, 64);
#pragma unroll
    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) mk[d] += __shfl_xor(mk[d], 32, 64);
    if (lh == 0 && qok && cok) {
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d)
            if (d < n) Mk[((size_t)q * n + d) * C + c] = M0[c * n + d] + mk[d];
#pragma unroll
        for (int a = 0; a < C; ++a) {
            float v = s2 * Bm[c * C + a] - gram[a];
            if (a == c && jitter2 != nullptr) v += jitter2[(size_t)q * C + c];
            Bk[((size_t)q * C + c) * C + a] = v;
        }
    }
}

}  // namespace bcbf

extern "C" int bcbf_posterior_shared_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                         const float* ell, const float* s2, const float* Bm, const float* M0,
                                         const float* xq, const float* jitter2, float* Mk, float* Bk, float* W,
                                         int nq, int N, int n, int m, void* stream) {
    using namespace bcbf;
    if (nq <= 0) return BCBF_OK;
    if (!Lop || !Vw || !X || !UHB || !ell || !s2 || !Bm || !M0 || !xq || !Mk || !Bk) return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > 3) return BCBF_EINVAL;
    const int Np = round_up(N, NB);
    const size_t lds = (size_t)Np * 32 * sizeof(float);
    if (lds > 160 * 1024) return BCBF_EINVAL;            // N <= 1280: the W slab of a wave lives in LDS
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((nq + 7) / 8), block(64);
#define BCBF_PSH(CC)                                                                                                    \
    do {                                                                                                                \
        if (lds > 64 * 1024)                                                                                            \
            (void)hipFuncSetAttribute((const void*)posterior_shared_kernel<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((posterior_shared_kernel<CC>), grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq,   \
                           jitter2, Mk, Bk, W, nq, N, Np, n);                                                              \
    } while (0)
    switch (m) {
        case 1: BCBF_PSH(2); break;
        case 2: BCBF_PSH(3); break;
        case 3: BCBF_PSH(4); break;
        default: return BCBF_EINVAL;
    }
#undef BCBF_PSH
    return check_launch("posterior_shared");
}
