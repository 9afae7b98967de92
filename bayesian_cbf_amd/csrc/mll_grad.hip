// K12: the O(N^2) part of the marginal-log-likelihood gradient of the matrix-variate GP,
//   log p(Y) = -1/2 tr(A^-1 R' K_b^-1 R) - n/2 logdet K_b - N/2 logdet A - N n/2 log 2 pi,   R = Y - UH M0,
// which the reference obtains from gpytorch's ExactMarginalLogLikelihood + autograd
// (control_affine_model.py:268-335, matrix_variate_multitask_kernel.py:99-204).  With alpha = K_b^-1 R:
//   G = d log p / d K_b = 1/2 (alpha A^-1 alpha' - n K_b^-1),      K_b = s2 k(X,X) o (UH B UH')
//   d/d s2 = sum_ij G_ij k_ij u_ij,   d/d ell_d = sum_ij G_ij K_ij (x_id - x_jd)^2 / ell_d^3,
//   d/d B  = UH' (s2 G o k) UH,       logdet K_b = -2 sum_i log inv(L)_ii   (diagonal blocks are stored inverted)
// plus the two small products R' alpha [n,n] and UH' alpha [C,n] the host needs for d/dA, d/dM0 and the value.
// One workgroup per GP (a fit is one model at a time; N^2 pair terms, nothing to tile).
#include "bcbf_common.h"
#include <stdlib.h>

namespace bcbf {

constexpr int MG_T = 256;
// CM = compile-time bound on the columns of UH: BCBF_MAX_CTRL_DIM + 1 for the matrix-variate model, BCBF_MAX_TASK_DIM
// for the expanded CoGP system with more than four task outputs (a comparator: one workgroup, registers to spare)

// output slot o of the MG_MAXOUT sums of one model -> its place in the gradient arrays
template <typename T, int CM>
__device__ inline void mll_store(int o, double v, int b, int n, int C, T* g_ell, T* g_s2, T* g_B, T* g_lin) {
    constexpr int NR = BCBF_MAX_STATE_DIM + 2 + CM * CM;
    if (o < BCBF_MAX_STATE_DIM) { if (o < n) g_ell[(size_t)b * n + o] = (T)v; }
    else if (o == BCBF_MAX_STATE_DIM) g_s2[b] = (T)v;
    else if (o == NR - 1) { if (g_lin != nullptr) g_lin[b] = (T)v; }
    else {
        const int q = o - BCBF_MAX_STATE_DIM - 1, a = q / CM, c = q % CM;
        if (a < C && c < C) g_B[((size_t)b * C + a) * C + c] = (T)v;
    }
}

// second launch of the split form: the G partial sums of every output, added in the order part = 0 .. G-1
template <typename T, int CM>
__global__ void mll_reduce_kernel(const double* __restrict__ work, int G, int n, int C, T* g_ell, T* g_s2, T* g_B, T* g_lin) {
    constexpr int NR = BCBF_MAX_STATE_DIM + 2 + CM * CM;
    const int b = blockIdx.x, o = threadIdx.x;
    if (o >= NR) return;
    double v = 0.0;
    for (int part = 0; part < G; ++part) v += work[((size_t)b * G + part) * NR + o];
    mll_store<T, CM>(o, v, b, n, C, g_ell, g_s2, g_B, g_lin);
}

template <typename T, int CM>
__global__ void __launch_bounds__(MG_T)
mll_grad_kernel(const T* __restrict__ Lop, const T* __restrict__ alpha, const T* __restrict__ Kinv,
                const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ R, const T* __restrict__ Ainv,
                const T* __restrict__ Bm, const T* __restrict__ ell, const T* __restrict__ s2p,
                T* __restrict__ g_ell, T* __restrict__ g_s2, T* __restrict__ g_B, T* __restrict__ logdetK,
                T* __restrict__ RtA, T* __restrict__ UHtA, int N, int Np, int n, int C, int nt,
                const T* __restrict__ lin, T* __restrict__ g_lin, double* __restrict__ work, int kind) {
    // nt = number of target columns of R / alpha / A (== n for the matrix-variate model; 1 for the expanded
    // CoGP system); lin (optional) = weight of the linear part of the data kernel, k = exp(..) + lin x'x'.
    constexpr int V = Vec<T>::V;
    constexpr int MG_MAXOUT = BCBF_MAX_STATE_DIM + 2 + CM * CM;
    __shared__ double red[4][MG_MAXOUT];
    // grid (Bt, G): the N^2 pair terms of one GP are split over G workgroups (a fit is ONE model: a single workgroup
    // left 255 CUs idle for 2 ms at N = 512).  G > 1: every workgroup writes its partial sums (fp64) to the caller's
    // workspace, work[(b G + part) NR + o], and mll_reduce_kernel adds them IN A FIXED ORDER -- two launches of the same
    // inputs give bit-identical gradients (float atomics, the first form of this split, did not)
    const int b = blockIdx.x, tid = threadIdx.x, part = blockIdx.y, G = gridDim.y;
    const T* Xb = X + (size_t)b * N * n;
    const T* UHb = UH + (size_t)b * N * C;
    const T* Rb = R + (size_t)b * N * nt;
    const T* al = alpha + (size_t)b * N * nt;
    const T* Kib = Kinv + (size_t)b * N * N;
    const T* lop = Lop + (size_t)b * lop_elems<V>(Np);
    double iell[BCBF_MAX_STATE_DIM], Ai[BCBF_MAX_STATE_DIM][BCBF_MAX_STATE_DIM];
    double Bl[CM][CM];
    const double s2 = (double)s2p[b];
    const double linv = lin != nullptr ? (double)lin[b] : 0.0;
    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) {
        iell[d] = d < n ? 1.0 / (double)ell[(size_t)b * n + d] : 0.0;
        for (int e = 0; e < BCBF_MAX_STATE_DIM; ++e) Ai[d][e] = (d < nt && e < nt) ? (double)Ainv[((size_t)b * nt + d) * nt + e] : 0.0;
    }
    for (int a = 0; a < CM; ++a)
        for (int c = 0; c < CM; ++c) Bl[a][c] = (a < C && c < C) ? (double)Bm[((size_t)b * C + a) * C + c] : 0.0;

    // ---- phase 1: the N^2 pair terms (register accumulators, fully unrolled over the compile-time maxima)
    double gl[BCBF_MAX_STATE_DIM], gB[CM * CM], gs = 0.0, glin = 0.0;
#pragma unroll
    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) gl[d] = 0.0;
#pragma unroll
    for (int a = 0; a < CM * CM; ++a) gB[a] = 0.0;
    for (long long idx = (long long)part * MG_T + tid; idx < (long long)N * N; idx += (long long)G * MG_T) {
        const int i = (int)(idx / N), j = (int)(idx - (long long)i * N);
        double d2 = 0.0, dz2[BCBF_MAX_STATE_DIM], dot = 0.0;
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) {
            const double xi = d < n ? (double)Xb[(size_t)i * n + d] : 0.0, xj = d < n ? (double)Xb[(size_t)j * n + d] : 0.0;
            const double z = (xi - xj) * iell[d];
            dz2[d] = z * z;
            d2 += z * z;
            dot += xi * xj;
        }
        double krbf, kder;                                   // kernel shape; its d / d ell_d is kder * z_d^2 / ell_d
        kernel_shape(kind, d2, [](double v) { return exp(v); }, krbf, kder);      // (RBF | Matern-5/2 | their product)
        const double kij = krbf + linv * dot;
        double ui[CM], uj[CM], uij = 0.0;
#pragma unroll
        for (int a = 0; a < CM; ++a) {
            ui[a] = a < C ? (double)UHb[(size_t)i * C + a] : 0.0;
            uj[a] = a < C ? (double)UHb[(size_t)j * C + a] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < CM; ++a) {
            double t = 0.0;
#pragma unroll
            for (int c = 0; c < CM; ++c) t += Bl[a][c] * uj[c];
            uij += ui[a] * t;
        }
        double q = 0.0;                                      // alpha_i' A^-1 alpha_j
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) {
            if (d < nt) {
                double t = 0.0;
#pragma unroll
                for (int e = 0; e < BCBF_MAX_STATE_DIM; ++e)
                    if (e < nt) t += Ai[d][e] * (double)al[(size_t)j * nt + e];
                q += (double)al[(size_t)i * nt + d] * t;
            }
        }
        const double G = 0.5 * (q - (double)nt * (double)Kib[(size_t)i * N + j]);
        const double Gk = G * kij;
        gs += Gk * uij;
        glin += G * s2 * uij * dot;
        const double GK = G * kder * s2 * uij;               // G_ij times the length-scale derivative's shape factor
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) gl[d] += GK * dz2[d] * iell[d];       // z^2 / ell = dx^2 / ell^3
#pragma unroll
        for (int a = 0; a < CM; ++a)
#pragma unroll
            for (int c = 0; c < CM; ++c) gB[a * CM + c] += Gk * s2 * ui[a] * uj[c];
    }
    // workgroup reduction of the 8 + 1 + CM^2 + 1 sums (each wave's total goes straight to LDS)
    constexpr int NR = MG_MAXOUT;
    const bool lead = (tid & 63) == 0;
    auto put = [&](int o, double v) { v = wave_sum(v); if (lead) red[tid >> 6][o] = v; };
    put(NR - 1, glin);
#pragma unroll
    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) put(d, gl[d]);
    put(BCBF_MAX_STATE_DIM, gs);
#pragma unroll
    for (int a = 0; a < CM * CM; ++a) put(BCBF_MAX_STATE_DIM + 1 + a, gB[a]);
    __syncthreads();
    if (tid < NR) {
        const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (G > 1) work[((size_t)b * G + part) * NR + tid] = v;
        else mll_store<T, CM>(tid, v, b, n, C, g_ell, g_s2, g_B, g_lin);
    }
    if (part != 0) return;
    // ---- phase 2: the small products (one output per thread, a loop over the N rows) and logdet (last wave)
    if (tid < nt * nt) {
        const int d = tid / nt, e = tid - d * nt;
        double v = 0.0;
        for (int i = 0; i < N; ++i) v += (double)Rb[(size_t)i * nt + d] * (double)al[(size_t)i * nt + e];
        RtA[(size_t)b * nt * nt + tid] = (T)v;
    } else if (tid >= 64 && tid < 64 + C * nt) {
        const int o = tid - 64, a = o / nt, d = o - a * nt;
        double v = 0.0;
        for (int i = 0; i < N; ++i) v += (double)UHb[(size_t)i * C + a] * (double)al[(size_t)i * nt + d];
        UHtA[(size_t)b * C * nt + o] = (T)v;
    } else if (tid >= 192) {
        double v = 0.0;
        for (int i = tid - 192; i < N; i += 64) v -= 2.0 * log((double)lop[lop_dinv(i / NB, i % NB, i % NB, Np)]);
        v = wave_sum(v);
        if (tid == 192) logdetK[b] = (T)v;
    }
}

// Batches (round 6): the same sums with one THREAD PER ROW i and the columns j streamed through LDS in tiles.  The form above spends
// ~300 fp64 operations per pair (every loop unrolled over the compile-time maxima -- 8 state dimensions, an 8 x 8 product with
// A^-1 and a 4 x 4 one with B PER PAIR) and reads K_b^-1 row-wise per thread; 4096 models of 512 points took 27 ms (fp64) / 51 ms
// (fp32 inputs), more than the factorisation.  Here the per-ROW quantities are formed once per tile -- t_j = A^-1 alpha_j,
// v_j = B uh_j -- so a pair costs  q = alpha_i . t_j,  u_ij = uh_i . v_j,  the kernel value and  w_i += G k uh_j  (g_B = s2
// sum_i uh_i w_i' afterwards); K_b^-1 is read through its symmetry, element (j, i), so consecutive threads read consecutive
// addresses -- and, being symmetric in (i, j) term by term, only the pairs j <= i are visited (a wave of rows stops at its last row:
// 0.56 of the pair work at N = 512, half of K_b^-1's bytes).  Grid (Bt, RC * JS): RC = ceil(N / 256) row chunks times JS column slices (JS > 1 only for a handful of models);
// partial sums go to the caller's workspace and mll_reduce_kernel adds them in a fixed order (bit-identical run to run).
// n, nt <= 4 and C <= 4 (every matrix-variate model the device path takes); no linear kernel part.
constexpr int MR_TJ = 128;
#ifndef BCBF_MR_U32
#define BCBF_MR_U32 2
#endif
#ifndef BCBF_MR_U64
#define BCBF_MR_U64 1
#endif
#ifndef BCBF_MR_OCC32
#define BCBF_MR_OCC32 7
#endif
#ifndef BCBF_MR_OCC64
#define BCBF_MR_OCC64 4
#endif
// pairs per group of the row form's loop (MR_TJ is a multiple) and the waves per SIMD its registers are held to.  The loop is a chain
// of dependent operations per pair: waves, not unrolling, fill the SIMD.  4096 x 512, ms (U, waves): fp32 (4, 4) 1.42, (2, 5) 1.34,
// (2, 6) 1.30, (2, 7) 1.24, (2, 8) 2.21 (spills), (1, 8) 1.30, (8, 4) 8.6; fp64 (1, 4) 2.63, (2, 3) 2.71, (4, 2) 3.14, (2, 4) 6.4 (spills),
// (1, 5) 7.4.  Whatever the compiler picks without a bound: fp32 1.65, fp64 4.8 (256 registers, one wave per SIMD)
template <typename T> struct MRows { static constexpr int U = BCBF_MR_U32, OCC = BCBF_MR_OCC32; };
template <> struct MRows<double> { static constexpr int U = BCBF_MR_U64, OCC = BCBF_MR_OCC64; };
__device__ inline float mll_exp(float v) { return __expf(v); }
__device__ inline double mll_exp(double v) { return exp(v); }              // (exp_neg64 of bcbf_common.h measured SLOWER here: 4.5 against 3.75 ms)
template <typename T>
__global__ void __launch_bounds__(MG_T, MRows<T>::OCC)
mll_grad_rows_kernel(const T* __restrict__ Lop, const T* __restrict__ alpha, const T* __restrict__ Kinv,
                     const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ R, const T* __restrict__ Ainv,
                     const T* __restrict__ Bm, const T* __restrict__ ell, const T* __restrict__ s2p,
                     T* __restrict__ logdetK, T* __restrict__ RtA, T* __restrict__ UHtA, int N, int Np, int n, int C, int nt,
                     double* __restrict__ work, int kind, int RC) {
    constexpr int V = Vec<T>::V;
    constexpr int NS = 4, CM = BCBF_MAX_CTRL_DIM + 1, MR_U = MRows<T>::U;
    constexpr int NR = BCBF_MAX_STATE_DIM + 2 + CM * CM;
    // the pair loop's arithmetic type: fp32 inputs are summed in fp32 (K_b^-1 itself carries cond x 6e-8 there; an fp64 pair loop costs
    // 4096 x 512: 5.5 ms against the fp64 kernel's 3.7 -- conversions and the double exp), fp64 inputs in fp64
    using M = T;
    __shared__ M xj[MR_TJ][NS], uj[MR_TJ][CM], vj[MR_TJ][CM], tj[MR_TJ][NS];
    __shared__ double red[4][NR];
    const int b = blockIdx.x, tid = threadIdx.x, part = blockIdx.y, G = gridDim.y;
    const int rc = part % RC, js = part / RC, JS = G / RC;
    const T* Xb = X + (size_t)b * N * n;
    const T* UHb = UH + (size_t)b * N * C;
    const T* Rb = R + (size_t)b * N * nt;
    const T* al = alpha + (size_t)b * N * nt;
    const T* Kib = Kinv + (size_t)b * N * N;
    const T* lop = Lop + (size_t)b * lop_elems<V>(Np);
    M iell[NS], Ai[NS][NS], Bl[CM][CM];
    const M s2 = (M)s2p[b];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        iell[d] = d < n ? M(1.0) / (M)ell[(size_t)b * n + d] : M(0.0);
#pragma unroll
        for (int e = 0; e < NS; ++e) Ai[d][e] = (d < nt && e < nt) ? (M)Ainv[((size_t)b * nt + d) * nt + e] : M(0.0);
    }
#pragma unroll
    for (int a = 0; a < CM; ++a)
#pragma unroll
        for (int c = 0; c < CM; ++c) Bl[a][c] = (a < C && c < C) ? (M)Bm[((size_t)b * C + a) * C + c] : M(0.0);
    // this thread's row
    const int i = rc * MG_T + tid;
    const bool vi = i < N;
    M xi[NS], ui[CM], ai[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        xi[d] = (vi && d < n) ? (M)Xb[(size_t)i * n + d] : M(0.0);
        ai[d] = (vi && d < nt) ? (M)al[(size_t)i * nt + d] : M(0.0);
    }
#pragma unroll
    for (int a = 0; a < CM; ++a) ui[a] = (vi && a < C) ? (M)UHb[(size_t)i * C + a] : M(0.0);
    M gl[NS], w[CM], gs = M(0.0);
#pragma unroll
    for (int d = 0; d < NS; ++d) gl[d] = M(0.0);
#pragma unroll
    for (int c = 0; c < CM; ++c) w[c] = M(0.0);
    // this workgroup's columns: slice js of JS, in tiles of MR_TJ -- the pairs j <= i only: every term of the sums is the same for
    // (i, j) and (j, i) up to the transposition of uh_i' B uh_j, so a pair below the diagonal carries both (weight 1 on the sum of
    // the two), the diagonal half of that; a wave stops at its last row
    const int per = ((N + JS - 1) / JS + MR_TJ - 1) / MR_TJ * MR_TJ;
    const int jbeg = js * per, jend = min(min(N, jbeg + per), rc * MG_T + MG_T);
    const int jlast = rc * MG_T + (tid | 63);       // the wave's last row
    for (int j0 = jbeg; j0 < jend; j0 += MR_TJ) {
        __syncthreads();
        if (tid < MR_TJ) {
            const int j = j0 + tid;
            const bool vj_ = j < jend;
            M a_[NS], u_[CM];
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                xj[tid][d] = (vj_ && d < n) ? (M)Xb[(size_t)j * n + d] : M(0.0);
                a_[d] = (vj_ && d < nt) ? (M)al[(size_t)j * nt + d] : M(0.0);
            }
#pragma unroll
            for (int a = 0; a < CM; ++a) u_[a] = (vj_ && a < C) ? (M)UHb[(size_t)j * C + a] : M(0.0);
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                M t = M(0.0);
#pragma unroll
                for (int e = 0; e < NS; ++e) t += Ai[d][e] * a_[e];
                tj[tid][d] = t;
            }
#pragma unroll
            for (int a = 0; a < CM; ++a) {
                M t = M(0.0);
#pragma unroll
                for (int c = 0; c < CM; ++c) t += (Bl[a][c] + Bl[c][a]) * u_[c];       // (B + B') uh_j: uh_i' B uh_j + uh_j' B uh_i in one product
                vj[tid][a] = t;
                uj[tid][a] = u_[a];
            }
        }
        __syncthreads();
        const int cnt = min(min(MR_TJ, jend - j0), jlast - j0 + 1);
        // K_b^-1's elements are fetched MR_U pairs ahead (one load per pair, ~100 operations between its issue and its use otherwise).  The pairs of a group beyond `cnt` meet zero-filled
        // columns or weight 0; their addresses are clamped into the tile
        const int jcl = min(MR_TJ, jend - j0) - 1;
        if (vi) {
          M kv[MR_U], kn[MR_U];
          const T* kp = Kib + (size_t)j0 * N + i;                      // K_b^-1 [j][i] = [i][j]
#pragma unroll
          for (int u = 0; u < MR_U; ++u) kv[u] = (M)kp[(size_t)min(u, jcl) * N];
          for (int jg = 0; jg < cnt; jg += MR_U) {
#pragma unroll
            for (int u = 0; u < MR_U; ++u) kn[u] = (M)kp[(size_t)min(jg + MR_U + u, jcl) * N];
#pragma unroll
            for (int u = 0; u < MR_U; ++u) {
                const int jj = jg + u, j = j0 + jj;
                const M wgt = j < i ? M(1.0) : (j == i ? M(0.5) : M(0.0));
                const M kinv = kv[u];
                M d2 = M(0.0), dz2[NS];
#pragma unroll
                for (int d = 0; d < NS; ++d) {
                    const M z = (xi[d] - xj[jj][d]) * iell[d];
                    dz2[d] = z * z;
                    d2 += z * z;
                }
                M krbf, kder;
                kernel_shape(kind, d2, [](M v) { return mll_exp(v); }, krbf, kder);
                M uij = M(0.0), q = M(0.0);
#pragma unroll
                for (int a = 0; a < CM; ++a) uij += ui[a] * vj[jj][a];                            // uh_i' B uh_j + uh_j' B uh_i
#pragma unroll
                for (int d = 0; d < NS; ++d) q += ai[d] * tj[jj][d];
                const M Gm = wgt * M(0.5) * (q - (M)nt * kinv);
                const M Gk = Gm * krbf;
                gs += Gk * uij;
                const M GK = Gm * kder * s2 * uij;
#pragma unroll
                for (int d = 0; d < NS; ++d) gl[d] += GK * dz2[d];                 // (x 1 / l_d after the loop)
#pragma unroll
                for (int c = 0; c < CM; ++c) w[c] += Gk * uj[jj][c];              // M = sum_{j <= i} G k uh_i uh_j';  g_B = s2 (M + M')
            }
#pragma unroll
            for (int u = 0; u < MR_U; ++u) kv[u] = kn[u];
          }
        }
    }
    // workgroup sums into the NR slots of mll_store's layout
    const bool lead = (tid & 63) == 0;
    auto put = [&](int o, double v) { v = wave_sum(v); if (lead) red[tid >> 6][o] = v; };
#pragma unroll
    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) put(d, d < NS ? (double)(gl[d < NS ? d : 0] * iell[d < NS ? d : 0]) : 0.0);
    put(BCBF_MAX_STATE_DIM, (double)gs);
#pragma unroll
    for (int a = 0; a < CM; ++a)
#pragma unroll
        for (int c = 0; c < CM; ++c) put(BCBF_MAX_STATE_DIM + 1 + a * CM + c, (double)(s2 * (ui[a] * w[c] + ui[c] * w[a])));
    put(NR - 1, 0.0);
    __syncthreads();
    if (tid < NR) work[((size_t)b * G + part) * NR + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (part != 0) return;
    // ---- the small products and logdet (as in the form above)
    if (tid < nt * nt) {
        const int d = tid / nt, e = tid - d * nt;
        double v = 0.0;
        for (int r = 0; r < N; ++r) v += (double)Rb[(size_t)r * nt + d] * (double)al[(size_t)r * nt + e];
        RtA[(size_t)b * nt * nt + tid] = (T)v;
    } else if (tid >= 64 && tid < 64 + C * nt) {
        const int o = tid - 64, a = o / nt, d = o - a * nt;
        double v = 0.0;
        for (int r = 0; r < N; ++r) v += (double)UHb[(size_t)r * C + a] * (double)al[(size_t)r * nt + d];
        UHtA[(size_t)b * C * nt + o] = (T)v;
    } else if (tid >= 192) {
        double v = 0.0;
        for (int r = tid - 192; r < N; r += 64) v -= 2.0 * log((double)lop[lop_dinv(r / NB, r % NB, r % NB, Np)]);
        v = wave_sum(v);
        if (tid == 192) logdetK[b] = (T)v;
    }
}

// the row form's grid: RC row chunks x JS column slices (JS > 1 for a handful of models only)
static void mll_rows_split(int Bt, int N, int& RC, int& JS) {
    RC = (N + MG_T - 1) / MG_T;
    JS = 1;
    if (Bt < 64) {
        const int tiles = (N + MR_TJ - 1) / MR_TJ;
        JS = 128 / RC;
        if (JS > tiles) JS = tiles;
        if (JS < 1) JS = 1;
    }
}

// workgroups per model of the split form (0 < Bt < 64 models: ~32 pair terms per thread, at most 128 workgroups)
static int mll_split(int Bt, int N) {
    if (Bt >= 64) return 1;
    const long long per_wg = (long long)MG_T * 32;
    long long G = ((long long)N * N + per_wg - 1) / per_wg;
    return (int)(G > 128 ? 128 : (G < 1 ? 1 : G));
}

template <typename T>
static int launch_mll_grad(const T* Lop, const T* alpha, const T* Kinv, const T* X, const T* UH, const T* R,
                           const T* Ainv, const T* Bm, const T* ell, const T* s2, T* g_ell, T* g_s2, T* g_B, T* logdetK,
                           T* RtA, T* UHtA, int Bt, int N, int n, int m, void* stream, void* work, int nt = -1,
                           const T* lin = nullptr, T* g_lin = nullptr, int kind = 0) {
    if (nt < 0) nt = n;
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !alpha || !Kinv || !X || !UH || !R || !Ainv || !Bm || !ell || !s2 || !g_ell || !g_s2 || !g_B || !logdetK ||
        !RtA || !UHtA)
        return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m + 1 > BCBF_MAX_TASK_DIM || nt < 1 || nt > BCBF_MAX_STATE_DIM)
        return BCBF_EINVAL;
    if ((m + 1) * nt > 128) return BCBF_EINVAL;                    // phase 2: one thread per entry of UH' alpha
    // few models: spread each one's pair terms over G workgroups (needs the caller's workspace for the partial sums;
    // without one the model stays on one workgroup)
    hipStream_t st = (hipStream_t)stream;
    static const bool pairs_form = [] { const char* e = getenv("BCBF_MLL_PAIRS"); return e && e[0] == '1'; }();   // (development)
    if (work != nullptr && lin == nullptr && n <= 4 && nt <= 4 && m <= BCBF_MAX_CTRL_DIM && !pairs_form) {
        constexpr int CM = BCBF_MAX_CTRL_DIM + 1;
        int RC, JS;
        mll_rows_split(Bt, N, RC, JS);
        hipLaunchKernelGGL((mll_grad_rows_kernel<T>), dim3(Bt, RC * JS), dim3(MG_T), 0, st, Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell,
                           s2, logdetK, RtA, UHtA, N, round_up(N, NB), n, m + 1, nt, (double*)work, kind, RC);
        hipLaunchKernelGGL((mll_reduce_kernel<T, CM>), dim3(Bt), dim3(64), 0, st, (const double*)work, RC * JS, n, m + 1, g_ell,
                           g_s2, g_B, g_lin);
        return check_launch("mll_grad");
    }
    const int G = work != nullptr ? mll_split(Bt, N) : 1;
    if (m <= BCBF_MAX_CTRL_DIM) {
        constexpr int CM = BCBF_MAX_CTRL_DIM + 1;
        hipLaunchKernelGGL((mll_grad_kernel<T, CM>), dim3(Bt, G), dim3(MG_T), 0, st, Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell,
                           s2, g_ell, g_s2, g_B, logdetK, RtA, UHtA, N, round_up(N, NB), n, m + 1, nt, lin, g_lin, (double*)work, kind);
        if (G > 1)
            hipLaunchKernelGGL((mll_reduce_kernel<T, CM>), dim3(Bt), dim3(64), 0, st, (const double*)work, G, n, m + 1, g_ell,
                               g_s2, g_B, g_lin);
    } else {
        constexpr int CM = BCBF_MAX_TASK_DIM;
        hipLaunchKernelGGL((mll_grad_kernel<T, CM>), dim3(Bt, G), dim3(MG_T), 0, st, Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell,
                           s2, g_ell, g_s2, g_B, logdetK, RtA, UHtA, N, round_up(N, NB), n, m + 1, nt, lin, g_lin, (double*)work, kind);
        if (G > 1)
            hipLaunchKernelGGL((mll_reduce_kernel<T, CM>), dim3(Bt), dim3(192), 0, st, (const double*)work, G, n, m + 1, g_ell,
                               g_s2, g_B, g_lin);
    }
    return check_launch("mll_grad");
}

}  // namespace bcbf

extern "C" {
// bytes of the workspace `work` of bcbf_mll_grad* (partial sums of the split form; 0 = one workgroup per model anyway)
size_t bcbf_mll_grad_work_bytes(int Bt, int N, int m) {
    if (Bt <= 0 || N < 1 || m < 1) return 0;
    int G = bcbf::mll_split(Bt, N);
    const int CM = m <= BCBF_MAX_CTRL_DIM ? BCBF_MAX_CTRL_DIM + 1 : BCBF_MAX_TASK_DIM;
    if (m <= BCBF_MAX_CTRL_DIM) {                       // the row form (batches; n <= 4): one slot set per row chunk x column slice
        int RC, JS;
        bcbf::mll_rows_split(Bt, N, RC, JS);
        if (RC * JS > G) G = RC * JS;
        return sizeof(double) * (size_t)Bt * G * (BCBF_MAX_STATE_DIM + 2 + CM * CM);
    }
    return G > 1 ? sizeof(double) * (size_t)Bt * G * (BCBF_MAX_STATE_DIM + 2 + CM * CM) : 0;
}
int bcbf_mll_grad_f32(const float* Lop, const float* alpha, const float* Kinv, const float* X, const float* UH,
                      const float* R, const float* Ainv, const float* Bm, const float* ell, const float* s2, float* g_ell,
                      float* g_s2, float* g_B, float* logdetK, float* RtA, float* UHtA, int Bt, int N, int n, int m,
                      void* work, void* stream) {
    return bcbf::launch_mll_grad<float>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                        UHtA, Bt, N, n, m, stream, work);
}
int bcbf_mll_grad_f64(const double* Lop, const double* alpha, const double* Kinv, const double* X, const double* UH,
                      const double* R, const double* Ainv, const double* Bm, const double* ell, const double* s2,
                      double* g_ell, double* g_s2, double* g_B, double* logdetK, double* RtA, double* UHtA, int Bt, int N,
                      int n, int m, void* work, void* stream) {
    return bcbf::launch_mll_grad<double>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                         UHtA, Bt, N, n, m, stream, work);
}
// ... for the opt-in Matern-5/2 data kernel (the reference has none; bcbf_kb_build_matern52)
int bcbf_mll_grad_matern52_f32(const float* Lop, const float* alpha, const float* Kinv, const float* X, const float* UH,
                               const float* R, const float* Ainv, const float* Bm, const float* ell, const float* s2, float* g_ell,
                               float* g_s2, float* g_B, float* logdetK, float* RtA, float* UHtA, int Bt, int N, int n, int m,
                               void* work, void* stream) {
    return bcbf::launch_mll_grad<float>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                        UHtA, Bt, N, n, m, stream, work, -1, nullptr, nullptr, 1);
}
int bcbf_mll_grad_matern52_f64(const double* Lop, const double* alpha, const double* Kinv, const double* X, const double* UH,
                               const double* R, const double* Ainv, const double* Bm, const double* ell, const double* s2,
                               double* g_ell, double* g_s2, double* g_B, double* logdetK, double* RtA, double* UHtA, int Bt, int N,
                               int n, int m, void* work, void* stream) {
    return bcbf::launch_mll_grad<double>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                         UHtA, Bt, N, n, m, stream, work, -1, nullptr, nullptr, 1);
}
int bcbf_mll_grad_rbfm52_f32(const float* Lop, const float* alpha, const float* Kinv, const float* X, const float* UH,
                             const float* R, const float* Ainv, const float* Bm, const float* ell, const float* s2, float* g_ell,
                             float* g_s2, float* g_B, float* logdetK, float* RtA, float* UHtA, int Bt, int N, int n, int m,
                             void* work, void* stream) {
    return bcbf::launch_mll_grad<float>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                        UHtA, Bt, N, n, m, stream, work, -1, nullptr, nullptr, 2);
}
int bcbf_mll_grad_rbfm52_f64(const double* Lop, const double* alpha, const double* Kinv, const double* X, const double* UH,
                             const double* R, const double* Ainv, const double* Bm, const double* ell, const double* s2,
                             double* g_ell, double* g_s2, double* g_B, double* logdetK, double* RtA, double* UHtA, int Bt, int N,
                             int n, int m, void* work, void* stream) {
    return bcbf::launch_mll_grad<double>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                         UHtA, Bt, N, n, m, stream, work, -1, nullptr, nullptr, 2);
}
// Same sums for the data kernel s2 (exp(..) + lin x'x') and nt target columns (R, alpha [Bt,N,nt], Ainv [Bt,nt,nt],
// RtA [Bt,nt,nt], UHtA [Bt,C,nt]); g_lin[Bt] = d log p / d lin.  nt = 1 with expanded inputs is the CoGP comparator
// (ControlAffineRegressorVector.fit, control_affine_model.py:268-335 on ControlAffineVectorGP :1106-1126).
int bcbf_mll_grad_rbflin_f32(const float* Lop, const float* alpha, const float* Kinv, const float* X, const float* UH,
                             const float* R, const float* Ainv, const float* Bm, const float* ell, const float* s2,
                             const float* lin, float* g_ell, float* g_s2, float* g_lin, float* g_B, float* logdetK,
                             float* RtA, float* UHtA, int Bt, int N, int n, int m, int nt, void* work, void* stream) {
    return bcbf::launch_mll_grad<float>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                        UHtA, Bt, N, n, m, stream, work, nt, lin, g_lin);
}
int bcbf_mll_grad_rbflin_f64(const double* Lop, const double* alpha, const double* Kinv, const double* X,
                             const double* UH, const double* R, const double* Ainv, const double* Bm, const double* ell,
                             const double* s2, const double* lin, double* g_ell, double* g_s2, double* g_lin,
                             double* g_B, double* logdetK, double* RtA, double* UHtA, int Bt, int N, int n, int m,
                             int nt, void* work, void* stream) {
    return bcbf::launch_mll_grad<double>(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, g_ell, g_s2, g_B, logdetK, RtA,
                                         UHtA, Bt, N, n, m, stream, work, nt, lin, g_lin);
}
}
