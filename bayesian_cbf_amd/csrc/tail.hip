// Online path, window mode with a ROW-MAJOR TAIL (round 5): between two window refits the factor of the window's
// N0 points stays as the refit laid it out, and the points observed since are the rows of a bordered factor
//     [ L0   0  ]     Rb   [t, N0]  rows l_p' = (L0^-1 k(X0, x_p) o ...)'          (one contiguous row per point)
//     [ Rb   Lt ]     Rinv [t, t]   inv(Lt), lower triangular, built a row per point
// An in-place append (solve.hip, gp_append_inplace_kernel) writes element (N, j) of EVERY column j of the packed operator: one 4-byte
// store per 128-byte line, 4096 x ~480 dirty lines per step at the unicycle shape, which drain while the next pass streams and cost it
// 0.14 of 0.51 ms (DESIGN_NOTES, round 5).  Here an append writes one contiguous row, the streaming pass runs over the static prefix
// (posterior_step_kernel<.., XC = 1>, all solved columns to Wfull) and this kernel continues it over the tail:
//     w_t = inv(Lt) (Phi_t - Rb W0)          the tail rows of  L^-1 [Phi(xq), phi(x_new)]
//     M_k += Vw_t' w_t,  B_k -= w_t' w_t     (control_affine_model.py:1051-1059 over all N0 + t points)
//     l = [W0[:, C]; w_t[:, C]],  d = sqrt(kappa - l'l):  the new point's row -- Rb[t] = l0', Rinv[t] = (-(l_t' inv(Lt)) / d, 1 / d)
// A failed pivot enters a neutral point (zero row, unit pivot), as bcbf_gp_append does.  The next window refit rebuilds L0 from the raw
// rows (ops.ReservedGP.drop_oldest_block) and the tail starts empty: nothing is ever committed to the column layout.
#include <hip/hip_runtime.h>

#include "bcbf.h"
#include "bcbf_common.h"

namespace bcbf {

template <typename T>
int launch_posterior_query_column_reserved(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2,
                                           const T* Bm, const T* M0, const T* xq, const T* x_new, const T* uh_new, T* Mk, T* Bk,
                                           T* lvec, T* lsum, int Bt, int N, int Ncap, int n, int m, void* stream,
                                           T* Wfull, int Lcap, int kind);   // posterior_step.hip

constexpr int TT = 256;                 // threads per instance
constexpr int TMAXT = 64;               // tail rows held in LDS

template <typename T> __device__ inline T tail_exp(T x);
template <> __device__ inline float tail_exp<float>(float x) { return expf(x); }        // (as the streaming pass: the tail rows are rows of the same solve)
template <> __device__ inline double tail_exp<double>(double x) { return exp(x); }

// global -> LDS, count elements, 8 independent loads in flight per thread (a plain copy loop waits out one round trip per element)
template <typename T>
__device__ inline void tail_stage(T* dst, const T* __restrict__ src, int count, int tid) {
    constexpr int D = 8;
    for (int i0 = 0; i0 < count; i0 += D * TT) {
        T v[D];
#pragma unroll
        for (int u = 0; u < D; ++u) { const int i = i0 + u * TT + tid; v[u] = i < count ? src[i] : T{}; }
#pragma unroll
        for (int u = 0; u < D; ++u) { const int i = i0 + u * TT + tid; if (i < count) dst[i] = v[u]; }
    }
}

template <typename T, int CT>
__global__ void __launch_bounds__(TT)
gp_tail_step_kernel(const T* __restrict__ W0, T* __restrict__ Rb, T* __restrict__ Rinv, T* __restrict__ X, T* __restrict__ UHB,
                    T* __restrict__ Vw, const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ Bm,
                    const T* __restrict__ M0, const T* __restrict__ xq, const T* __restrict__ x_new,
                    const T* __restrict__ uh_new, const T* __restrict__ xdot_new, const T* __restrict__ jitter_new,
                    const T* __restrict__ lsum, T* __restrict__ Mk, T* __restrict__ Bk, int* __restrict__ info,
                    T* __restrict__ rawUH, T* __restrict__ rawY, T* __restrict__ rawJ, int N0, int Np0, int t, int tcap, int ldR,
                    int Ncap, int n, int do_append, int kind) {
    // Everything small is staged in LDS with coalesced loads up front (the first form read Rinv / Vw / X rows from global memory
    // inside its short loops: a chain of exposed load latencies, 158 us per launch at 4096 x 472 + <= 40 where the bytes take 30)
    constexpr int CTM = BCBF_MAX_CTRL_DIM + 2, NSM = 4, C = CT - 1;        // NSM: the entry point takes n <= 4 (gp_tail_step below)
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    T* Wl = reinterpret_cast<T*>(tail_smem);                    // [Np0][CT]  the prefix's solved columns
    T* Ri = Wl + (size_t)Np0 * (C + 1);                         // [tcap][tcap] inv(Lt)
    __shared__ T rhs[TMAXT][CTM];
    __shared__ T wt[TMAXT][CTM];
    __shared__ T xt[TMAXT][NSM], vt[TMAXT][NSM], ut[TMAXT][CTM];   // tail rows of X, Vw, UH B
    __shared__ T part[TT / 64][TMAXT][CTM];
    __shared__ double sums[1 + NSM];                            // l'l, Vw'l over base + tail
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* W0b = W0 + (size_t)b * Np0 * CT;
    T* Rbb = Rb + (size_t)b * tcap * ldR;
    T* Rib = Rinv + (size_t)b * tcap * tcap;
    T* Xb = X + (size_t)b * Ncap * n;
    T* UHBb = UHB + (size_t)b * Ncap * C;
    T* Vwb = Vw + (size_t)b * Ncap * n;
    // every per-instance scalar the later phases need: ONE load per thread, all in flight with the copies below (read where they are
    // used, each was its own exposed round trip -- ~25 of them in a row, 45 us per workgroup at ~2 us a trip under load)
    __shared__ T s_ell[NSM], s_xq[NSM], s_xn[NSM], s_xd[NSM], s_uh[CTM], s_ls[1 + NSM], s_Bm[(CTM - 1) * (CTM - 1)], s_M0[(CTM - 1) * NSM], s_sc[2];
    {
        const T* src = nullptr;
        T* dst = nullptr;
        if (tid < n) { src = ell + (size_t)b * n + tid; dst = s_ell + tid; }
        else if (tid >= 8 && tid < 8 + n) { src = xq + (size_t)b * n + tid - 8; dst = s_xq + tid - 8; }
        else if (tid >= 16 && tid < 16 + n) { src = x_new + (size_t)b * n + tid - 16; dst = s_xn + tid - 16; }
        else if (tid >= 24 && tid < 24 + n) { if (xdot_new) { src = xdot_new + (size_t)b * n + tid - 24; dst = s_xd + tid - 24; } }
        else if (tid >= 32 && tid < 32 + C) { src = uh_new + (size_t)b * C + tid - 32; dst = s_uh + tid - 32; }
        else if (tid >= 40 && tid < 40 + 1 + n) { src = lsum + (size_t)b * (1 + n) + tid - 40; dst = s_ls + tid - 40; }
        else if (tid == 48) { src = s2p + b; dst = s_sc; }
        else if (tid == 49) { if (jitter_new) { src = jitter_new + b; dst = s_sc + 1; } }
        else if (tid >= 64 && tid < 64 + C * C) { src = Bm + (size_t)b * C * C + tid - 64; dst = s_Bm + tid - 64; }
        else if (tid >= 128 && tid < 128 + C * n) { src = M0 + (size_t)b * C * n + tid - 128; dst = s_M0 + tid - 128; }
        const T val = src ? *src : T(0);
        // (t n, t C <= 256: one element per thread; issued before the two larger copies so that all of it is in flight together)
        const bool hx = tid < t * n, hu = tid < t * C;
        const T xv = hx ? Xb[(size_t)N0 * n + tid] : T(0), vv = hx ? Vwb[(size_t)N0 * n + tid] : T(0);
        const T uv = hu ? UHBb[(size_t)N0 * C + tid] : T(0);
        tail_stage(reinterpret_cast<typename Vec<T>::type*>(Wl), reinterpret_cast<const typename Vec<T>::type*>(W0b), Np0 * CT / Vec<T>::V, tid);
        tail_stage(Ri, Rib, t * tcap, tid);
        if (hx) { xt[tid / n][tid % n] = xv; vt[tid / n][tid % n] = vv; }
        if (hu) ut[tid / C][tid % C] = uv;
        if (dst) *dst = val;
        if (tid == 49 && !jitter_new) s_sc[1] = T(0);
    }
    // (the posterior entries this kernel corrects: read now, used after the tail's solve)
    const T bk_in = tid < C * C ? Bk[(size_t)b * C * C + tid] : T(0);
    const T mk_in = (tid >= 64 && tid < 64 + n * C) ? Mk[(size_t)b * n * C + tid - 64] : T(0);
    // (b) tail residuals, first half: the partial sums of Rb[p] W0 -- wave w takes the columns j = 64 w + lane + 256 k of EVERY row
    // (all four waves stream every row together: t coalesced row segments in flight per wave instead of t / 4 whole rows in turn)
    __syncthreads();
    if (t > 0) {
        // lane = (tail row p, column group h), wave = a quarter of the columns: every lane owns whole dot products over its share of
        // the columns (no cross-lane sums: the first forms spent most of their instructions in butterfly reductions), W comes
        // from LDS as 16-byte reads shared by the lanes of a group, the row segments as 16-byte loads.  G = 4 / 2 / 1 groups for
        // t <= 16 / 32 / 64 rows: the instruction count follows the tail's size (a lane-per-row form ran 480 x CT multiply-adds per
        // wave whatever t was -- half the instructions of the whole streaming pass).  Columns N0 .. Np0 - 1: W is zero there.
        constexpr int V = Vec<T>::V;
        using VT = typename Vec<T>::type;
        const int G = t <= 16 ? 4 : t <= 32 ? 2 : 1, R = 64 / G;
        const int p = lane & (R - 1), h = lane / R;
        const int nv = Np0 / V / (TT / 64), vb = wave * nv;    // this wave's column vectors [vb, vb + nv)
        // (lanes beyond the last row read row t - 1 with the rest -- no predicate, no extra line -- and their sums are never read;
        //  U loads are issued before the first multiply-add: a load per loop trip was a round trip per trip)
        constexpr int U = 4;                                    // (8 in flight: slower, 44 -> 154 us over t against 40 -> 111)
        const T* row = Rbb + (size_t)min(p, t - 1) * ldR;
        T acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = T(0);
        for (int k0 = h; k0 < nv; k0 += G * U) {
            VT r[U];
            int j0[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * G;
                j0[u] = (vb + min(k, nv - 1)) * V;
                r[u] = *reinterpret_cast<const VT*>(row + j0[u]);
                if (k >= nv) r[u] = VT{};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                T wv[V * CT];
#pragma unroll
                for (int e = 0; e < V * CT; ++e) wv[e] = Wl[j0[u] * CT + e];
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] += reinterpret_cast<const T*>(&r[u])[v] * wv[v * CT + c];
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) part[wave][lane][c] = acc[c];
    }
    __syncthreads();
    // ... second half: Phi_p - the sums.  Phi_p: k(x_p, xq) (UH B)_p for the query's C columns, k(x_p, x_new) (UH B)_p . uh_new for
    // the append's column; a thread per tail row
    if (tid < t) {
        const int p = tid;
        T e1 = T(0), e2 = T(0), ud = T(0);
        for (int d = 0; d < n; ++d) {
            const T xv = xt[p][d], ie = T(1) / s_ell[d];
            const T z1 = (xv - s_xq[d]) * ie, z2 = (xv - s_xn[d]) * ie;
            e1 += z1 * z1;
            e2 += z2 * z2;
        }
        const T s2 = s_sc[0];
        T k1, k2, dsh_;
        if (kind != 0) {                                   // opt-in data kernels (bcbf_common.h), as the streaming pass in front
            kernel_shape(kind, e1, [](T q_) { return tail_exp<T>(q_); }, k1, dsh_);
            kernel_shape(kind, e2, [](T q_) { return tail_exp<T>(q_); }, k2, dsh_);
            k1 *= s2;
            k2 *= s2;
        } else { k1 = s2 * tail_exp<T>(T(-0.5) * e1); k2 = s2 * tail_exp<T>(T(-0.5) * e2); }
        const int G = t <= 16 ? 4 : t <= 32 ? 2 : 1, R = 64 / G;
        for (int c = 0; c < CT; ++c) {
            T sub = T(0);
#pragma unroll
            for (int w = 0; w < TT / 64; ++w)
                for (int h = 0; h < G; ++h) sub += part[w][h * R + p][c];
            if (c < C) {
                rhs[p][c] = k1 * ut[p][c] - sub;
                ud += ut[p][c] * s_uh[c];
            } else {
                rhs[p][C] = k2 * ud - sub;
            }
        }
    }
    __syncthreads();
    // (c) w_t = inv(Lt) rhs
    for (int e = tid; e < t * CT; e += TT) {
        const int p = e / CT, c = e - p * CT;
        double acc = 0.0;
        for (int q = 0; q <= p; ++q) acc += (double)Ri[p * tcap + q] * (double)rhs[q][c];
        wt[p][c] = (T)acc;
    }
    __syncthreads();
    // (d) the posterior at xq over base + tail, and the append's sums
    if (tid < C * C) {
        const int a = tid / C, c = tid - a * C;
        double g = 0.0;
        for (int p = 0; p < t; ++p) g += (double)wt[p][a] * (double)wt[p][c];
        Bk[(size_t)b * C * C + tid] = (T)((double)bk_in - g);
    } else if (tid >= 64 && tid < 64 + n * C) {
        const int e = tid - 64, d = e / C, c = e - d * C;
        double g = 0.0;
        for (int p = 0; p < t; ++p) g += (double)vt[p][d] * (double)wt[p][c];
        Mk[(size_t)b * n * C + e] = (T)((double)mk_in + g);
    } else if (tid >= 128 && tid < 128 + 1 + n) {
        const int e = tid - 128;
        double g = (double)s_ls[e];
        if (e == 0) for (int p = 0; p < t; ++p) g += (double)wt[p][C] * (double)wt[p][C];
        else for (int p = 0; p < t; ++p) g += (double)vt[p][e - 1] * (double)wt[p][C];
        sums[e] = g;
    }
    __syncthreads();
    if (!do_append) return;
    // (e) the new point's row
    T uh[BCBF_MAX_CTRL_DIM + 1];
#pragma unroll
    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) uh[c] = c < C ? s_uh[c] : T(0);
    T q = T(0);
#pragma unroll
    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) {
        T sacc = T(0);
        if (c < C) for (int a = 0; a < C; ++a) sacc += s_Bm[c * C + a] * uh[a];
        q += uh[c] * sacc;
    }
    const T kap = s_sc[0] * q + s_sc[1];
    const double d2 = (double)kap - sums[0];
    const bool ok = d2 > 0.0;
    const T dv = ok ? (T)sqrt(d2) : T(1);
    const int N = N0 + t;
    if (tid == 0) info[b] = ok ? 0 : N + 1;
    T* nrow = Rbb + (size_t)t * ldR;
    for (int j = tid; j < N0; j += TT) __builtin_nontemporal_store(ok ? Wl[j * CT + C] : T(0), &nrow[j]);     // (read next by another step's kernel, from memory)
    if (tid <= t) {
        T val;
        if (tid == t) val = ok ? T(1) / dv : T(1);
        else {
            double a2 = 0.0;
            for (int p = tid; p < t; ++p) a2 += (double)wt[p][C] * (double)Ri[p * tcap + tid];
            val = ok ? (T)(-a2 / (double)dv) : T(0);
        }
        Rib[(size_t)t * tcap + tid] = val;
    }
    if (tid >= 64 && tid < 64 + n) {
        const int c = tid - 64;
        T yv = s_xd[c];
        for (int a = 0; a < C; ++a) yv -= uh[a] * s_M0[a * n + c];
        Vwb[(size_t)N * n + c] = ok ? (T)(((double)yv - sums[1 + c]) / (double)dv) : T(0);
        Xb[(size_t)N * n + c] = s_xn[c];
        if (rawY != nullptr) rawY[((size_t)b * Ncap + N) * n + c] = ok ? s_xd[c] : T(0);
    }
    if (tid >= 128 && tid < 128 + C) {
        const int c = tid - 128;
        T sacc = T(0);
        for (int a = 0; a < C; ++a) sacc += uh[a] * s_Bm[a * C + c];
        UHBb[(size_t)N * C + c] = ok ? sacc : T(0);
        if (rawUH != nullptr) rawUH[((size_t)b * Ncap + N) * C + c] = ok ? uh[c] : T(0);
    }
    if (rawJ != nullptr && tid == 192) rawJ[(size_t)b * Ncap + N] = ok ? s_sc[1] : T(1);
}

template <typename T>
static int gp_tail_step(const T* Lop_r, T* Vw_r, T* X_r, T* UHB_r, const T* ell, const T* s2, const T* Bm, const T* M0,
                        const T* xq, const T* x_new, const T* uh_new, const T* xdot_new, const T* jitter_new, T* Rb, T* Rinv,
                        int* info, T* Wwork, T* swork, T* Mk, T* Bk, T* rawUH, T* rawY, T* rawJ, int Bt, int N0, int t,
                        int tcap, int Ncap, int Lcap, int n, int m, int do_append, void* stream, int kind = 0) {
    if (Bt <= 0) return BCBF_OK;
    if (kind < 0 || kind >= BCBF_KINDS) return BCBF_EINVAL;
    if (!Lop_r || !Vw_r || !X_r || !UHB_r || !ell || !s2 || !Bm || !M0 || !xq || !x_new || !uh_new || !Rb || !Rinv || !Wwork ||
        !swork || !Mk || !Bk)
        return BCBF_EINVAL;
    if (do_append && (!xdot_new || !info)) return BCBF_EINVAL;
    if ((rawUH || rawY || rawJ) && !(rawUH && rawY && rawJ)) return BCBF_EINVAL;
    if (N0 < 1 || t < 0 || tcap < 1 || tcap > TMAXT || t + (do_append ? 1 : 0) > tcap || N0 + t + (do_append ? 1 : 0) > Ncap) return BCBF_EINVAL;
    if (n < 1 || n > 4 || m < 1 || m > 3 || Lcap < N0) return BCBF_EINVAL;
    const int Np0 = round_up(N0, NB), CT = m + 2;
    const size_t smem = ((size_t)Np0 * CT + (size_t)tcap * tcap) * sizeof(T);
    // the window's W0 [Np0 x CT] and the tail's inverse [tcap x tcap] live in LDS beside ~14 KB (fp64) of static arrays: gfx950 gives a
    // workgroup 160 KB.  What bounds the window in practice is the streaming pass in front (posterior_step.hip: its solved columns
    // stay in LDS, Np0 (m + 2) sizeof(T) <= 100 KB, and its workgroup covers Np0 <= 2048): N0 <= 2048 in both precisions (bcbf.h).
    if (smem > 120 * 1024) return BCBF_EINVAL;
    const int rc = launch_posterior_query_column_reserved<T>(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, xq, x_new, uh_new, Mk, Bk,
                                                             (T*)nullptr, swork, Bt, N0, Ncap, n, m, stream, Wwork, Lcap, kind);
    if (rc != BCBF_OK) return rc;
#define BCBF_TAIL_LAUNCH(CTV)                                                                                                      \
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)gp_tail_step_kernel<T, CTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((gp_tail_step_kernel<T, CTV>), dim3(Bt), dim3(TT), smem, (hipStream_t)stream, Wwork, Rb, Rinv, X_r, UHB_r,     \
                       Vw_r, ell, s2, Bm, M0, xq, x_new, uh_new, xdot_new, jitter_new, swork, Mk, Bk, info, rawUH, rawY, rawJ, N0,   \
                       Np0, t, tcap, round_up(Ncap, NB), Ncap, n, do_append, kind)
    switch (m) {
        case 1: BCBF_TAIL_LAUNCH(3); break;
        case 2: BCBF_TAIL_LAUNCH(4); break;
        default: BCBF_TAIL_LAUNCH(5); break;
    }
#undef BCBF_TAIL_LAUNCH
    return check_launch("gp_tail_step");
}
// GROWTH with a tail (round 6): without a window the model keeps growing, so the tail cannot wait for a refit -- every
// 32 appends its rows are COMMITTED to the reserved column layout as one whole block row: element (N0 + r, j) of column j for r =
// 0..31 is 32 consecutive elements of that column, a full 128-byte line in fp32 (two in fp64), where the in-place append wrote one
// element per line and step (the dirty partial lines were what the next streaming pass waited for).  The block's own diagonal tile is
// the tail's triangular block: its inverse Rinv goes to the packed-triangle and the full-tile copies.  N0 a multiple of 32, t = 32.
// grid (N0 / 32 + 1, Bt): workgroup (Jc, b) transposes the 32 x 32 tile Rb[0..31][32 Jc ..] through LDS; the last one writes the
// diagonal block.
template <typename T>
__global__ void __launch_bounds__(256)
gp_tail_commit_kernel(T* __restrict__ Lop, const T* __restrict__ Rb, const T* __restrict__ Rinv, int N0, int tcap, int ldR, int Nl) {
    constexpr int V = Vec<T>::V;
    __shared__ T tile[NB][NB + 1];
    const int b = blockIdx.y, Jc = blockIdx.x, Jn = N0 / NB, tid = threadIdx.x;
    T* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Nl);
    const int c = tid & 31, r0 = tid >> 5;                       // 8 rows per pass
    if (Jc < Jn) {
        const T* Rbb = Rb + (size_t)b * tcap * ldR;
#pragma unroll
        for (int r = r0; r < NB; r += 8) tile[r][c] = Rbb[(size_t)r * ldR + Jc * NB + c];        // row r: 32 consecutive columns
        __syncthreads();
#pragma unroll
        for (int cc = r0; cc < NB; cc += 8)                        // column cc: rows N0 .. N0 + 31, consecutive
            lop[lop_base<V>(Jc * NB + cc, Nl) + N0 + c] = tile[c][cc];
        return;
    }
    const T* Rib = Rinv + (size_t)b * tcap * tcap;
#pragma unroll
    for (int r = r0; r < NB; r += 8) tile[r][c] = c <= r ? Rib[(size_t)r * tcap + c] : T(0);     // inv(Lt)[r][c]
    __syncthreads();
#pragma unroll
    for (int cc = r0; cc < NB; cc += 8) {
        lop[lop_dfull(Jn, c, cc, Nl)] = tile[c][cc];              // full tile, column-major, zeros above the diagonal
        if (c >= cc) lop[lop_dinv(Jn, c, cc, Nl)] = tile[c][cc];  // packed lower triangle
    }
}

template <typename T>
static int gp_tail_commit(T* Lop_r, const T* Rb, const T* Rinv, int Bt, int N0, int t, int tcap, int Ncap, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lop_r || !Rb || !Rinv) return BCBF_EINVAL;
    if (N0 < NB || N0 % NB != 0 || t != NB || tcap < NB || tcap > TMAXT || N0 + NB > Ncap) return BCBF_EINVAL;
    const int Nl = round_up(Ncap, NB);
    hipLaunchKernelGGL((gp_tail_commit_kernel<T>), dim3(N0 / NB + 1, Bt), dim3(256), 0, (hipStream_t)stream, Lop_r, Rb, Rinv, N0, tcap,
                       Nl, Nl);
    return check_launch("gp_tail_commit");
}
}  // namespace bcbf

extern "C" {
#define BCBF_TAIL_ENTRY(SUF, T)                                                                                              \
    int bcbf_gp_tail_step_##SUF(const T* Lop_r, T* Vw_r, T* X_r, T* UHB_r, const T* ell, const T* s2, const T* Bm, const T* M0, \
                                const T* xq, const T* x_new, const T* uh_new, const T* xdot_new, const T* jitter_new, T* Rb,  \
                                T* Rinv, int* info, T* Wwork, T* swork, T* Mk, T* Bk, T* rawUH, T* rawY, T* rawJ,              \
                                int Bt, int N0, int t, int tcap, int Ncap, int Lcap, int n, int m, int do_append,             \
                                void* stream) {                                                                              \
        return bcbf::gp_tail_step<T>(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, xq, x_new, uh_new, xdot_new, jitter_new, Rb,    \
                                     Rinv, info, Wwork, swork, Mk, Bk, rawUH, rawY, rawJ, Bt, N0, t, tcap, Ncap, Lcap, n, m,   \
                                     do_append, stream);                                                                     \
    }                                                                                                                        \
    /* ... with the opt-in data kernels: kernel_kind 0 RBF, 1 Matern-5/2, 2 RBF x Matern-5/2 (the window's factor from the      \
       matching bcbf_refit_*; the tail rows and the new point's row are evaluated with the same kernel) */                    \
    int bcbf_gp_tail_step_kind_##SUF(const T* Lop_r, T* Vw_r, T* X_r, T* UHB_r, const T* ell, const T* s2, const T* Bm,        \
                                     const T* M0, const T* xq, const T* x_new, const T* uh_new, const T* xdot_new,           \
                                     const T* jitter_new, T* Rb, T* Rinv, int* info, T* Wwork, T* swork, T* Mk, T* Bk,        \
                                     T* rawUH, T* rawY, T* rawJ, int Bt, int N0, int t, int tcap, int Ncap, int Lcap, int n,  \
                                     int m, int do_append, int kernel_kind, void* stream) {                                  \
        return bcbf::gp_tail_step<T>(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, xq, x_new, uh_new, xdot_new, jitter_new, Rb,    \
                                     Rinv, info, Wwork, swork, Mk, Bk, rawUH, rawY, rawJ, Bt, N0, t, tcap, Ncap, Lcap, n, m,   \
                                     do_append, stream, kernel_kind);                                                        \
    }
BCBF_TAIL_ENTRY(f32, float)
BCBF_TAIL_ENTRY(f64, double)
#undef BCBF_TAIL_ENTRY
int bcbf_gp_tail_commit_f32(float* Lop_r, const float* Rb, const float* Rinv, int Bt, int N0, int t, int tcap, int Ncap, void* stream) {
    return bcbf::gp_tail_commit<float>(Lop_r, Rb, Rinv, Bt, N0, t, tcap, Ncap, stream);
}
int bcbf_gp_tail_commit_f64(double* Lop_r, const double* Rb, const double* Rinv, int Bt, int N0, int t, int tcap, int Ncap, void* stream) {
    return bcbf::gp_tail_commit<double>(Lop_r, Rb, Rinv, Bt, N0, t, tcap, Ncap, stream);
}
}
