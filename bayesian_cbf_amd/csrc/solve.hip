// K3: triangular solves against the packed operator (refit path, once per refit).
//   forward :  Vw    = L^-1 Y,      Y = Xdot - UH M0
//   backward:  alpha = L^-T Vw  (= K_b^-1 Y)
// Thread-per-row; the 32x32 diagonal blocks are stored inverted, so each block step is a
// mat-vec, and the rows below / columns above are reached with loads that are contiguous
// across the workgroup (the operator is column-major).
#include "bcbf_common.h"

namespace bcbf {

constexpr int ST = 256;
constexpr int SMAXR = 8;                       // rows per thread (N <= 2048)
constexpr int SC = BCBF_MAX_STATE_DIM;         // max RHS columns

template <typename T>
__device__ inline void block_sum(T* vals, int count, T* scratch /* [4][SC] */) {
    // sum `count` (<= SC) values over the 256-thread workgroup; result valid in every thread
    for (int c = 0; c < count; ++c) vals[c] = wave_sum(vals[c]);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < count; ++c) scratch[w * SC + c] = vals[c];
    __syncthreads();
    for (int c = 0; c < count; ++c)
        vals[c] = scratch[c] + scratch[SC + c] + scratch[2 * SC + c] + scratch[3 * SC + c];
}

// IDENT (bcbf_potri): the right-hand sides are 8 columns of the identity -- workgroup (chunk, b) solves columns
// 8 chunk .. 8 chunk + 7 of K_b^-1 and writes them straight into the dense inverse Kinv[Bt,N,N] (`alpha`, row stride N).
template <typename T, bool IDENT>
__global__ void __launch_bounds__(ST)
potrs_kernel(const T* __restrict__ Lop, const T* __restrict__ Xdot, const T* __restrict__ UH,
             const T* __restrict__ M0, T* __restrict__ Vw, T* __restrict__ alpha, int N, int Np, int n, int C) {
    constexpr int V = Vec<T>::V;
    __shared__ T rbuf[NB][SC];
    __shared__ T wbuf[NB][SC];
    __shared__ T scratch[4 * SC];
    const int b = IDENT ? blockIdx.y : blockIdx.x, tid = threadIdx.x;
    const int j0 = IDENT ? blockIdx.x * SC : 0;          // first identity column of this workgroup
    const int ostride = IDENT ? N : n;                   // row stride of the alpha output
    const T* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    const int rpt = (Np + ST - 1) / ST;
    const int nblk = Np / NB;
    if (IDENT) n = (N - j0 < SC) ? N - j0 : SC;

    // ---- Y rows owned by this thread
    T y[SMAXR][SC];
#pragma unroll
    for (int r = 0; r < SMAXR; ++r) {
        const int i = tid + r * ST;
#pragma unroll
        for (int c = 0; c < SC; ++c) {
            T v = T(0);
            if (IDENT) v = (r < rpt && i == j0 + c && c < n) ? T(1) : T(0);
            else if (r < rpt && i < N && c < n) {
                v = Xdot[((size_t)b * N + i) * n + c];
                for (int a = 0; a < C; ++a) v -= UH[((size_t)b * N + i) * C + a] * M0[((size_t)b * C + a) * n + c];
            }
            y[r][c] = v;
        }
    }
    // ---- forward substitution
    for (int J = 0; J < nblk; ++J) {
        const int col0 = J * NB;
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            if (r < rpt && i >= col0 && i < col0 + NB) {
#pragma unroll
                for (int c = 0; c < SC; ++c) rbuf[i - col0][c] = y[r][c];
            }
        }
        __syncthreads();
        if (tid < NB) {
            T w[SC];
#pragma unroll
            for (int c = 0; c < SC; ++c) w[c] = T(0);
            for (int jj = 0; jj <= tid; ++jj) {
                const T val = lop[lop_dinv(J, tid, jj, Np)];
#pragma unroll
                for (int c = 0; c < SC; ++c) w[c] += val * rbuf[jj][c];
            }
#pragma unroll
            for (int c = 0; c < SC; ++c) wbuf[tid][c] = w[c];
            if (!IDENT && col0 + tid < N)
                for (int c = 0; c < n; ++c) Vw[((size_t)b * N + col0 + tid) * n + c] = w[c];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            if (r < rpt && i >= col0 + NB && i < Np) {
                for (int jj = 0; jj < NB; ++jj) {
                    const T val = lop[lop_base<V>(col0 + jj, Np) + i];
#pragma unroll
                    for (int c = 0; c < SC; ++c) y[r][c] -= val * wbuf[jj][c];
                }
            }
            if (r < rpt && i >= col0 && i < col0 + NB) {   // keep Vw rows in registers for the backward pass
#pragma unroll
                for (int c = 0; c < SC; ++c) y[r][c] = wbuf[i - col0][c];
            }
        }
        __syncthreads();
    }
    if (alpha == nullptr) return;
    if (IDENT && C == 1) {
        // forward only (bcbf_trtri): the solved rows are 8 columns of L^-1
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            if (r < rpt && i < N)
                for (int c = 0; c < n; ++c) alpha[((size_t)b * N + i) * ostride + j0 + c] = y[r][c];
        }
        return;
    }

    // ---- backward substitution: y holds Vw rows; solved rows are overwritten with alpha
    for (int J = nblk - 1; J >= 0; --J) {
        const int col0 = J * NB;
        // t_J[jj] = Vw[jj] - sum_{i in blocks > J} L[i][col0+jj] alpha[i]
        for (int jj = 0; jj < NB; ++jj) {
            T s[SC];
#pragma unroll
            for (int c = 0; c < SC; ++c) s[c] = T(0);
            const int base = lop_base<V>(col0 + jj, Np);
#pragma unroll
            for (int r = 0; r < SMAXR; ++r) {
                const int i = tid + r * ST;
                if (r < rpt && i >= col0 + NB && i < Np) {
                    const T val = lop[base + i];
#pragma unroll
                    for (int c = 0; c < SC; ++c) s[c] += val * y[r][c];
                }
            }
            block_sum(s, n, scratch);
            if (tid == 0)
                for (int c = 0; c < SC; ++c) rbuf[jj][c] = c < n ? s[c] : T(0);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            if (r < rpt && i >= col0 && i < col0 + NB) {
#pragma unroll
                for (int c = 0; c < SC; ++c) rbuf[i - col0][c] = y[r][c] - rbuf[i - col0][c];
            }
        }
        __syncthreads();
        if (tid < NB) {   // alpha_J = inv(L_JJ)^T t_J : lane jj walks its own (contiguous) column
            T a[SC];
#pragma unroll
            for (int c = 0; c < SC; ++c) a[c] = T(0);
            const int base = lop_dinv_block(J, Np) + lop_dinv_col(tid);
            for (int ii = tid; ii < NB; ++ii) {
                const T val = lop[base + ii];
#pragma unroll
                for (int c = 0; c < SC; ++c) a[c] += val * rbuf[ii][c];
            }
#pragma unroll
            for (int c = 0; c < SC; ++c) wbuf[tid][c] = a[c];
            if (col0 + tid < N)
                for (int c = 0; c < n; ++c) alpha[((size_t)b * N + col0 + tid) * ostride + j0 + c] = a[c];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            if (r < rpt && i >= col0 && i < col0 + NB) {
#pragma unroll
                for (int c = 0; c < SC; ++c) y[r][c] = wbuf[i - col0][c];
            }
        }
        __syncthreads();
    }
}

template <typename T>
int launch_forward_stream(const T* Lop, const T* Xdot, const T* UH, const T* M0, T* Vw, int Bt, int N, int n, int cu,
                          void* stream);                       // posterior_step.hip

template <typename T>
static int launch_potrs(const T* Lop, const T* Xdot, const T* UH, const T* M0, T* Vw, T* alpha,
                        int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !Xdot || !UH || !M0 || !Vw) return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    // Only the whitened targets are wanted (every refit of the control path): the forward solve runs on the posterior
    // kernel's streaming structure (mirrored row pairs, software-pipelined nt loads) -- the thread-per-row kernel below
    // reached 26 % of the HBM peak, the stream holds 80 %.  alpha = K_b^-1 Y (fit path) keeps the kernel below.
    if (alpha == nullptr) {
        const int rc = launch_forward_stream<T>(Lop, Xdot, UH, M0, Vw, Bt, N, n, m + 1, stream);
        if (rc <= 0) return rc;
    }
    const int Np = round_up(N, NB);
    if (Np > ST * SMAXR) return BCBF_EINVAL;
    hipLaunchKernelGGL((potrs_kernel<T, false>), dim3(Bt), dim3(ST), 0, (hipStream_t)stream, Lop, Xdot, UH, M0, Vw,
                       alpha, N, Np, n, m + 1);
    return check_launch("potrs");
}

int launch_trtri_mfma_f32(const float* Lop, float* Linv, int Bt, int N, void* stream);       // trtri.hip
int launch_trtri_mfma_f64(const double* Lop, double* Linv, int Bt, int N, void* stream);

template <typename T>
static int launch_potri(const T* Lop, T* Kinv, int Bt, int N, void* stream, int forward_only = 0) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !Kinv || N < 1) return BCBF_EINVAL;
    const int Np = round_up(N, NB);
    if (Np > ST * SMAXR) return BCBF_EINVAL;
    hipLaunchKernelGGL((potrs_kernel<T, true>), dim3((N + SC - 1) / SC, Bt), dim3(ST), 0, (hipStream_t)stream, Lop,
                       nullptr, nullptr, nullptr, nullptr, Kinv, N, Np, SC, forward_only);
    return check_launch("potri");
}


// --------------------------------------------------------------------------------------------
// K11: bordered Cholesky -- append one training point to the packed operator.
//   l = L^-1 knew,  d = sqrt(kappa - l'l);  new last row of L = [l', d].
// The new row lands in diagonal block J* = N/32: its inverse gains the row
// [-(1/d) l_J*' inv(L_J*J*), 1/d]; every other stored element is copied (re-laid out when the
// padded size grows by a block).
// FUSED (bcbf_gp_append): the whole online update of one observation (x, uh, xdot) per instance -- the new kernel
// column k(X, x) o (UH B uh) and its diagonal are formed here instead of being read, and the whitened-target row
// Vw[N] = (y - l' Vw) / d and the new rows of X / UH B are appended to the (re-packed) per-refit arrays.
template <typename T>
struct AppendFused {
    const T *Vw_in, *X_in, *UHB_in, *ell, *s2, *Bm, *M0, *x_new, *uh_new, *xdot_new, *jitter_new;
    T *Vw_out, *X_out, *UHB_out;
    int n, C;
    int kind;            // data kernel of the new column: 0 = RBF, 1 = Matern-5/2 (opt-in)
    const T* Wgiven;     // optional: W = L^-1 Phi(x_new) [Bt, NpI, C] from the streaming posterior kernel (bcbf_gp_append_stream):
                         // l = W uh_new, the forward solve below is skipped
};

// The packed operator of NpI (padded) points re-laid out for NpO >= NpI points: every stored element keeps its (row,
// column); rows / columns of the new padding are those of the identity.  (The layout's column lengths and block offsets
// depend on the padded size: bcbf_common.h.)
template <typename T>
__device__ inline void relayout_operator(const T* __restrict__ lin, T* __restrict__ lout, int NpI, int NpO, int tid) {
    constexpr int V = Vec<T>::V;
    // off-diagonal part: column j keeps its rows below its diagonal block; rows / columns of the new padding are zero
    // (16-byte vectors -- every column starts on a 128-byte line and NpI, NpO, `first` are multiples of 32 --, four columns per trip so
    //  that a thread has four loads in flight: one 4-byte element of one column per trip moved 4.2 GB in 1.3 ms at 4096 x 512)
    using VT = typename Vec<T>::type;
    for (int j0 = 0; j0 < NpO; j0 += 4) {
        VT v[4];
        int io[4], bo[4];
        bool ok[4];
        const int first = (j0 / NB + 1) * NB;                   // (j0 .. j0 + 3 lie in one block column: NB is a multiple of 4)
        const int nv = (NpO - first) / V;                        // vectors per column of this block column
        for (int k0 = 0; k0 < nv; k0 += ST) {
            const int k = k0 + tid;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u, i = first + k * V;
                ok[u] = k < nv;
                io[u] = i;
                bo[u] = lop_base<V>(j, NpO);
                v[u] = (ok[u] && j < NpI && i < NpI) ? *reinterpret_cast<const VT*>(lin + lop_base<V>(j, NpI) + i) : VT{};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u]) *reinterpret_cast<VT*>(lout + bo[u] + io[u]) = v[u];
        }
    }
    // inverted diagonal blocks: copied; a new padding block is the identity
    const int nbI = NpI / NB, nbO = NpO / NB;
    for (int e = tid; e < nbO * LOP_DB; e += ST) {
        const int Jb = e / LOP_DB, o = e - Jb * LOP_DB;
        T v;
        if (Jb < nbI) v = lin[lop_dinv_block(Jb, NpI) + o];
        else {                       // identity: o is a diagonal position iff o == lop_dinv_col(c) + c for some c
            v = T(0);
            for (int c = 0; c < NB; ++c) if (o == lop_dinv_col(c) + c) v = T(1);
        }
        lout[lop_dinv_block(Jb, NpO) + o] = v;
    }
    // ... and their full-tile copies (read by the shared-model kernel)
    for (int e = tid; e < nbO * NB * NB; e += ST) {
        const int Jb = e / (NB * NB), o = e - Jb * NB * NB;
        lout[lop_dfull_block(Jb, NpO) + o] = Jb < nbI ? lin[lop_dfull_block(Jb, NpI) + o] : ((o / NB == o % NB) ? T(1) : T(0));
    }
}

template <typename T, bool FUSED>
__global__ void __launch_bounds__(ST)
chol_append_kernel(const T* Lin, const T* __restrict__ knew, const T* __restrict__ kappa,
                   T* Lout, int* __restrict__ info, int N, int NpI, int NpO, AppendFused<T> f) {   // Lout may alias Lin
    constexpr int V = Vec<T>::V;
    __shared__ T rbuf[NB];
    __shared__ T wbuf[NB];
    __shared__ T lrow[ST * SMAXR];      // l, all rows
    __shared__ T scratch[4 * SC];
    const int b = blockIdx.x, tid = threadIdx.x;
    const T* lin = Lin + (size_t)b * lop_elems<V>(NpI);
    T* lout = Lout + (size_t)b * lop_elems<V>(NpO);
    const int rpt = (NpI + ST - 1) / ST;
    const int nblk = NpI / NB;

    // ---- forward solve l = L^-1 knew on the old operator
    T y[SMAXR];
    T kap = T(0);
    if (FUSED) {
        const int n = f.n, C = f.C;
        T xn[BCBF_MAX_STATE_DIM], ie[BCBF_MAX_STATE_DIM];
        const T s2v = f.s2[b];
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) {
            xn[d] = d < n ? f.x_new[(size_t)b * n + d] : T(0);
            ie[d] = d < n ? T(1) / f.ell[(size_t)b * n + d] : T(0);
        }
        T q = T(0);
#pragma unroll
        for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) {
            T sacc = T(0);
            if (c < C) for (int a = 0; a < C; ++a) sacc += f.Bm[((size_t)b * C + c) * C + a] * f.uh_new[(size_t)b * C + a];
            if (c < C) q += f.uh_new[(size_t)b * C + c] * sacc;
        }
        kap = s2v * q + (f.jitter_new ? f.jitter_new[b] : T(0));
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            T v = T(0);
            if (f.Wgiven != nullptr) {
                // l_i = sum_c W[i][c] uh_new[c]:  Phi(x) uh = k(X, x) o (UH B uh) is the new kernel column, so L^-1 of it is W uh
                if (r < rpt && i < NpI) {
#pragma unroll
                    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c)
                        if (c < C) v += f.Wgiven[((size_t)b * NpI + i) * C + c] * f.uh_new[(size_t)b * C + c];
                    lrow[i] = i < N ? v : T(0);
                }
            } else if (r < rpt && i < N) {
                T d2 = T(0), uu = T(0);
#pragma unroll
                for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d)
                    if (d < n) { const T z = (f.X_in[((size_t)b * N + i) * n + d] - xn[d]) * ie[d]; d2 += z * z; }
                // UHB_i . uh_new  (Bm symmetric: (UH_i Bm) uh = UH_i (Bm uh))
#pragma unroll
                for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c)
                    if (c < C) uu += f.UHB_in[((size_t)b * N + i) * C + c] * f.uh_new[(size_t)b * C + c];
                double shp, dshp_;
                kernel_shape(f.kind, (double)d2, [](double q) { return exp(q); }, shp, dshp_);
                v = s2v * (T)shp * uu;
            }
            y[r] = v;
        }
    } else {
        kap = kappa[b];
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            y[r] = (r < rpt && i < N) ? knew[(size_t)b * N + i] : T(0);
        }
    }
    const bool solved = FUSED && f.Wgiven != nullptr;
    if (solved) __syncthreads();
    for (int J = 0; J < (solved ? 0 : nblk); ++J) {
        const int col0 = J * NB;
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            if (r < rpt && i >= col0 && i < col0 + NB) rbuf[i - col0] = y[r];
        }
        __syncthreads();
        if (tid < NB) {
            T w = T(0);
            for (int jj = 0; jj <= tid; ++jj) w += lin[lop_dinv(J, tid, jj, NpI)] * rbuf[jj];
            wbuf[tid] = w;
            lrow[col0 + tid] = w;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SMAXR; ++r) {
            const int i = tid + r * ST;
            if (r < rpt && i >= col0 + NB && i < NpI) {
                T acc = y[r];
                for (int jj = 0; jj < NB; ++jj) acc -= lin[lop_base<V>(col0 + jj, NpI) + i] * wbuf[jj];
                y[r] = acc;
            }
        }
        __syncthreads();
    }
    T ss[SC];
#pragma unroll
    for (int c = 0; c < SC; ++c) ss[c] = T(0);
    for (int i = tid; i < N; i += ST) ss[0] += lrow[i] * lrow[i];
    block_sum(ss, 1, scratch);
    const T d2 = kap - ss[0];
    const bool ok = d2 > T(0);
    const T d = ok ? (T)sqrt((double)d2) : T(1);
    if (tid == 0) info[b] = ok ? 0 : N + 1;

    // ---- copy / re-layout the old operator
    if (lout != lin) relayout_operator<T>(lin, lout, NpI, NpO, tid);
    __threadfence_block();
    __syncthreads();
    // ---- the new row N.  A non-positive pivot (info = N+1) writes nothing: row N stays the identity padding row it
    // was (in place) or has just become (re-packed), so the operator still describes the old N points exactly and
    // the caller can retry with a larger jitter (make_psd's schedule, control_affine_model.py:905-919)
    const int Js = N / NB, col0 = Js * NB, rr = N - col0;
    if (ok) for (int j = tid; j < col0; j += ST) __builtin_nontemporal_store(lrow[j], &lout[lop_base<V>(j, NpO) + N]);   // (one element per line: see gp_append_inplace_kernel)
    if (ok && tid <= rr) {
        const int jj = tid;          // column inside the diagonal block
        T val;
        if (jj == rr) val = T(1) / d;
        else {
            T acc = T(0);
            for (int ii = jj; ii < rr; ++ii) acc += lrow[col0 + ii] * lout[lop_dinv(Js, ii, jj, NpO)];
            val = -acc / d;
        }
        lout[lop_dinv(Js, rr, jj, NpO)] = val;
        lout[lop_dfull(Js, rr, jj, NpO)] = val;
    }
    if (FUSED) {
        const int n = f.n, C = f.C;
        // Vw[N] = (y - l' Vw) / d,  y = xdot_new - M0' uh_new
        T vs[SC];
#pragma unroll
        for (int c = 0; c < SC; ++c) vs[c] = T(0);
        for (int i = tid; i < N; i += ST) {
            const T li = lrow[i];
#pragma unroll
            for (int c = 0; c < SC; ++c) if (c < n) vs[c] += li * f.Vw_in[((size_t)b * N + i) * n + c];
        }
        block_sum(vs, n, scratch);
        const size_t N1 = (size_t)N + 1;
        if (tid < n) {
            T yv = f.xdot_new[(size_t)b * n + tid];
            for (int a = 0; a < C; ++a) yv -= f.uh_new[(size_t)b * C + a] * f.M0[((size_t)b * C + a) * n + tid];
            f.Vw_out[((size_t)b * N1 + N) * n + tid] = ok ? (yv - vs[tid]) / d : T(0);   // !ok: a neutral row
            f.X_out[((size_t)b * N1 + N) * n + tid] = f.x_new[(size_t)b * n + tid];
        }
        if (tid < C) {
            T sacc = T(0);
            for (int a = 0; a < C; ++a) sacc += f.uh_new[(size_t)b * C + a] * f.Bm[((size_t)b * C + a) * C + tid];
            f.UHB_out[((size_t)b * N1 + N) * C + tid] = ok ? sacc : T(0);     // !ok: Phi row 0 -> the point has no effect
        }
        // re-pack the per-refit arrays (batch stride N -> N+1)
        for (int e = tid; e < N * n; e += ST) {
            f.Vw_out[(size_t)b * N1 * n + e] = f.Vw_in[(size_t)b * N * n + e];
            f.X_out[(size_t)b * N1 * n + e] = f.X_in[(size_t)b * N * n + e];
        }
        for (int e = tid; e < N * C; e += ST) f.UHB_out[(size_t)b * N1 * C + e] = f.UHB_in[(size_t)b * N * C + e];
    }
}

template <typename T>
static int launch_chol_append(const T* Lin, const T* knew, const T* kappa, T* Lout, int* info, int Bt, int N,
                              void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lin || !knew || !kappa || !Lout || !info || N < 1) return BCBF_EINVAL;
    const int NpI = round_up(N, NB), NpO = round_up(N + 1, NB);
    if (NpO > ST * SMAXR) return BCBF_EINVAL;
    if (Lin == Lout && NpI != NpO) return BCBF_EINVAL;
    hipLaunchKernelGGL((chol_append_kernel<T, false>), dim3(Bt), dim3(ST), 0, (hipStream_t)stream, Lin, knew, kappa,
                       Lout, info, N, NpI, NpO, AppendFused<T>{});
    return check_launch("chol_append");
}

template <typename T>
static int launch_gp_append(const T* Lin, const T* Vw_in, const T* X_in, const T* UHB_in, const T* ell, const T* s2,
                            const T* Bm, const T* M0, const T* x_new, const T* uh_new, const T* xdot_new,
                            const T* jitter_new, T* Lout, T* Vw_out, T* X_out, T* UHB_out, int* info, int Bt, int N,
                            int n, int m, void* stream, const T* Wgiven = nullptr, int kind = 0) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lin || !Vw_in || !X_in || !UHB_in || !ell || !s2 || !Bm || !M0 || !x_new || !uh_new || !xdot_new || !Lout ||
        !Vw_out || !X_out || !UHB_out || !info || N < 1)
        return BCBF_EINVAL;
    if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    if (Vw_out == Vw_in || X_out == X_in || UHB_out == UHB_in) return BCBF_EINVAL;     // batch stride changes
    const int NpI = round_up(N, NB), NpO = round_up(N + 1, NB);
    if (NpO > ST * SMAXR) return BCBF_EINVAL;
    if (Lin == Lout && NpI != NpO) return BCBF_EINVAL;
    AppendFused<T> f{Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new, Vw_out, X_out, UHB_out,
                     n, m + 1, kind, Wgiven};
    hipLaunchKernelGGL((chol_append_kernel<T, true>), dim3(Bt), dim3(ST), 0, (hipStream_t)stream, Lin, nullptr, nullptr,
                       Lout, info, N, NpI, NpO, f);
    return check_launch("gp_append");
}


// --------------------------------------------------------------------------------------------
// Capacity-reserving GP storage (online path, BASELINE configs[4]).  The packed layout depends on the padded size, and
// X / UH B / Vw are [Bt, N, .] arrays, so bcbf_gp_append must copy every per-instance array on every append and re-pack
// the whole operator whenever N crosses a multiple of 32.  Reserved storage lays the operator out ONCE for Ncap points
// (rows / columns beyond the live N are identity padding) and gives the arrays Ncap rows per instance: an append then
// writes one operator row, one inverted-diagonal-block row and one row of each array IN PLACE -- O(N) bytes instead of
// O(N^2), no allocation, nothing copied.  The kernels that read it take (N, Ncap): bcbf_posterior_query_reserved.
template <typename T>
__global__ void __launch_bounds__(ST)
gp_reserve_kernel(const T* __restrict__ Lin, const T* __restrict__ Vw_in, const T* __restrict__ X_in,
                  const T* __restrict__ UHB_in, T* __restrict__ Lout, T* __restrict__ Vw_out, T* __restrict__ X_out,
                  T* __restrict__ UHB_out, int N, int NcapIn, int Ncap, int n, int C) {
    // NcapIn = 0: the inputs are the packed state of exactly N points; > 0: reserved storage of that capacity (growing)
    constexpr int V = Vec<T>::V;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ldI = NcapIn ? NcapIn : N;
    const int NpI = round_up(ldI, NB), NpO = round_up(Ncap, NB);
    relayout_operator<T>(Lin + (size_t)b * lop_elems<V>(NpI), Lout + (size_t)b * lop_elems<V>(NpO), NpI, NpO, tid);
    for (int e = tid; e < Ncap * n; e += ST) {
        const bool live = e < N * n;
        Vw_out[(size_t)b * Ncap * n + e] = live ? Vw_in[(size_t)b * ldI * n + e] : T(0);
        X_out[(size_t)b * Ncap * n + e] = live ? X_in[(size_t)b * ldI * n + e] : T(0);
    }
    for (int e = tid; e < Ncap * C; e += ST) UHB_out[(size_t)b * Ncap * C + e] = e < N * C ? UHB_in[(size_t)b * ldI * C + e] : T(0);
}

// One observation per instance enters reserved storage in place.  W = L^-1 Phi(x_new) [Bt, Np, C] comes from the
// streaming posterior kernel (one pass over the factor at the HBM rate); l = W uh_new is the new operator row,
// d = sqrt(kappa - l'l) its pivot.  A non-positive pivot (info = N+1) leaves the operator as it was (row N is still an
// identity padding row) and enters a NEUTRAL point (Vw row 0, UH B row 0), as bcbf_gp_append does.
template <typename T>
__global__ void __launch_bounds__(ST)
gp_append_inplace_kernel(T* __restrict__ Lop, T* __restrict__ Vw, T* __restrict__ X, T* __restrict__ UHB,
                         const T* __restrict__ s2, const T* __restrict__ Bm, const T* __restrict__ M0,
                         const T* __restrict__ x_new, const T* __restrict__ uh_new, const T* __restrict__ xdot_new,
                         const T* __restrict__ jitter_new, const T* __restrict__ W, int* __restrict__ info,
                         int N, int Ncap, int n, int C, int wq, const T* __restrict__ Mk2, const T* __restrict__ Bk2,
                         T* __restrict__ Mk_out, T* __restrict__ Bk_out, const T* __restrict__ lsum = nullptr,
                         T* __restrict__ rawUH = nullptr, T* __restrict__ rawY = nullptr, T* __restrict__ rawJ = nullptr) {
    // rawUH / rawY / rawJ (optional, [Bt, Ncap, C] / [Bt, Ncap, n] / [Bt, Ncap]): the caller's store of the RAW rows (uh, xdot,
    // jitter) a later refit of a sliding window is made from -- row N is written here, neutral (0, 0, unit pivot) where the
    // new pivot failed, so that a refit sees exactly what the in-place path holds (ops.ReservedGP(window=...))
    // wq = 2: W / Mk2 / Bk2 hold TWO queries per instance (slot 0 = the caller's posterior query, slot 1 = x_new, both
    // from one pass over the factor); slot 0's posterior is handed to Mk_out / Bk_out here
    // wq = 3: W IS the column l [Bt, Np] (posterior_step_kernel<.., XC = 1>) and lsum[Bt, 1 + n] = (l'l, Vw'l) came with it
    constexpr int V = Vec<T>::V;
    __shared__ T lrow[ST * SMAXR];
    __shared__ T scratch[4 * SC];
    __shared__ T dinv_s[LOP_DB];                                // the inverted diagonal block the new row extends
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Np = round_up(N, NB), Nl = round_up(Ncap, NB);
    T* lop = Lop + (size_t)b * lop_elems<V>(Nl);
    T* Vwb = Vw + (size_t)b * Ncap * n;
    // (fetched by the whole workgroup up front: the new row of inv(L_JJ) is a chain of up to 31 multiply-adds per lane, and
    //  with the operands in global memory every term waited out a memory round trip -- a quarter of the online step at
    //  4096 instances)
    for (int k = tid; k < LOP_DB; k += ST) dinv_s[k] = lop[lop_dinv_block(N / NB, Nl) + k];
    T uh[BCBF_MAX_CTRL_DIM + 1];
#pragma unroll
    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) uh[c] = c < C ? uh_new[(size_t)b * C + c] : T(0);
    // l = W uh_new; l'l and l'Vw in the same pass
    T acc[SC + 1];
#pragma unroll
    for (int c = 0; c < SC + 1; ++c) acc[c] = T(0);
    T ssq[1];
    if (wq == 3) {
        for (int i = tid; i < N; i += ST) lrow[i] = W[(size_t)b * Np + i];
        ssq[0] = lsum[(size_t)b * (1 + n)];
#pragma unroll
        for (int c = 0; c < SC; ++c) acc[c] = c < n ? lsum[(size_t)b * (1 + n) + 1 + c] : T(0);
    } else {
    for (int i = tid; i < N; i += ST) {
        T v = T(0);
#pragma unroll
        for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c)
            if (c < C) v += W[((size_t)(b * wq + wq - 1) * Np + i) * C + c] * uh[c];
        lrow[i] = v;
        acc[SC] += v * v;
#pragma unroll
        for (int c = 0; c < SC; ++c)
            if (c < n) acc[c] += v * Vwb[(size_t)i * n + c];
    }
    ssq[0] = acc[SC];
    block_sum(ssq, 1, scratch);
    __syncthreads();
    block_sum(acc, n, scratch);
    }
    T q = T(0);
#pragma unroll
    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) {
        T sacc = T(0);
        if (c < C) for (int a = 0; a < C; ++a) sacc += Bm[((size_t)b * C + c) * C + a] * uh[a];
        q += uh[c] * sacc;
    }
    const T kap = s2[b] * q + (jitter_new ? jitter_new[b] : T(0));
    const T d2 = kap - ssq[0];
    const bool ok = d2 > T(0);
    const T d = ok ? (T)sqrt((double)d2) : T(1);
    if (tid == 0) info[b] = ok ? 0 : N + 1;
    __syncthreads();                                            // lrow complete for every thread
    const int Js = N / NB, col0 = Js * NB, rr = N - col0;
    if (ok) {
        // (non-temporal: one element per 128-byte line of every column -- written through instead of left dirty in L2, where the
        //  partial lines drained while the NEXT streaming pass ran and cost it far more than their bytes)
        for (int j = tid; j < col0; j += ST) __builtin_nontemporal_store(lrow[j], &lop[lop_base<V>(j, Nl) + N]);
        if (tid <= rr) {
            const int jj = tid;                                 // column inside the diagonal block
            T val;
            if (jj == rr) val = T(1) / d;
            else {
                T a2 = T(0);
                for (int ii = jj; ii < rr; ++ii) a2 += lrow[col0 + ii] * dinv_s[lop_dinv_col(jj) + ii];
                val = -a2 / d;
            }
            __builtin_nontemporal_store(val, &lop[lop_dinv(Js, rr, jj, Nl)]);
            __builtin_nontemporal_store(val, &lop[lop_dfull(Js, rr, jj, Nl)]);      // (a row of a column-major tile: 32 lines)
        }
    }
    if (tid < n) {
        T yv = xdot_new[(size_t)b * n + tid];
        for (int a = 0; a < C; ++a) yv -= uh[a] * M0[((size_t)b * C + a) * n + tid];
        Vwb[(size_t)N * n + tid] = ok ? (yv - acc[tid]) / d : T(0);
        X[((size_t)b * Ncap + N) * n + tid] = x_new[(size_t)b * n + tid];
    }
    if (tid < C) {
        T sacc = T(0);
        for (int a = 0; a < C; ++a) sacc += uh[a] * Bm[((size_t)b * C + a) * C + tid];
        UHB[((size_t)b * Ncap + N) * C + tid] = ok ? sacc : T(0);
        if (rawUH != nullptr) rawUH[((size_t)b * Ncap + N) * C + tid] = ok ? uh[tid] : T(0);
    }
    if (rawY != nullptr && tid < n) rawY[((size_t)b * Ncap + N) * n + tid] = ok ? xdot_new[(size_t)b * n + tid] : T(0);
    if (rawJ != nullptr && tid == 0) rawJ[(size_t)b * Ncap + N] = ok ? (jitter_new ? jitter_new[b] : T(0)) : T(1);
    if (wq == 2 && Mk_out != nullptr) {
        if (tid < n * C) Mk_out[(size_t)b * n * C + tid] = Mk2[(size_t)(2 * b) * n * C + tid];
        else if (tid >= 64 && tid < 64 + C * C) Bk_out[(size_t)b * C * C + tid - 64] = Bk2[(size_t)(2 * b) * C * C + tid - 64];
    }
}

template <typename T>
static int launch_gp_reserve(const T* Lin, const T* Vw_in, const T* X_in, const T* UHB_in, T* Lout, T* Vw_out, T* X_out,
                             T* UHB_out, int Bt, int N, int NcapIn, int Ncap, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lin || !Vw_in || !X_in || !UHB_in || !Lout || !Vw_out || !X_out || !UHB_out) return BCBF_EINVAL;
    if (N < 1 || Ncap < N || round_up(Ncap, NB) > ST * SMAXR) return BCBF_EINVAL;
    if (NcapIn != 0 && (NcapIn < N || NcapIn > Ncap)) return BCBF_EINVAL;
    if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    if (Lin == Lout || Vw_in == Vw_out || X_in == X_out || UHB_in == UHB_out) return BCBF_EINVAL;
    hipLaunchKernelGGL((gp_reserve_kernel<T>), dim3(Bt), dim3(ST), 0, (hipStream_t)stream, Lin, Vw_in, X_in, UHB_in, Lout,
                       Vw_out, X_out, UHB_out, N, NcapIn, Ncap, n, m + 1);
    return check_launch("gp_reserve");
}

template <typename T>
int launch_posterior_pair_reserved(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2, const T* Bm,
                                   const T* M0, const T* xq, const T* xq2, T* Mk2, T* Bk2, T* W2, int Bt, int N, int Ncap, int n,
                                   int m, void* stream, int kind = 0);          // posterior_step.hip
template <typename T>
int launch_posterior_query_column_reserved(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2,
                                           const T* Bm, const T* M0, const T* xq, const T* x_new, const T* uh_new, T* Mk, T* Bk,
                                           T* lvec, T* lsum, int Bt, int N, int Ncap, int n, int m, void* stream,
                                           T* Wfull = nullptr, int Lcap = 0, int kind = 0);   // posterior_step.hip

template <typename T>
static int launch_gp_append_inplace(T* Lop, T* Vw, T* X, T* UHB, const T* s2, const T* Bm, const T* M0, const T* x_new,
                                    const T* uh_new, const T* xdot_new, const T* jitter_new, const T* W, int* info, int Bt,
                                    int N, int Ncap, int n, int m, void* stream, int wq = 1, const T* Mk2 = nullptr,
                                    const T* Bk2 = nullptr, T* Mk_out = nullptr, T* Bk_out = nullptr, const T* lsum = nullptr,
                                    T* const* raw = nullptr) {
    hipLaunchKernelGGL((gp_append_inplace_kernel<T>), dim3(Bt), dim3(ST), 0, (hipStream_t)stream, Lop, Vw, X, UHB, s2, Bm, M0,
                       x_new, uh_new, xdot_new, jitter_new, W, info, N, Ncap, n, m + 1, wq, Mk2, Bk2, Mk_out, Bk_out, lsum,
                       raw ? raw[0] : (T*)nullptr, raw ? raw[1] : (T*)nullptr, raw ? raw[2] : (T*)nullptr);
    return check_launch("gp_append_reserved");
}
}  // namespace bcbf

extern "C" {
int bcbf_potrs_f32(const float* Lop, const float* Xdot, const float* UH, const float* M0,
                   float* Vw, float* alpha, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_potrs<float>(Lop, Xdot, UH, M0, Vw, alpha, Bt, N, n, m, stream);
}
int bcbf_potrs_f64(const double* Lop, const double* Xdot, const double* UH, const double* M0,
                   double* Vw, double* alpha, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_potrs<double>(Lop, Xdot, UH, M0, Vw, alpha, Bt, N, n, m, stream);
}
int bcbf_potri_f32(const float* Lop, float* Kinv, int Bt, int N, void* stream) {
    return bcbf::launch_potri<float>(Lop, Kinv, Bt, N, stream);
}
int bcbf_potri_f64(const double* Lop, double* Kinv, int Bt, int N, void* stream) {
    return bcbf::launch_potri<double>(Lop, Kinv, Bt, N, stream);
}
// Dense inverse of the factor itself, Linv[Bt,N,N] = L^-1 (lower triangular).  Batches: the blocked matrix-core form (trtri.hip:
// one wave per block column, N^3 / 3 flops per model on 32 x 32 tiles).  A handful of models: the forward half of bcbf_potri
// (N / 8 workgroups per model, each a forward solve of 8 identity columns -- more parallel for ONE model, N / 8 passes over the
// factor each).  K_b^-1 = Linv' Linv is then bcbf_syrk_lt.  BCBF_TRTRI_SOLVE=1 forces the solve form (development).
int bcbf_trtri_f32(const float* Lop, float* Linv, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !Linv || N < 1) return BCBF_EINVAL;
    static const bool solve = [] { const char* e = getenv("BCBF_TRTRI_SOLVE"); return e && e[0] == '1'; }();
    if (Bt < 4 || solve) return bcbf::launch_potri<float>(Lop, Linv, Bt, N, stream, 1);
    return bcbf::launch_trtri_mfma_f32(Lop, Linv, Bt, N, stream);
}
int bcbf_trtri_f64(const double* Lop, double* Linv, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !Linv || N < 1) return BCBF_EINVAL;
    static const bool solve = [] { const char* e = getenv("BCBF_TRTRI_SOLVE"); return e && e[0] == '1'; }();
    if (Bt < 4 || solve) return bcbf::launch_potri<double>(Lop, Linv, Bt, N, stream, 1);
    return bcbf::launch_trtri_mfma_f64(Lop, Linv, Bt, N, stream);
}
int bcbf_gp_append_f32(const float* Lop_in, const float* Vw_in, const float* X_in, const float* UHB_in,
                       const float* ell, const float* s2, const float* Bm, const float* M0, const float* x_new,
                       const float* uh_new, const float* xdot_new, const float* jitter_new, float* Lop_out,
                       float* Vw_out, float* X_out, float* UHB_out, int* info, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_gp_append<float>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                         jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream);
}
int bcbf_gp_append_f64(const double* Lop_in, const double* Vw_in, const double* X_in, const double* UHB_in,
                       const double* ell, const double* s2, const double* Bm, const double* M0, const double* x_new,
                       const double* uh_new, const double* xdot_new, const double* jitter_new, double* Lop_out,
                       double* Vw_out, double* X_out, double* UHB_out, int* info, int Bt, int N, int n, int m,
                       void* stream) {
    return bcbf::launch_gp_append<double>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                          jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream);
}
// The same update with the forward solve on the streaming posterior kernel: W = L^-1 Phi(x_new) (bcbf_posterior_query,
// one query per instance, at the HBM roofline) and l = W uh_new; the append kernel then only copies / re-packs and writes
// the new rows.  Work buffers from the caller (the library allocates nothing): Wwork[Bt, Np, 1+m] (Np = N rounded up to
// 32), Mk_work[Bt, n, 1+m], Bk_work[Bt, 1+m, 1+m].
int bcbf_gp_append_stream_f32(const float* Lop_in, const float* Vw_in, const float* X_in, const float* UHB_in,
                              const float* ell, const float* s2, const float* Bm, const float* M0, const float* x_new,
                              const float* uh_new, const float* xdot_new, const float* jitter_new, float* Lop_out,
                              float* Vw_out, float* X_out, float* UHB_out, int* info, float* Wwork, float* Mk_work,
                              float* Bk_work, int Bt, int N, int n, int m, void* stream) {
    if (!Wwork || !Mk_work || !Bk_work) return BCBF_EINVAL;
    const int rc = bcbf_posterior_query_f32(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, nullptr, Mk_work, Bk_work,
                                            Wwork, 0, Bt, N, n, m, stream);
    if (rc != BCBF_OK) return rc;
    return bcbf::launch_gp_append<float>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                         jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream, Wwork);
}
int bcbf_gp_append_stream_f64(const double* Lop_in, const double* Vw_in, const double* X_in, const double* UHB_in,
                              const double* ell, const double* s2, const double* Bm, const double* M0, const double* x_new,
                              const double* uh_new, const double* xdot_new, const double* jitter_new, double* Lop_out,
                              double* Vw_out, double* X_out, double* UHB_out, int* info, double* Wwork, double* Mk_work,
                              double* Bk_work, int Bt, int N, int n, int m, void* stream) {
    if (!Wwork || !Mk_work || !Bk_work) return BCBF_EINVAL;
    const int rc = bcbf_posterior_query_f64(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, nullptr, Mk_work, Bk_work,
                                            Wwork, 0, Bt, N, n, m, stream);
    if (rc != BCBF_OK) return rc;
    return bcbf::launch_gp_append<double>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                          jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream, Wwork);
}
// ... bcbf_gp_append_stream with the opt-in data kernels (kernel_kind 0 RBF, 1 Matern-5/2, 2 RBF x Matern-5/2): the forward solve on
// the streaming kernel of that kind
#define BCBF_STREAM_KIND_ENTRY(SUF, T)                                                                                       \
    int bcbf_gp_append_stream_kind_##SUF(const T* Lop_in, const T* Vw_in, const T* X_in, const T* UHB_in, const T* ell,       \
                                         const T* s2, const T* Bm, const T* M0, const T* x_new, const T* uh_new,             \
                                         const T* xdot_new, const T* jitter_new, T* Lop_out, T* Vw_out, T* X_out,            \
                                         T* UHB_out, int* info, T* Wwork, T* Mk_work, T* Bk_work, int Bt, int N, int n,       \
                                         int m, int kernel_kind, void* stream) {                                             \
        if (!Wwork || !Mk_work || !Bk_work || kernel_kind < 0 || kernel_kind >= bcbf::BCBF_KINDS) return BCBF_EINVAL;              \
        const int rc = kernel_kind == 0 ? bcbf_posterior_query_##SUF(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new,      \
                                                                     nullptr, Mk_work, Bk_work, Wwork, 0, Bt, N, n, m, stream) \
                     : kernel_kind == 1 ? bcbf_posterior_query_matern52_##SUF(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0,   \
                                                                              x_new, nullptr, Mk_work, Bk_work, Wwork, 0, Bt, \
                                                                              N, n, m, stream)                              \
                                        : bcbf_posterior_query_rbfm52_##SUF(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0,     \
                                                                            x_new, nullptr, Mk_work, Bk_work, Wwork, 0, Bt,   \
                                                                            N, n, m, stream);                               \
        if (rc != BCBF_OK) return rc;                                                                                        \
        return bcbf::launch_gp_append<T>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new,   \
                                         Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream, Wwork);                 \
    }
BCBF_STREAM_KIND_ENTRY(f32, float)
BCBF_STREAM_KIND_ENTRY(f64, double)
#undef BCBF_STREAM_KIND_ENTRY
// ... with the opt-in Matern-5/2 data kernel for the new column (the simple forward solve; no reference counterpart)
int bcbf_gp_append_matern52_f32(const float* Lop_in, const float* Vw_in, const float* X_in, const float* UHB_in,
                                const float* ell, const float* s2, const float* Bm, const float* M0, const float* x_new,
                                const float* uh_new, const float* xdot_new, const float* jitter_new, float* Lop_out,
                                float* Vw_out, float* X_out, float* UHB_out, int* info, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_gp_append<float>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                         jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream, nullptr, 1);
}
int bcbf_gp_append_matern52_f64(const double* Lop_in, const double* Vw_in, const double* X_in, const double* UHB_in,
                                const double* ell, const double* s2, const double* Bm, const double* M0, const double* x_new,
                                const double* uh_new, const double* xdot_new, const double* jitter_new, double* Lop_out,
                                double* Vw_out, double* X_out, double* UHB_out, int* info, int Bt, int N, int n, int m,
                                void* stream) {
    return bcbf::launch_gp_append<double>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                          jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream, nullptr, 1);
}
int bcbf_gp_append_rbfm52_f32(const float* Lop_in, const float* Vw_in, const float* X_in, const float* UHB_in,
                              const float* ell, const float* s2, const float* Bm, const float* M0, const float* x_new,
                              const float* uh_new, const float* xdot_new, const float* jitter_new, float* Lop_out,
                              float* Vw_out, float* X_out, float* UHB_out, int* info, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_gp_append<float>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                         jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream, nullptr, 2);
}
int bcbf_gp_append_rbfm52_f64(const double* Lop_in, const double* Vw_in, const double* X_in, const double* UHB_in,
                              const double* ell, const double* s2, const double* Bm, const double* M0, const double* x_new,
                              const double* uh_new, const double* xdot_new, const double* jitter_new, double* Lop_out,
                              double* Vw_out, double* X_out, double* UHB_out, int* info, int Bt, int N, int n, int m,
                              void* stream) {
    return bcbf::launch_gp_append<double>(Lop_in, Vw_in, X_in, UHB_in, ell, s2, Bm, M0, x_new, uh_new, xdot_new,
                                          jitter_new, Lop_out, Vw_out, X_out, UHB_out, info, Bt, N, n, m, stream, nullptr, 2);
}
int bcbf_chol_append_f32(const float* Lop_in, const float* knew, const float* kappa, float* Lop_out,
                         int* info, int Bt, int N, void* stream) {
    return bcbf::launch_chol_append<float>(Lop_in, knew, kappa, Lop_out, info, Bt, N, stream);
}
int bcbf_chol_append_f64(const double* Lop_in, const double* knew, const double* kappa, double* Lop_out,
                         int* info, int Bt, int N, void* stream) {
    return bcbf::launch_chol_append<double>(Lop_in, knew, kappa, Lop_out, info, Bt, N, stream);
}

// Capacity-reserving storage of the online path (see gp_reserve_kernel above).
#define BCBF_RESERVED_ENTRY(SUF, T)                                                                                      \
    int bcbf_gp_reserve_##SUF(const T* Lop_in, const T* Vw_in, const T* X_in, const T* UHB_in, T* Lop_r, T* Vw_r, T* X_r,   \
                              T* UHB_r, int Bt, int N, int Ncap_in, int Ncap, int n, int m, void* stream) {             \
        return bcbf::launch_gp_reserve<T>(Lop_in, Vw_in, X_in, UHB_in, Lop_r, Vw_r, X_r, UHB_r, Bt, N, Ncap_in, Ncap, n, m,  \
                                          stream);                                                                       \
    }                                                                                                                    \
    static int gp_append_reserved_impl_##SUF(T* Lop_r, T* Vw_r, T* X_r, T* UHB_r, const T* ell, const T* s2, const T* Bm,   \
                                      const T* M0, const T* x_new, const T* uh_new, const T* xdot_new,                   \
                                      const T* jitter_new, int* info, T* Wwork, T* Mk_work, T* Bk_work, const T* xq,     \
                                      T* Mk, T* Bk, int Bt, int N, int Ncap, int n, int m, void* stream, T* const* raw,  \
                                      int kind = 0) {                                                                    \
        if (Bt <= 0) return BCBF_OK;                                                                                     \
        if (kind < 0 || kind >= bcbf::BCBF_KINDS) return BCBF_EINVAL;                                                    \
        if (!Lop_r || !Vw_r || !X_r || !UHB_r || !ell || !s2 || !Bm || !M0 || !x_new || !uh_new || !xdot_new || !info ||  \
            !Wwork || !Mk_work || !Bk_work || (xq && (!Mk || !Bk)))                                                      \
            return BCBF_EINVAL;                                                                                          \
        if (N < 1 || N >= Ncap || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;  \
        if (bcbf::round_up(Ncap, bcbf::NB) > bcbf::ST * bcbf::SMAXR) return BCBF_EINVAL;                                  \
        if (xq != nullptr && n <= 4 && !getenv("BCBF_APPEND_PAIR")) {                                                    \
            /* the caller's posterior query rides along: its C columns + the ONE column the append needs, one pass over  \
               the factor (Wwork: the column l [Bt, Np]; Mk_work: its sums [Bt, 1 + n]) */                               \
            const int rc = bcbf::launch_posterior_query_column_reserved<T>(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, xq,  \
                                                                           x_new, uh_new, Mk, Bk, Wwork, Mk_work, Bt, N,  \
                                                                           Ncap, n, m, stream, nullptr, 0, kind);        \
            if (rc != BCBF_OK) return rc;                                                                                \
            return bcbf::launch_gp_append_inplace<T>(Lop_r, Vw_r, X_r, UHB_r, s2, Bm, M0, x_new, uh_new, xdot_new,        \
                                                     jitter_new, Wwork, info, Bt, N, Ncap, n, m, stream, 3, nullptr,      \
                                                     nullptr, nullptr, nullptr, Mk_work, raw);                           \
        }                                                                                                                \
        if (xq != nullptr && n <= 4 && m <= 2) {                                                                         \
            /* (round 3's form, BCBF_APPEND_PAIR=1: two full queries per instance on one pass) */                        \
            const int rc = bcbf::launch_posterior_pair_reserved<T>(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, xq, x_new,   \
                                                                   Mk_work, Bk_work, Wwork, Bt, N, Ncap, n, m, stream,   \
                                                                   kind);                                                \
            if (rc != BCBF_OK) return rc;                                                                                \
            return bcbf::launch_gp_append_inplace<T>(Lop_r, Vw_r, X_r, UHB_r, s2, Bm, M0, x_new, uh_new, xdot_new,        \
                                                     jitter_new, Wwork, info, Bt, N, Ncap, n, m, stream, 2, Mk_work,      \
                                                     Bk_work, Mk, Bk, nullptr, raw);                                     \
        }                                                                                                                \
        if (xq != nullptr) {                                                                                             \
            const int rq = bcbf_posterior_query_reserved_kind_##SUF(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, xq, nullptr, \
                                                                    Mk, Bk, nullptr, Bt, N, Ncap, n, m, kind, stream);   \
            if (rq != BCBF_OK) return rq;                                                                                \
        }                                                                                                                \
        const int rc = bcbf_posterior_query_reserved_kind_##SUF(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, x_new, nullptr,  \
                                                                Mk_work, Bk_work, Wwork, Bt, N, Ncap, n, m, kind, stream); \
        if (rc != BCBF_OK) return rc;                                                                                    \
        return bcbf::launch_gp_append_inplace<T>(Lop_r, Vw_r, X_r, UHB_r, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new, \
                                                 Wwork, info, Bt, N, Ncap, n, m, stream, 1, nullptr, nullptr, nullptr,   \
                                                 nullptr, nullptr, raw);                                                 \
    }                                                                                                                    \
    int bcbf_gp_append_reserved_##SUF(T* Lop_r, T* Vw_r, T* X_r, T* UHB_r, const T* ell, const T* s2, const T* Bm,         \
                                      const T* M0, const T* x_new, const T* uh_new, const T* xdot_new,                   \
                                      const T* jitter_new, int* info, T* Wwork, T* Mk_work, T* Bk_work, const T* xq,     \
                                      T* Mk, T* Bk, int Bt, int N, int Ncap, int n, int m, void* stream) {               \
        return gp_append_reserved_impl_##SUF(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new, \
                                             info, Wwork, Mk_work, Bk_work, xq, Mk, Bk, Bt, N, Ncap, n, m, stream, nullptr); \
    }                                                                                                                    \
    /* ... that also records the RAW rows of the new point (what a sliding window's refit is made from) */              \
    int bcbf_gp_append_reserved_raw_##SUF(T* Lop_r, T* Vw_r, T* X_r, T* UHB_r, const T* ell, const T* s2, const T* Bm,     \
                                          const T* M0, const T* x_new, const T* uh_new, const T* xdot_new,               \
                                          const T* jitter_new, int* info, T* Wwork, T* Mk_work, T* Bk_work, const T* xq, \
                                          T* Mk, T* Bk, T* rawUH, T* rawY, T* rawJ, int Bt, int N, int Ncap, int n,      \
                                          int m, void* stream) {                                                         \
        if (!rawUH || !rawY || !rawJ) return BCBF_EINVAL;                                                                \
        T* const raw[3] = {rawUH, rawY, rawJ};                                                                           \
        return gp_append_reserved_impl_##SUF(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new, \
                                             info, Wwork, Mk_work, Bk_work, xq, Mk, Bk, Bt, N, Ncap, n, m, stream, raw);  \
    }                                                                                                                    \
    /* ... with the opt-in data kernels (kernel_kind 0 RBF, 1 Matern-5/2, 2 RBF x Matern-5/2; a state built by the matching \
       bcbf_refit_* and bcbf_gp_reserve); rawUH = rawY = rawJ = NULL: no raw store */                                    \
    int bcbf_gp_append_reserved_kind_##SUF(T* Lop_r, T* Vw_r, T* X_r, T* UHB_r, const T* ell, const T* s2, const T* Bm,    \
                                           const T* M0, const T* x_new, const T* uh_new, const T* xdot_new,              \
                                           const T* jitter_new, int* info, T* Wwork, T* Mk_work, T* Bk_work,             \
                                           const T* xq, T* Mk, T* Bk, T* rawUH, T* rawY, T* rawJ, int Bt, int N, int Ncap, \
                                           int n, int m, int kernel_kind, void* stream) {                                \
        if ((rawUH || rawY || rawJ) && !(rawUH && rawY && rawJ)) return BCBF_EINVAL;                                     \
        T* const raw[3] = {rawUH, rawY, rawJ};                                                                           \
        return gp_append_reserved_impl_##SUF(Lop_r, Vw_r, X_r, UHB_r, ell, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new, \
                                             info, Wwork, Mk_work, Bk_work, xq, Mk, Bk, Bt, N, Ncap, n, m, stream,        \
                                             rawUH ? raw : nullptr, kernel_kind);                                        \
    }
BCBF_RESERVED_ENTRY(f32, float)
BCBF_RESERVED_ENTRY(f64, double)
#undef BCBF_RESERVED_ENTRY
}
