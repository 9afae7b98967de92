// K1 + K2 (+ packing): kernel-matrix build and blocked left-looking Cholesky, one workgroup per
// instance, writing the packed triangular operator the per-step kernel streams (bcbf.h "Lop").
//
// For block column J (32 columns):
//   A. every thread owns rows i >= 32J (two at a time) and forms
//        S[i][c] = K_b(i, 32J+c) - sum_{kk < 32J} L[i][kk] L[32J+c][kk]
//      K_b is generated on the fly from (X, UH Bm) or read from a dense input; previous panels are
//      read back from the packed operator (column-major => coalesced over rows); the 32x32 tile of
//      block row J is staged through LDS and broadcast;
//   B. wave 0 factors the 32x32 diagonal block in registers (lane = row, v_readlane broadcasts)
//      and inverts it (lane = column of the inverse);
//   C. rows below the block: L[i][J] = S[i][:] inv(L_JJ)^T, written column-major.
#include "bcbf_common.h"

namespace bcbf {

template <typename T> __device__ inline T tsqrt(T x);
template <> __device__ inline float tsqrt<float>(float x) { return sqrtf(x); }
template <> __device__ inline double tsqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ inline T texp2(T x);
template <> __device__ inline float texp2<float>(float x) { return expf(x); }
template <> __device__ inline double texp2<double>(double x) { return exp(x); }

template <typename T> __device__ inline T lane_bcast(T v, int lane) { return __shfl(v, lane, 64); }

constexpr int RT = 256;        // threads per workgroup
constexpr int TP = NB + 4;     // padded LDS row (keeps 16-byte alignment, breaks the 32-stride)

// --------------------------------------------------------------------------------------------
template <typename T, bool FROM_DENSE>
__global__ void __launch_bounds__(RT)
refit_kernel(const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ Bm,
             const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ jitter,
             const T* __restrict__ Kdense, T* __restrict__ Lop, T* __restrict__ UHBout,
             T* __restrict__ Ldense, int* __restrict__ info, int N, int Np, int n, int C) {
    constexpr int V = Vec<T>::V;
    __shared__ __attribute__((aligned(16))) T tile[NB][TP];     // tile[kk][c] = L[32J+c][32K+kk]
    __shared__ __attribute__((aligned(16))) T dblk[NB][TP];     // diagonal block S, then inv(L_JJ)[c][c']
    __shared__ T colX[NB][BCBF_MAX_STATE_DIM];
    __shared__ T colUH[NB][BCBF_MAX_CTRL_DIM + 1];
    __shared__ int fail;

    const int b = blockIdx.x, tid = threadIdx.x;
    T* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    const T* Xb = FROM_DENSE ? nullptr : X + (size_t)b * N * n;
    const T* UHb = FROM_DENSE ? nullptr : UH + (size_t)b * N * C;
    const T* Kb = FROM_DENSE ? Kdense + (size_t)b * N * N : nullptr;
    T* Ld = Ldense ? Ldense + (size_t)b * N * N : nullptr;
    T iell[BCBF_MAX_STATE_DIM];
    T s2 = T(0);
    T Bmr[(BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1)];
    if (!FROM_DENSE) {
        s2 = s2p[b];
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) iell[d] = d < n ? T(1) / ell[(size_t)b * n + d] : T(0);
#pragma unroll
        for (int a = 0; a < (BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1); ++a)
            Bmr[a] = a < C * C ? Bm[(size_t)b * C * C + a] : T(0);
        // UHB = UH Bm (also an output: the per-step kernel reads it)
        for (int i = tid; i < N; i += RT) {
            for (int c = 0; c < C; ++c) {
                T s = T(0);
                for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                UHBout[((size_t)b * N + i) * C + c] = s;
            }
        }
    }
    if (tid == 0) fail = 0;
    if (Ld) {  // zero the strict upper triangle of the dense output
        for (int e = tid; e < N * N; e += RT) { const int i = e / N, j = e - i * N; if (j > i) Ld[e] = T(0); }
    }
    __syncthreads();

    const int nblk = Np / NB;
    for (int J = 0; J < nblk; ++J) {
        const int col0 = J * NB;
        if (!FROM_DENSE) {   // stage x_j, uh_j of the 32 columns
            for (int e = tid; e < NB * n; e += RT) {
                const int c = e / n, d = e - c * n;
                colX[c][d] = (col0 + c < N) ? Xb[(size_t)(col0 + c) * n + d] : T(0);
            }
            for (int e = tid; e < NB * C; e += RT) {
                const int c = e / C, a = e - c * C;
                colUH[c][a] = (col0 + c < N) ? UHb[(size_t)(col0 + c) * C + a] : T(0);
            }
        }
        __syncthreads();
        const int nrows = Np - col0;
        for (int q0 = 0; q0 < nrows; q0 += 2 * RT) {
            // ---- phase A: S for rows i0, i1
            const int i0 = col0 + q0 + tid, i1 = i0 + RT;
            const bool ok0 = i0 < Np, ok1 = i1 < Np;
            T S0[NB], S1[NB];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = r == 0 ? i0 : i1;
                const bool ok = r == 0 ? ok0 : ok1;
                T* S = r == 0 ? S0 : S1;
                if (!ok) {
#pragma unroll
                    for (int c = 0; c < NB; ++c) S[c] = T(0);
                    continue;
                }
                if (i >= N) {   // padding rows: identity
#pragma unroll
                    for (int c = 0; c < NB; ++c) S[c] = (col0 + c == i) ? T(1) : T(0);
                } else if (FROM_DENSE) {
#pragma unroll
                    for (int c = 0; c < NB; ++c) S[c] = (col0 + c <= i && col0 + c < N) ? Kb[(size_t)i * N + col0 + c] : T(0);
                } else {
                    T xi[BCBF_MAX_STATE_DIM], ub[BCBF_MAX_CTRL_DIM + 1];
#pragma unroll
                    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) xi[d] = d < n ? Xb[(size_t)i * n + d] : T(0);
#pragma unroll
                    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) {
                        T s = T(0);
                        if (c < C) for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                        ub[c] = s;
                    }
                    const T jit = jitter ? jitter[(size_t)b * N + i] : T(0);
#pragma unroll
                    for (int c = 0; c < NB; ++c) {
                        T d2 = T(0), uu = T(0);
#pragma unroll
                        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d)
                            if (d < n) { const T z = (xi[d] - colX[c][d]) * iell[d]; d2 += z * z; }
#pragma unroll
                        for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a)
                            if (a < C) uu += ub[a] * colUH[c][a];
                        T val = s2 * texp2<T>(T(-0.5) * d2) * uu;
                        if (col0 + c == i) val += jit;
                        S[c] = (col0 + c < N) ? val : T(0);
                    }
                }
            }
            for (int K = 0; K < J; ++K) {
                __syncthreads();   // previous tile fully consumed
                for (int e = tid; e < NB * NB; e += RT) {
                    const int kk = e / NB, c = e - kk * NB;
                    tile[kk][c] = lop[lop_base<V>(K * NB + kk, Np) + col0 + c];
                }
                __syncthreads();
#pragma unroll 4
                for (int kk = 0; kk < NB; ++kk) {
                    const int base = lop_base<V>(K * NB + kk, Np);
                    const T l0 = ok0 ? lop[base + i0] : T(0);
                    const T l1 = ok1 ? lop[base + i1] : T(0);
#pragma unroll
                    for (int c = 0; c < NB; ++c) {
                        const T tv = tile[kk][c];
                        S0[c] -= l0 * tv;
                        S1[c] -= l1 * tv;
                    }
                }
            }
            // park S: diagonal rows -> LDS, rows below -> their final slots in the operator
            if (q0 == 0 && tid < NB) {
#pragma unroll
                for (int c = 0; c < NB; ++c) dblk[tid][c] = S0[c];
            } else if (ok0) {
#pragma unroll
                for (int c = 0; c < NB; ++c) lop[lop_base<V>(col0 + c, Np) + i0] = S0[c];
            }
            if (ok1) {
#pragma unroll
                for (int c = 0; c < NB; ++c) lop[lop_base<V>(col0 + c, Np) + i1] = S1[c];
            }
        }
        __threadfence_block();
        __syncthreads();

        // ---- phase B: wave 0 factors the diagonal block (lane = row) and inverts it (lane = column)
        if (tid < 64) {
            const int lane = tid;
            T row[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) row[c] = lane < NB ? dblk[lane][c] : T(0);
            int bad = 0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const T piv = lane_bcast(row[c], c);
                if (!(piv > T(0)) && bad == 0) bad = col0 + c + 1;
                const T lcc = tsqrt<T>(piv > T(0) ? piv : T(1));
                const T inv = T(1) / lcc;
                row[c] = lane == c ? lcc : (lane > c ? row[c] * inv : T(0));
#pragma unroll
                for (int c2 = c + 1; c2 < NB; ++c2) {
                    const T lc2 = lane_bcast(row[c], c2);   // L[c2][c]
                    row[c2] -= row[c] * lc2;
                }
            }
            // inverse: lane j solves L x = e_j
            T x[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                T s = lane == i ? T(1) : T(0);
#pragma unroll
                for (int k = 0; k < i; ++k) s -= lane_bcast(row[k], i) * x[k];
                x[i] = s / lane_bcast(row[i], i);
            }
            if (lane < NB) {
                const int j = col0 + lane;
                const int base = lop_base<V>(j, Np);
                const int first = 0;          // the whole diagonal-block column is stored (zeros above the diagonal)
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    dblk[i][lane] = x[i];                       // dblk[c][c'] = inv(L_JJ)[c][c']
                    if (i >= first) lop[base + col0 + i] = x[i];
                }
                if (Ld && j < N) {
#pragma unroll
                    for (int c = 0; c < NB; ++c)
                        if (c <= lane && col0 + c < N) Ld[(size_t)j * N + col0 + c] = row[c];
                }
            }
            if (lane == 0 && bad != 0 && bad <= N) { fail = bad; }
        }
        __syncthreads();
        if (fail != 0) break;

        // ---- phase C: L[i][J] = S[i][:] inv(L_JJ)^T for the rows below the block
        for (int i = col0 + NB + tid; i < Np; i += RT) {
            T S[NB], Lr[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) S[c] = lop[lop_base<V>(col0 + c, Np) + i];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                T s = T(0);
#pragma unroll
                for (int c2 = 0; c2 <= c; ++c2) s += S[c2] * dblk[c][c2];
                Lr[c] = s;
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) lop[lop_base<V>(col0 + c, Np) + i] = Lr[c];
            if (Ld && i < N) {
#pragma unroll
                for (int c = 0; c < NB; ++c) if (col0 + c < N) Ld[(size_t)i * N + col0 + c] = Lr[c];
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) info[b] = fail;
}

// --------------------------------------------------------------------------------------------
// K1 alone: dense K_b (API completeness / parity tests)
template <typename T>
__global__ void kb_build_kernel(const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ Bm,
                                const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ jitter,
                                T* __restrict__ Kb, int N, int n, int C) {
    const int b = blockIdx.y;
    const int i = blockIdx.x;
    const T* Xb = X + (size_t)b * N * n;
    const T* UHb = UH + (size_t)b * N * C;
    const T* Bmb = Bm + (size_t)b * C * C;
    T ub[BCBF_MAX_CTRL_DIM + 1];
    for (int c = 0; c < C; ++c) {
        T s = T(0);
        for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmb[a * C + c];
        ub[c] = s;
    }
    const T s2 = s2p[b];
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        T d2 = T(0), uu = T(0);
        for (int d = 0; d < n; ++d) {
            const T z = (Xb[(size_t)i * n + d] - Xb[(size_t)j * n + d]) / ell[(size_t)b * n + d];
            d2 += z * z;
        }
        for (int a = 0; a < C; ++a) uu += ub[a] * UHb[(size_t)j * C + a];
        T val = s2 * texp2<T>(T(-0.5) * d2) * uu;
        if (i == j && jitter) val += jitter[(size_t)b * N + i];
        Kb[((size_t)b * N + i) * N + j] = val;
    }
}

template <typename T>
static int launch_refit(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2, const T* jitter,
                        const T* Kdense, T* Lop, T* UHB, T* Ldense, int* info, int Bt, int N, int n, int m,
                        void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !info || N < 1) return BCBF_EINVAL;
    const int Np = round_up(N, NB);
    hipStream_t st = (hipStream_t)stream;
    if (Kdense) {
        hipLaunchKernelGGL((refit_kernel<T, true>), dim3(Bt), dim3(RT), 0, st, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, Kdense, Lop, nullptr, Ldense, info, N, Np, 0, 0);
    } else {
        if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
        if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
        hipLaunchKernelGGL((refit_kernel<T, false>), dim3(Bt), dim3(RT), 0, st, X, UH, Bm, ell, s2, jitter,
                           nullptr, Lop, UHB, Ldense, info, N, Np, n, m + 1);
    }
    return check_launch("refit");
}

template <typename T>
static int launch_kb_build(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2, const T* jitter,
                           T* Kb, int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!X || !UH || !Bm || !ell || !s2 || !Kb) return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    hipLaunchKernelGGL((kb_build_kernel<T>), dim3(N, Bt), dim3(256), 0, (hipStream_t)stream, X, UH, Bm, ell, s2,
                       jitter, Kb, N, n, m + 1);
    return check_launch("kb_build");
}

}  // namespace bcbf

extern "C" {
size_t bcbf_lop_elems_f32(int N) { return bcbf::lop_elems<4>(bcbf::round_up(N, bcbf::NB)); }
size_t bcbf_lop_elems_f64(int N) { return bcbf::lop_elems<2>(bcbf::round_up(N, bcbf::NB)); }

int bcbf_kb_build_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                      const float* jitter, float* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<float>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream);
}
int bcbf_kb_build_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                      const double* jitter, double* Kb, int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_kb_build<double>(X, UH, Bm, ell, s2, jitter, Kb, Bt, N, n, m, stream);
}
// fp32 goes to the matrix-core kernel (refit_mfma.hip); the VALU kernel above serves fp64.
extern "C" int bcbf_refit_mfma_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                                   const float* jitter, const float* Kdense, float* Lop, float* UHB, float* Ldense,
                                   int* info, int Bt, int N, int n, int m, void* stream);
int bcbf_refit_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                   const float* jitter, float* Lop, float* UHB, float* Ldense, int* info,
                   int Bt, int N, int n, int m, void* stream) {
    if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
    return bcbf_refit_mfma_f32(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, Ldense, info, Bt, N, n, m, stream);
}
extern "C" int bcbf_refit_mfma_f64(const double* X, const double* UH, const double* Bm, const double* ell,
                                   const double* s2, const double* jitter, const double* Kdense, double* Lop,
                                   double* UHB, double* Ldense, int* info, int Bt, int N, int n, int m, void* stream);
int bcbf_refit_f64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                   const double* jitter, double* Lop, double* UHB, double* Ldense, int* info,
                   int Bt, int N, int n, int m, void* stream) {
    if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
    return bcbf_refit_mfma_f64(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, Ldense, info, Bt, N, n, m, stream);
}
int bcbf_potrf_f32(const float* Kb, float* Lop, float* Ldense, int* info, int Bt, int N, void* stream) {
    if (!Kb) return BCBF_EINVAL;
    return bcbf_refit_mfma_f32(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Kb, Lop, nullptr, Ldense, info, Bt, N, 0, 0, stream);
}
int bcbf_potrf_f64(const double* Kb, double* Lop, double* Ldense, int* info, int Bt, int N, void* stream) {
    if (!Kb) return BCBF_EINVAL;
    return bcbf_refit_mfma_f64(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Kb, Lop, nullptr, Ldense, info, Bt, N, 0, 0, stream);
}
}
