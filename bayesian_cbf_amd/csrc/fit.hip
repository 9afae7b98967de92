// Batched hyper-parameter fit (SURVEY 8f #1 at regime-I scale): what ControlAffineRegressor.fit does per Adam iteration
// AROUND the N^3 / N^2 work (control_affine_model.py:268-335 of the reference: gpytorch's ExactMarginalLogLikelihood, autograd
// through softplus / IndexKernel's W W' + diag(softplus v), torch.optim.Adam under MultiStepLR), for Bt independent models at
// once and without a host round trip:
//   bcbf_fit_derive      raw parameters theta[Bt,P] -> ell, s2, A, B, M0 (what bcbf_refit / bcbf_mll_grad take), A^-1, logdet A
//   bcbf_fit_adam_step   the sums of bcbf_mll_grad (+ A^-1, logdet A) -> loss, d loss / d theta by the chain rule, one Adam update
//   bcbf_kinv_apply      alpha = K_b^-1 R  [Bt,N,nt] from the dense symmetric K_b^-1 of bcbf_trtri + bcbf_syrk_lt
// One iteration of a fit = bcbf_fit_derive, bcbf_refit, bcbf_trtri, bcbf_syrk_lt, bcbf_kinv_apply, bcbf_mll_grad,
// bcbf_fit_adam_step; the host only draws the jitter / target perturbation and looks at `info` (the x10 jitter retry).
//
// theta (one row per model; the reference's raw gpytorch parameters, KernelParams in control_affine_model.py of this package):
//   [ raw_lengthscale (n) | raw_outputscale (1) | U.covar_factor (n x rA, row-major) | U.raw_var (n)
//     | V.covar_factor (C x rB) | V.raw_var (C) | mean constants (C x n) ],   C = 1 + m,
//   ell = softplus(raw), s2 = softplus(raw), A = Wa Wa' + diag softplus(va), B = Wb Wb' + diag softplus(vb), M0 = constants.
// Everything here is a few hundred flops per model: one thread per model, arithmetic in double whatever T is.
#include "bcbf_common.h"
#include <math.h>

namespace bcbf {

struct FitLayout {
    int n, C, rA, rB;
    __host__ __device__ int o_ell() const { return 0; }
    __host__ __device__ int o_s2() const { return n; }
    __host__ __device__ int o_Wa() const { return n + 1; }
    __host__ __device__ int o_va() const { return n + 1 + n * rA; }
    __host__ __device__ int o_Wb() const { return o_va() + n; }
    __host__ __device__ int o_vb() const { return o_Wb() + C * rB; }
    __host__ __device__ int o_M0() const { return o_vb() + C; }
    __host__ __device__ int P() const { return o_M0() + C * n; }
};

// torch.nn.functional.softplus (beta = 1, threshold = 20) and its derivative as autograd forms it
__device__ inline double softplus_d(double x) { return x > 20.0 ? x : log1p(exp(x)); }
__device__ inline double dsoftplus_d(double x) { if (x > 20.0) return 1.0; const double z = exp(x); return z / (z + 1.0); }

constexpr int FN = BCBF_MAX_STATE_DIM, FC = BCBF_MAX_CTRL_DIM + 1;

// S[k x k] = W W' + diag(softplus v)  (W [k x r] row-major)
template <typename T, int KM>
__device__ inline void index_kernel_matrix(const T* W, const T* v, int k, int r, double (&S)[KM][KM]) {
    for (int i = 0; i < k; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int q = 0; q < r; ++q) s += (double)W[i * r + q] * (double)W[j * r + q];
            if (i == j) s += softplus_d((double)v[i]);
            S[i][j] = S[j][i] = s;
        }
}

template <typename T>
__global__ void __launch_bounds__(64)
fit_derive_kernel(const T* __restrict__ theta, T* __restrict__ ell, T* __restrict__ s2, T* __restrict__ A, T* __restrict__ Bm,
                  T* __restrict__ M0, T* __restrict__ Ainv, T* __restrict__ logdetA, int Bt, FitLayout lay) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= Bt) return;
    const int n = lay.n, C = lay.C;
    const T* th = theta + (size_t)b * lay.P();
    for (int d = 0; d < n; ++d) ell[(size_t)b * n + d] = (T)softplus_d((double)th[lay.o_ell() + d]);
    s2[b] = (T)softplus_d((double)th[lay.o_s2()]);
    double Sa[FN][FN], Sb[FC][FC];
    index_kernel_matrix<T, FN>(th + lay.o_Wa(), th + lay.o_va(), n, lay.rA, Sa);
    index_kernel_matrix<T, FC>(th + lay.o_Wb(), th + lay.o_vb(), C, lay.rB, Sb);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[((size_t)b * n + i) * n + j] = (T)Sa[i][j];
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) Bm[((size_t)b * C + i) * C + j] = (T)Sb[i][j];
    for (int i = 0; i < C * n; ++i) M0[(size_t)b * C * n + i] = th[lay.o_M0() + i];
    if (Ainv == nullptr) return;
    // A^-1 and logdet A by Cholesky (A = W W' + positive diagonal: positive definite); the value the likelihood and dA need
    double L[FN][FN], Li[FN][FN], ld = 0.0;
    for (int j = 0; j < n; ++j) {
        double d = Sa[j][j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        d = sqrt(d);
        L[j][j] = d;
        ld += 2.0 * log(d);
        for (int i = j + 1; i < n; ++i) {
            double s = Sa[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s / d;
        }
    }
    for (int c = 0; c < n; ++c)                       // Li = L^-1 (lower), column c by forward substitution
        for (int i = 0; i < n; ++i) {
            if (i < c) { Li[i][c] = 0.0; continue; }
            double s = i == c ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= L[i][k] * Li[k][c];
            Li[i][c] = s / L[i][i];
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = (i > j ? i : j); k < n; ++k) s += Li[k][i] * Li[k][j];
            Ainv[((size_t)b * n + i) * n + j] = (T)s;
        }
    if (logdetA != nullptr) logdetA[b] = (T)ld;
}

// One Adam update of every model from the gradient sums of bcbf_mll_grad.
//   loss = -log p(Y) / (N n)  (- log GammaPrior(ell) / (N n)),   log p = -1/2 tr(A^-1 R'alpha) - n/2 logdet K_b - N/2 logdet A - N n/2 log 2 pi
//   d log p / dA = 1/2 A^-1 (R'alpha) A^-1 - N/2 A^-1,   d log p / dM0 = (UH'alpha) A^-1,   d/d ell, d/d s2, d/dB given
//   chain rule: softplus' for ell, s2, the diagonals; (G + G') W for the covar factors (autograd of W W').
// torch.optim.Adam (no weight decay, no amsgrad), step = 1-based count of this update:
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  theta -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
// skip[b] != 0 (a factorisation the caller could not repair): the model is left as it is, loss[b] = NaN.
template <typename T>
__global__ void __launch_bounds__(64)
fit_adam_step_kernel(T* __restrict__ theta, T* __restrict__ mom1, T* __restrict__ mom2, const T* __restrict__ g_ell,
                     const T* __restrict__ g_s2, const T* __restrict__ g_B, const T* __restrict__ logdetK,
                     const T* __restrict__ RtA, const T* __restrict__ UHtA, const T* __restrict__ Ainv,
                     const T* __restrict__ logdetA, const int* __restrict__ skip, T* __restrict__ loss, T* __restrict__ grad_out,
                     int Bt, int N, FitLayout lay, int step, double lr, double beta1, double beta2, double eps, int has_prior,
                     double prior_c, double prior_r) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= Bt) return;
    const int n = lay.n, C = lay.C, rA = lay.rA, rB = lay.rB, P = lay.P();
    if (skip != nullptr && skip[b] != 0) {
        if (loss != nullptr) loss[b] = (T)NAN;
        return;
    }
    T* th = theta + (size_t)b * P;
    const double scale = 1.0 / ((double)N * n);
    double Ai[FN][FN], Rt[FN][FN], GA[FN][FN], tmp[FN][FN];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            Ai[i][j] = (double)Ainv[((size_t)b * n + i) * n + j];
            Rt[i][j] = (double)RtA[((size_t)b * n + i) * n + j];
        }
    double tr = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) tr += Ai[i][j] * Rt[j][i];
    double nll = 0.5 * tr + 0.5 * n * (double)logdetK[b] + 0.5 * N * (double)logdetA[b] + 0.5 * N * n * 1.8378770664093453;
    // GA = d loss / dA = -scale (1/2 Ai Rt Ai - N/2 Ai)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += Ai[i][k] * Rt[k][j];
            tmp[i][j] = s;
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += tmp[i][k] * Ai[k][j];
            GA[i][j] = -scale * (0.5 * s - 0.5 * N * Ai[i][j]);
        }
    double g[FN + 1 + FN * FN + FN + FC * FC + FC + FC * FN];
    double lp = 0.0;
    for (int d = 0; d < n; ++d) {
        const double raw = (double)th[lay.o_ell() + d];
        double gl = -scale * (double)g_ell[(size_t)b * n + d];
        if (has_prior) {
            const double l = softplus_d(raw);
            lp += prior_c * log(prior_r) - lgamma(prior_c) + (prior_c - 1.0) * log(l) - prior_r * l;
            gl += -scale * ((prior_c - 1.0) / l - prior_r);
        }
        g[lay.o_ell() + d] = gl * dsoftplus_d(raw);
    }
    g[lay.o_s2()] = -scale * (double)g_s2[b] * dsoftplus_d((double)th[lay.o_s2()]);
    for (int i = 0; i < n; ++i) {
        for (int q = 0; q < rA; ++q) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += (GA[i][k] + GA[k][i]) * (double)th[lay.o_Wa() + k * rA + q];
            g[lay.o_Wa() + i * rA + q] = s;
        }
        g[lay.o_va() + i] = GA[i][i] * dsoftplus_d((double)th[lay.o_va() + i]);
    }
    for (int i = 0; i < C; ++i) {
        for (int q = 0; q < rB; ++q) {
            double s = 0.0;
            for (int k = 0; k < C; ++k)
                s += -scale * ((double)g_B[((size_t)b * C + i) * C + k] + (double)g_B[((size_t)b * C + k) * C + i]) * (double)th[lay.o_Wb() + k * rB + q];
            g[lay.o_Wb() + i * rB + q] = s;
        }
        g[lay.o_vb() + i] = -scale * (double)g_B[((size_t)b * C + i) * C + i] * dsoftplus_d((double)th[lay.o_vb() + i]);
    }
    for (int c = 0; c < C; ++c)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += (double)UHtA[((size_t)b * C + c) * n + k] * Ai[k][j];
            g[lay.o_M0() + c * n + j] = -scale * s;
        }
    if (loss != nullptr) loss[b] = (T)(nll * scale - lp * scale);
    if (grad_out != nullptr)
        for (int p = 0; p < P; ++p) grad_out[(size_t)b * P + p] = (T)g[p];
    if (step <= 0) return;                                 // (value and gradient only)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2s = sqrt(1.0 - pow(beta2, (double)step));
    const double step_size = lr / bc1;
    T* m1 = mom1 + (size_t)b * P;
    T* m2 = mom2 + (size_t)b * P;
    for (int p = 0; p < P; ++p) {
        const double gp = (double)(T)g[p];                 // (the gradient in the parameters' precision, as torch holds it)
        double m = (double)m1[p], v = (double)m2[p];
        m = m + (gp - m) * (1.0 - beta1);
        v = v * beta2 + (1.0 - beta2) * gp * gp;
        m1[p] = (T)m;
        m2[p] = (T)v;
        const double denom = sqrt((double)(T)v) / bc2s + eps;
        th[p] = (T)((double)th[p] - step_size * ((double)(T)m / denom));
    }
}

// alpha[b][i][c] = sum_j Kinv[b][j][i] R[b][j][c]  (Kinv symmetric: column i read as ROW elements Kinv[j][i], consecutive
// lanes = consecutive i -> every load is a coalesced row segment and no lane reduces across the wave).  256 threads = 64
// columns x 4 slices of j, added through LDS in a fixed order.  HBM-bound: N^2 elements per model read once.
constexpr int KA_NT = BCBF_MAX_STATE_DIM;
template <typename T>
__global__ void __launch_bounds__(256)
kinv_apply_kernel(const T* __restrict__ Kinv, const T* __restrict__ R, T* __restrict__ alpha, int N, int nt) {
    __shared__ double part[4][64][KA_NT];
    const int b = blockIdx.y, col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const T* K = Kinv + (size_t)b * N * N;
    const T* Rb = R + (size_t)b * N * nt;
    double acc[KA_NT];
#pragma unroll
    for (int c = 0; c < KA_NT; ++c) acc[c] = 0.0;
    const int per = (N + 3) / 4, j0 = sl * per, j1 = min(N, j0 + per);
    if (i < N)
        for (int j = j0; j < j1; ++j) {
            const double k = (double)K[(size_t)j * N + i];
#pragma unroll
            for (int c = 0; c < KA_NT; ++c)
                if (c < nt) acc[c] += k * (double)Rb[(size_t)j * nt + c];
        }
#pragma unroll
    for (int c = 0; c < KA_NT; ++c) part[sl][col][c] = acc[c];
    __syncthreads();
    if (sl == 0 && i < N)
        for (int c = 0; c < nt; ++c)
            alpha[((size_t)b * N + i) * nt + c] = (T)(part[0][col][c] + part[1][col][c] + part[2][col][c] + part[3][col][c]);
}

// ... the same product with 16 bytes per lane and eight rows of K_b^-1 in flight per lane (batches; N a multiple of the vector width, nt <= 4): the form above
// issues one 4-byte load per multiply-add group and waits for it (4096 x 512: 2.6 ms fp32 / 3.0 ms fp64 for a 4.3 / 8.6 GB read).  A lane owns V consecutive
// columns, a wave one of four row slices (its rows of R staged in LDS as doubles), sums in fp64 as above.
#ifndef BCBF_KA_OCC
#define BCBF_KA_OCC 2          // (4096 x 512, ms fp32 / fp64 at 2 / 3 / 4 waves per SIMD: 0.735 / 0.89 / 1.96 (spills); 1.46 throughout)
#endif
#ifndef BCBF_KA_U
#define BCBF_KA_U 8            // rows of K_b^-1 in flight per lane (4 / 8 / 16 / 32: the same 1.43 - 1.47 ms fp64, 0.735 - 0.80 fp32 at 4096 x 512: 5.9 TB/s
                               // is what 1 KB pieces at a row's stride give; whole rows per workgroup would be the next form)
#endif
template <typename T, int NT>
__global__ void __launch_bounds__(256, BCBF_KA_OCC)
kinv_apply_vec_kernel(const T* __restrict__ Kinv, const T* __restrict__ R, T* __restrict__ alpha, int N, int nt) {
    constexpr int V = 16 / (int)sizeof(T), U = BCBF_KA_U;
    using VecT = typename Vec<T>::type;
    // dynamic LDS: the four slices' rows of R as doubles ([4][per][NT]) while the sums run, the four partial sums ([4][64][V][NT])
    // afterwards -- one buffer for both (64 KB of static arrays held a CU to two workgroups)
    extern __shared__ __attribute__((aligned(16))) double ka_smem[];
    const int b = blockIdx.y, lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 64 + lane) * V;
    const T* K = Kinv + (size_t)b * N * N;
    const T* Rb = R + (size_t)b * N * nt;
    const int per = (N + 3) / 4, j0 = sl * per, j1 = min(N, j0 + per), cnt = j1 - j0;
    double (*rs)[NT] = reinterpret_cast<double (*)[NT]>(ka_smem + (size_t)sl * per * NT);      // this wave's slice [per][NT]
    double (*part)[64][V][NT] = reinterpret_cast<double (*)[64][V][NT]>(ka_smem);
    for (int e = lane; e < per * NT; e += 64) {
        const int jj = e / NT, c = e - jj * NT;
        rs[jj][c] = (jj < cnt && c < nt) ? (double)Rb[(size_t)(j0 + jj) * nt + c] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();                              // (a wave reads its own slice only)
    double acc[V][NT];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[v][c] = 0.0;
    if (i0 < N)
        for (int jj = 0; jj < cnt; jj += U) {
            VecT k[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + min(jj + u, cnt - 1);                 // (a row past the slice repeats the last one and meets zeros below)
                k[u] = *reinterpret_cast<const VecT*>(K + (size_t)j * N + i0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = jj + u < cnt;
                const T* kv = reinterpret_cast<const T*>(&k[u]);
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const double r = in ? rs[min(jj + u, per - 1)][c] : 0.0;
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[v][c] += (double)kv[v] * r;
                }
            }
        }
    __syncthreads();                                              // (every slice has been read: the buffer changes hands)
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int c = 0; c < NT; ++c) part[sl][lane][v][c] = acc[v][c];
    __syncthreads();
    if (sl == 0 && i0 < N)
#pragma unroll
        for (int v = 0; v < V; ++v)
            for (int c = 0; c < nt; ++c)
                alpha[((size_t)b * N + i0 + v) * nt + c] = (T)(part[0][lane][v][c] + part[1][lane][v][c] + part[2][lane][v][c] + part[3][lane][v][c]);
}

static bool fit_shape_ok(int Bt, int n, int m, int rA, int rB) {
    return Bt >= 0 && n >= 1 && n <= BCBF_MAX_STATE_DIM && m >= 1 && m <= BCBF_MAX_CTRL_DIM && rA >= 0 && rA <= n && rB >= 0 && rB <= m + 1;
}

template <typename T>
static int launch_fit_derive(const T* theta, T* ell, T* s2, T* A, T* Bm, T* M0, T* Ainv, T* logdetA, int Bt, int n, int m, int rA,
                             int rB, void* stream) {
    if (!fit_shape_ok(Bt, n, m, rA, rB)) return BCBF_EINVAL;
    if (Bt == 0) return BCBF_OK;
    if (!theta || !ell || !s2 || !A || !Bm || !M0 || (logdetA && !Ainv)) return BCBF_EINVAL;
    const FitLayout lay{n, m + 1, rA, rB};
    hipLaunchKernelGGL((fit_derive_kernel<T>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, theta, ell, s2, A, Bm, M0, Ainv,
                       logdetA, Bt, lay);
    return check_launch("bcbf_fit_derive");
}

template <typename T>
static int launch_fit_adam_step(T* theta, T* mom1, T* mom2, const T* g_ell, const T* g_s2, const T* g_B, const T* logdetK,
                                const T* RtA, const T* UHtA, const T* Ainv, const T* logdetA, const int* skip, T* loss, T* grad_out,
                                int Bt, int N, int n, int m, int rA, int rB, int step, double lr, double beta1, double beta2,
                                double eps, const double* gamma_prior, void* stream) {
    if (!fit_shape_ok(Bt, n, m, rA, rB) || N < 1) return BCBF_EINVAL;
    if (Bt == 0) return BCBF_OK;
    if (!theta || !g_ell || !g_s2 || !g_B || !logdetK || !RtA || !UHtA || !Ainv || !logdetA) return BCBF_EINVAL;
    if (step > 0 && (!mom1 || !mom2 || !(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)))
        return BCBF_EINVAL;
    if (gamma_prior && !(gamma_prior[0] > 0.0 && gamma_prior[1] > 0.0)) return BCBF_EINVAL;
    const FitLayout lay{n, m + 1, rA, rB};
    hipLaunchKernelGGL((fit_adam_step_kernel<T>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, theta, mom1, mom2, g_ell, g_s2,
                       g_B, logdetK, RtA, UHtA, Ainv, logdetA, skip, loss, grad_out, Bt, N, lay, step, lr, beta1, beta2, eps,
                       gamma_prior ? 1 : 0, gamma_prior ? gamma_prior[0] : 0.0, gamma_prior ? gamma_prior[1] : 0.0);
    return check_launch("bcbf_fit_adam_step");
}

template <typename T>
static int launch_kinv_apply(const T* Kinv, const T* R, T* alpha, int Bt, int N, int nt, void* stream) {
    if (Bt < 0 || N < 1 || nt < 1 || nt > KA_NT) return BCBF_EINVAL;
    if (Bt == 0) return BCBF_OK;
    if (!Kinv || !R || !alpha || (const void*)R == (const void*)alpha) return BCBF_EINVAL;
    constexpr int V = 16 / (int)sizeof(T);
    if (Bt >= 16 && N % V == 0 && nt <= 4 && (N + 3) / 4 <= 256) {      // batches: 16 bytes per lane, eight rows in flight
        const size_t rs_b = (size_t)4 * ((N + 3) / 4) * 4 * sizeof(double), part_b = (size_t)4 * 64 * V * 4 * sizeof(double);
        hipLaunchKernelGGL((kinv_apply_vec_kernel<T, 4>), dim3((N / V + 63) / 64, Bt), dim3(256), rs_b > part_b ? rs_b : part_b,
                           (hipStream_t)stream, Kinv, R, alpha, N, nt);
        return check_launch("bcbf_kinv_apply");
    }
    hipLaunchKernelGGL((kinv_apply_kernel<T>), dim3((N + 63) / 64, Bt), dim3(256), 0, (hipStream_t)stream, Kinv, R, alpha, N, nt);
    return check_launch("bcbf_kinv_apply");
}

}  // namespace bcbf

extern "C" {
int bcbf_fit_param_count(int n, int m, int rA, int rB) {
    if (!bcbf::fit_shape_ok(0, n, m, rA, rB)) return BCBF_EINVAL;
    return bcbf::FitLayout{n, m + 1, rA, rB}.P();
}
int bcbf_fit_derive_f32(const float* theta, float* ell, float* s2, float* A, float* Bm, float* M0, float* Ainv, float* logdetA,
                        int Bt, int n, int m, int rA, int rB, void* stream) {
    return bcbf::launch_fit_derive<float>(theta, ell, s2, A, Bm, M0, Ainv, logdetA, Bt, n, m, rA, rB, stream);
}
int bcbf_fit_derive_f64(const double* theta, double* ell, double* s2, double* A, double* Bm, double* M0, double* Ainv,
                        double* logdetA, int Bt, int n, int m, int rA, int rB, void* stream) {
    return bcbf::launch_fit_derive<double>(theta, ell, s2, A, Bm, M0, Ainv, logdetA, Bt, n, m, rA, rB, stream);
}
int bcbf_fit_adam_step_f32(float* theta, float* mom1, float* mom2, const float* g_ell, const float* g_s2, const float* g_B,
                           const float* logdetK, const float* RtA, const float* UHtA, const float* Ainv, const float* logdetA,
                           const int* skip, float* loss, float* grad_out, int Bt, int N, int n, int m, int rA, int rB, int step,
                           double lr, double beta1, double beta2, double eps, const double* gamma_prior, void* stream) {
    return bcbf::launch_fit_adam_step<float>(theta, mom1, mom2, g_ell, g_s2, g_B, logdetK, RtA, UHtA, Ainv, logdetA, skip, loss,
                                             grad_out, Bt, N, n, m, rA, rB, step, lr, beta1, beta2, eps, gamma_prior, stream);
}
int bcbf_fit_adam_step_f64(double* theta, double* mom1, double* mom2, const double* g_ell, const double* g_s2, const double* g_B,
                           const double* logdetK, const double* RtA, const double* UHtA, const double* Ainv, const double* logdetA,
                           const int* skip, double* loss, double* grad_out, int Bt, int N, int n, int m, int rA, int rB, int step,
                           double lr, double beta1, double beta2, double eps, const double* gamma_prior, void* stream) {
    return bcbf::launch_fit_adam_step<double>(theta, mom1, mom2, g_ell, g_s2, g_B, logdetK, RtA, UHtA, Ainv, logdetA, skip, loss,
                                              grad_out, Bt, N, n, m, rA, rB, step, lr, beta1, beta2, eps, gamma_prior, stream);
}
int bcbf_kinv_apply_f32(const float* Kinv, const float* R, float* alpha, int Bt, int N, int nt, void* stream) {
    return bcbf::launch_kinv_apply<float>(Kinv, R, alpha, Bt, N, nt, stream);
}
int bcbf_kinv_apply_f64(const double* Kinv, const double* R, double* alpha, int Bt, int N, int nt, void* stream) {
    return bcbf::launch_kinv_apply<double>(Kinv, R, alpha, Bt, N, nt, stream);
}
}
