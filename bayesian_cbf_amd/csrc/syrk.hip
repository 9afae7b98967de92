// K_b^-1 = Linv' Linv from the dense triangular inverse Linv = L^-1 (bcbf_trtri) -- the fit path's K_b^-1
// (ControlAffineRegressor.fit, control_affine_model.py:268-335: gpytorch forms inv-quad / logdet terms by its own solves).
// One wave per 32 x 32 output tile (I, J), J <= I:  C[i][j] = sum_{k >= 32 I} Linv[k][32 I + i] Linv[k][32 J + j]
// (rows above tile row I are zero in column block I: Linv is lower triangular).  Both MFMA operands are read straight from
// global memory with the tile column on the lane -- 32 consecutive elements of one row of Linv, one 128-byte (fp32) or
// 256-byte (fp64) segment per half wave, no LDS, no transposes; the factor is cache resident (1 MiB at N = 512 in fp32).
// The mirror tile (J, I) is written from the same accumulators.
#include "bcbf_common.h"

namespace bcbf {

constexpr int SY_DEPTH = 8;        // k-steps whose operand loads are issued together

__global__ void __launch_bounds__(64)
syrk_lt_kernel_f32(const float* __restrict__ Linv, float* __restrict__ Kinv, int N, int nb) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    // tile index t -> (I, J), J <= I
    int t = blockIdx.x, I = 0;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const float* A = Linv + (size_t)blockIdx.y * N * N;
    float* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 31, kk = lane >> 5;          // v_mfma_f32_32x32x2: lane = (column, k of 2)
    const int ci = I * 32 + col, cj = J * 32 + col;
    const bool vi = ci < N, vj = cj < N;
    f32x16 acc = {0};
    for (int k0 = I * 32; k0 < N; k0 += 2 * SY_DEPTH) {
        float a[SY_DEPTH], b[SY_DEPTH];
#pragma unroll
        for (int s = 0; s < SY_DEPTH; ++s) {
            const int k = k0 + 2 * s + kk;
            a[s] = (vi && k < N) ? A[(size_t)k * N + ci] : 0.0f;
            b[s] = (vj && k < N) ? A[(size_t)k * N + cj] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < SY_DEPTH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    // accumulator: lane (column j = lane % 32, group g = lane / 32), register r -> row i = 8 (r / 4) + 4 g + r % 4
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r >> 2) + 4 * kk + (r & 3);
        const int gi = I * 32 + i;
        if (gi < N && vj) {
            C[(size_t)gi * N + cj] = acc[r];
            if (I != J) C[(size_t)cj * N + gi] = acc[r];
        }
    }
}

__global__ void __launch_bounds__(64)
syrk_lt_kernel_f64(const double* __restrict__ Linv, double* __restrict__ Kinv, int N, int nb) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    int t = blockIdx.x, I = 0;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const double* A = Linv + (size_t)blockIdx.y * N * N;
    double* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 15, kk = lane >> 4;          // v_mfma_f64_16x16x4: lane = (column, k of 4)
    f64x4 acc[2][2] = {};
    for (int k0 = I * 32; k0 < N; k0 += 4 * (SY_DEPTH / 2)) {
        double a[SY_DEPTH / 2][2], b[SY_DEPTH / 2][2];
#pragma unroll
        for (int s = 0; s < SY_DEPTH / 2; ++s) {
            const int k = k0 + 4 * s + kk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ci = I * 32 + 16 * h + col, cj = J * 32 + 16 * h + col;
                a[s][h] = (ci < N && k < N) ? A[(size_t)k * N + ci] : 0.0;
                b[s][h] = (cj < N && k < N) ? A[(size_t)k * N + cj] : 0.0;
            }
        }
#pragma unroll
        for (int s = 0; s < SY_DEPTH / 2; ++s)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int hj = 0; hj < 2; ++hj)
                    acc[hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][hi], b[s][hj], acc[hi][hj], 0, 0, 0);
    }
    // accumulator of one 16 x 16 tile: lane (column j = lane % 16, group g = lane / 16), register r -> row i = g + 4 r
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int hj = 0; hj < 2; ++hj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = I * 32 + 16 * hi + kk + 4 * r, gj = J * 32 + 16 * hj + col;
                if (gi < N && gj < N) {
                    C[(size_t)gi * N + gj] = acc[hi][hj][r];
                    if (I != J) C[(size_t)gj * N + gi] = acc[hi][hj][r];
                }
            }
}

// Batches (round 6): 2 x 2 output tiles per wave.  A wave that owns tiles (I, I+1) x (J, J+1) reads FOUR column blocks of Linv for FOUR tiles where the
// one-tile form reads two per tile -- half the operand loads, and eight (fp64: sixteen) independent accumulators per MFMA stream.  Super-tile (SI, SJ), SJ <= SI;
// the contraction starts at row 64 SI (the rows above are zero in column blocks 2 SI and 2 SI + 1: Linv is lower triangular and stored with its zeros).  In a
// diagonal super-tile the tile above the diagonal is the mirror of the one below it: not computed, written from the same accumulators like every mirror.
__global__ void __launch_bounds__(64)
syrk_lt2_kernel_f64(const double* __restrict__ Linv, double* __restrict__ Kinv, int N, int nb2) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    int t = blockIdx.x, SI = 0;
    while ((SI + 1) * (SI + 2) / 2 <= t) ++SI;
    const int SJ = t - SI * (SI + 1) / 2;
    const double* A = Linv + (size_t)blockIdx.y * N * N;
    double* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 15, kk = lane >> 4;
    f64x4 acc[2][2][2][2] = {};                                  // [ti][tj][hi][hj]
    constexpr int D = 4;                                         // k-steps (of 4 rows) whose operand loads are issued together
    for (int k0 = SI * 64; k0 < N; k0 += 4 * D) {
        double a[D][2][2], b[D][2][2];                           // [step][tile][half]
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const int k = k0 + 4 * q + kk;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ci = (2 * SI + ti) * 32 + 16 * h + col, cj = (2 * SJ + ti) * 32 + 16 * h + col;
                    a[q][ti][h] = (ci < N && k < N) ? A[(size_t)k * N + ci] : 0.0;
                    b[q][ti][h] = (cj < N && k < N) ? A[(size_t)k * N + cj] : 0.0;
                }
        }
#pragma unroll
        for (int q = 0; q < D; ++q)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    if (SI == SJ && tj > ti) continue;           // (the mirror of tile (1, 0))
#pragma unroll
                    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                        for (int hj = 0; hj < 2; ++hj)
                            acc[ti][tj][hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][ti][hi], b[q][tj][hj], acc[ti][tj][hi][hj], 0, 0, 0);
                }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            if (SI == SJ && tj > ti) continue;
            const int I = 2 * SI + ti, J = 2 * SJ + tj;
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int hj = 0; hj < 2; ++hj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gi = I * 32 + 16 * hi + kk + 4 * r, gj = J * 32 + 16 * hj + col;
                        if (gi < N && gj < N) {
                            C[(size_t)gi * N + gj] = acc[ti][tj][hi][hj][r];
                            if (I != J) C[(size_t)gj * N + gi] = acc[ti][tj][hi][hj][r];
                        }
                    }
        }
}

__global__ void __launch_bounds__(64)
syrk_lt2_kernel_f32(const float* __restrict__ Linv, float* __restrict__ Kinv, int N, int nb2) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    int t = blockIdx.x, SI = 0;
    while ((SI + 1) * (SI + 2) / 2 <= t) ++SI;
    const int SJ = t - SI * (SI + 1) / 2;
    const float* A = Linv + (size_t)blockIdx.y * N * N;
    float* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 31, kk = lane >> 5;
    f32x16 acc[2][2] = {};
    for (int k0 = SI * 64; k0 < N; k0 += 2 * 4) {
        float a[4][2], b[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + 2 * q + kk;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int ci = (2 * SI + ti) * 32 + col, cj = (2 * SJ + ti) * 32 + col;
                a[q][ti] = (ci < N && k < N) ? A[(size_t)k * N + ci] : 0.0f;
                b[q][ti] = (cj < N && k < N) ? A[(size_t)k * N + cj] : 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    if (SI == SJ && tj > ti) continue;
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][ti], b[q][tj], acc[ti][tj], 0, 0, 0);
                }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            if (SI == SJ && tj > ti) continue;
            const int I = 2 * SI + ti, J = 2 * SJ + tj;
            const int cj = J * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gi = I * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
                if (gi < N && cj < N) {
                    C[(size_t)gi * N + cj] = acc[ti][tj][r];
                    if (I != J) C[(size_t)cj * N + gi] = acc[ti][tj][r];
                }
            }
        }
}

}  // namespace bcbf

extern "C" int bcbf_syrk_lt_f32(const float* Linv, float* Kinv, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Linv || !Kinv || N < 1 || Linv == Kinv) return BCBF_EINVAL;
    const int nb = (N + 31) / 32;
    if (Bt >= 16 && nb >= 4) {                     // batches: 2 x 2 tiles per wave (a single model keeps the one-tile form: more waves)
        const int nb2 = (nb + 1) / 2;
        hipLaunchKernelGGL(bcbf::syrk_lt2_kernel_f32, dim3(nb2 * (nb2 + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb2);
        return bcbf::check_launch("syrk_lt");
    }
    hipLaunchKernelGGL(bcbf::syrk_lt_kernel_f32, dim3(nb * (nb + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb);
    return bcbf::check_launch("syrk_lt");
}
extern "C" int bcbf_syrk_lt_f64(const double* Linv, double* Kinv, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Linv || !Kinv || N < 1 || Linv == Kinv) return BCBF_EINVAL;
    const int nb = (N + 31) / 32;
    if (Bt >= 16 && nb >= 4) {
        const int nb2 = (nb + 1) / 2;
        hipLaunchKernelGGL(bcbf::syrk_lt2_kernel_f64, dim3(nb2 * (nb2 + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb2);
        return bcbf::check_launch("syrk_lt");
    }
    hipLaunchKernelGGL(bcbf::syrk_lt_kernel_f64, dim3(nb * (nb + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb);
    return bcbf::check_launch("syrk_lt");
}
