// Dense inverse of the packed Cholesky factor on the matrix cores: Linv[Bt,N,N] = L^-1 (lower triangular) -- the first half of
// K_b^-1 = L^-T L^-1 of a likelihood-gradient evaluation (ControlAffineRegressor.fit, control_affine_model.py:268-335 of the
// reference: gpytorch obtains the inverse-quadratic and log-determinant terms from its own solves).  The solve-based form
// (solve.hip: thread per row, 8 identity columns per workgroup, every workgroup streams the whole factor) costs N / 8 passes over
// the factor per model -- 94 ms for 4096 models of 512 points in fp64, two thirds of a batched Adam iteration (round 6 profile).
//
// Blocked form, one wave per (model, block column J) of 32 columns:
//   X_JJ = inv(L_JJ)                                   (stored in the packed operator: its full-tile copy)
//   X_IJ = -inv(L_II) sum_{K = J}^{I-1} L_IK X_KJ      I = J+1 .. nblk-1, top down
// Both products are MFMA chains on 32 x 32 tiles.  First product: A = L_IK read straight from the packed operator (column-major
// block columns: for a fixed k the lanes read 32 consecutive rows), B = X_KJ read back from the dense output this wave wrote
// (row-major: 32 consecutive columns).  Second product: the accumulator of the first IS the B operand -- register q of a lane holds
// rows rho(q), rho(q) + 4 (fp32) resp. 4 q + g (fp64) of its column, so the contraction runs over k in that order and the A
// operand inv(L_II)[i][k] is gathered accordingly from the full-tile copy (column-major: consecutive rows for a fixed k).
// Flops N^3 / 3 per model; operator bytes ~ nblk^3 / 6 tiles per model, shared by the nblk waves of a model through L2 (they are
// launched next to one another, heaviest block column first).
#include "bcbf_common.h"

namespace bcbf {

// A wave reads back tiles it wrote itself: its stores must have left the wave before the loads are issued -- workgroup scope (one
// CU, one vector L1, write-through).  NOT __threadfence(): the agent-scope release writes the XCD's L2 back on every call, 15
// times per wave -- the first version of this kernel spent 90 % of its time there (55 ms instead of 94, not 8).
__device__ inline void trtri_own_writes_visible() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__global__ void __launch_bounds__(64)
trtri_mfma_kernel_f32(const float* __restrict__ Lop, float* __restrict__ Linv, int N, int Np, int nblk) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    constexpr int V = 4;
    const int b = blockIdx.x / nblk, J = blockIdx.x - b * nblk;
    const float* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    float* __restrict__ X = Linv + (size_t)b * N * N;
    const int lane = threadIdx.x, col = lane & 31, g = lane >> 5;
    const int gc = J * NB + col;                                   // this lane's output column
    const bool vc = gc < N;
    // rows above block row J of this block column: zeros (the contract: a full lower-triangular matrix)
    for (int i = g; i < J * NB; i += 2)
        if (vc && i < N) X[(size_t)i * N + gc] = 0.0f;
    // X_JJ = inv(L_JJ): full tile (zeros above the diagonal), element (r, c) at lop_dfull(J, r, c)
    {
        const int base = lop_dfull_block(J, Np);
        for (int r = g; r < NB; r += 2) {
            const int gi = J * NB + r;
            if (vc && gi < N) X[(size_t)gi * N + gc] = lop[base + NB * col + r];
        }
    }
    trtri_own_writes_visible();
    for (int I = J + 1; I < nblk; ++I) {
        f32x16 acc = {0};
        const int gr = I * NB + col;                               // this lane's A row (tile row I)
        // operands of block K + 1 are in flight while block K's MFMA chain runs (two register sets, swapped by the unrolled pair)
        float av[2][16], bv[2][16];
        auto load = [&](int K, float (&a)[16], float (&bq)[16]) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = 2 * s + g, gk = K * NB + k;          // column of L / row of X
                a[s] = lop[lop_base<V>(gk, Np) + gr];
                bq[s] = (vc && gk < N) ? X[(size_t)gk * N + gc] : 0.0f;
            }
        };
        load(J, av[0], bv[0]);
        for (int K = J; K < I; K += 2) {
            if (K + 1 < I) load(K + 1, av[1], bv[1]);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][s], bv[0][s], acc, 0, 0, 0);
            if (K + 1 < I) {
                if (K + 2 < I) load(K + 2, av[0], bv[0]);
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][s], bv[1][s], acc, 0, 0, 0);
            }
        }
        // X_IJ = -inv(L_II) S: register q of the accumulator = rows rho(q) + 4 g of S in this lane's column
        f32x16 out = {0};
        const int dbase = lop_dfull_block(I, Np);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = 8 * (q >> 2) + (q & 3) + 4 * g;
            const float a = lop[dbase + NB * k + col];            // inv(L_II)[row = col][k]
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[q], out, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = I * NB + 8 * (q >> 2) + 4 * g + (q & 3);
            if (vc && gi < N) X[(size_t)gi * N + gc] = -out[q];
        }
        trtri_own_writes_visible();                                // the next block row reads this tile back
    }
}

__global__ void __launch_bounds__(64)
trtri_mfma_kernel_f64(const double* __restrict__ Lop, double* __restrict__ Linv, int N, int Np, int nblk) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    constexpr int V = 2;
    const int b = blockIdx.x / nblk, J = blockIdx.x - b * nblk;
    const double* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    double* __restrict__ X = Linv + (size_t)b * N * N;
    const int lane = threadIdx.x, c16 = lane & 15, g = lane >> 4;     // v_mfma_f64_16x16x4: lane = (row / column, k of 4)
    {
        const int col = lane & 31, h = lane >> 5, gc = J * NB + col;
        for (int i = h; i < J * NB; i += 2)
            if (gc < N && i < N) X[(size_t)i * N + gc] = 0.0;
        const int base = lop_dfull_block(J, Np);
        for (int r = h; r < NB; r += 2) {
            const int gi = J * NB + r;
            if (gc < N && gi < N) X[(size_t)gi * N + gc] = lop[base + NB * col + r];
        }
    }
    trtri_own_writes_visible();
    for (int I = J + 1; I < nblk; ++I) {
        f64x4 acc[2][2] = {};
        // operands of block K + 1 are in flight while block K's MFMA chain runs (two register sets, as in the fp32 kernel)
        double av[2][8][2], bv[2][8][2];
        auto load = [&](int K, double (&a)[8][2], double (&bq)[8][2]) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {                          // k = 4 s + g within the 32-wide tile
                const int k = 4 * s + g, gk = K * NB + k;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    a[s][h] = lop[lop_base<V>(gk, Np) + I * NB + 16 * h + c16];
                    const int gc = J * NB + 16 * h + c16;
                    bq[s][h] = (gc < N && gk < N) ? X[(size_t)gk * N + gc] : 0.0;
                }
            }
        };
        auto chain = [&](double (&a)[8][2], double (&bq)[8][2]) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                    for (int hj = 0; hj < 2; ++hj)
                        acc[hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][hi], bq[s][hj], acc[hi][hj], 0, 0, 0);
        };
        load(J, av[0], bv[0]);
        for (int K = J; K < I; K += 2) {
            if (K + 1 < I) load(K + 1, av[1], bv[1]);
            chain(av[0], bv[0]);
            if (K + 1 < I) {
                if (K + 2 < I) load(K + 2, av[0], bv[0]);
                chain(av[1], bv[1]);
            }
        }
        // X_IJ = -inv(L_II) S: register q of acc[hk][hj] = row 16 hk + 4 q + g of S in column 16 hj + c16
        f64x4 out[2][2] = {};
        const int dbase = lop_dfull_block(I, Np);
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * hk + 4 * q + g;
                double a[2];
#pragma unroll
                for (int hi = 0; hi < 2; ++hi) a[hi] = lop[dbase + NB * k + 16 * hi + c16];      // inv(L_II)[16 hi + c16][k]
#pragma unroll
                for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                    for (int hj = 0; hj < 2; ++hj)
                        out[hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[hi], acc[hk][hj][q], out[hi][hj], 0, 0, 0);
            }
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
#pragma unroll
            for (int hj = 0; hj < 2; ++hj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gi = I * NB + 16 * hi + g + 4 * q, gc = J * NB + 16 * hj + c16;
                    if (gi < N && gc < N) X[(size_t)gi * N + gc] = -out[hi][hj][q];
                }
        trtri_own_writes_visible();
    }
}

int launch_trtri_mfma_f32(const float* Lop, float* Linv, int Bt, int N, void* stream) {
    const int Np = round_up(N, NB), nblk = Np / NB;
    if ((long long)Bt * nblk > 0x7fffffffLL) return BCBF_EINVAL;
    hipLaunchKernelGGL(trtri_mfma_kernel_f32, dim3(Bt * nblk), dim3(64), 0, (hipStream_t)stream, Lop, Linv, N, Np, nblk);
    return check_launch("trtri_mfma");
}
int launch_trtri_mfma_f64(const double* Lop, double* Linv, int Bt, int N, void* stream) {
    const int Np = round_up(N, NB), nblk = Np / NB;
    if ((long long)Bt * nblk > 0x7fffffffLL) return BCBF_EINVAL;
    hipLaunchKernelGGL(trtri_mfma_kernel_f64, dim3(Bt * nblk), dim3(64), 0, (hipStream_t)stream, Lop, Linv, N, Np, nblk);
    return check_launch("trtri_mfma");
}

}  // namespace bcbf
