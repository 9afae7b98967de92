// Shared device/host helpers for libbcbf (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "bcbf.h"

namespace bcbf {

constexpr int NB = BCBF_NB;  // diagonal block of the packed operator

template <typename T> struct Vec;
template <> struct Vec<float> { using type = float4; static constexpr int V = 4; };
template <> struct Vec<double> { using type = double2; static constexpr int V = 2; };

__host__ __device__ inline int round_up(int x, int q) { return (x + q - 1) / q * q; }

// Packed operator (bcbf.h "Lop"), Np = N padded to 32, nblk = Np / 32:
//   * off-diagonal part first: column j of block column J = j / 32 stores rows 32 (J+1) .. Np-1 (everything BELOW its
//     32x32 diagonal block), columns back to back.  Every column starts on a 128-byte boundary and is a whole number of
//     128-byte lines, so a wave's 16-byte-per-lane loads never straddle a line shared with another column.
//     Element (i, j), i >= 32 (J+1), lives at lop_base(j) + i.
//   * then the INVERTED diagonal blocks, lower triangles only, column-major packed, LOP_DB = 544 elements per block
//     (528 used: a block starts on a 128-byte line): inv(L_JJ)[r][c], r >= c, lives at lop_dinv(J, r, c).
//     This is what the per-instance streaming kernels read: Np (Np + 2) / 2 elements, the exact triangle plus half a
//     row of padding per block.
//   * last, a second copy of the inverted diagonal blocks as full 32x32 column-major tiles (zeros above the diagonal):
//     inv(L_JJ)[r][c] at lop_dfull(J, r, c).  Only the shared-model matrix-core kernel reads it (its factor is cache
//     resident, bytes do not matter there, but an MFMA operand tile wants the same affine addressing as the
//     off-diagonal tiles); the streaming kernels never touch it, so it costs capacity (+1024 per block), not traffic.
// (The template parameter V is the vector width of the caller; the layout itself does not depend on it.)
constexpr int LOP_DB = 544;
template <int V>
__host__ __device__ inline int lop_base(int j, int Np) {
    const int J = j / NB, c = j - J * NB;
    return NB * J * Np - 512 * J * (J + 1) + c * (Np - NB * (J + 1)) - NB * (J + 1);
}
__host__ __device__ inline int lop_offd_elems(int Np) { return Np * Np / 2 - 16 * Np; }
__host__ __device__ inline int lop_dinv_col(int c) { return NB * c - c * (c - 1) / 2 - c; }   // start of column c, minus c
// offset of inv(L_JJ)[r][c] (r >= c):  lop_dinv_block(J) + lop_dinv_col(c) + r
__host__ __device__ inline int lop_dinv_block(int J, int Np) { return lop_offd_elems(Np) + LOP_DB * J; }
__host__ __device__ inline int lop_dinv(int J, int r, int c, int Np) { return lop_dinv_block(J, Np) + lop_dinv_col(c) + r; }
__host__ __device__ inline int lop_dfull_block(int J, int Np) { return Np * (Np + 2) / 2 + NB * NB * J; }
__host__ __device__ inline int lop_dfull(int J, int r, int c, int Np) { return lop_dfull_block(J, Np) + NB * c + r; }
template <int V>
__host__ __device__ inline size_t lop_elems(int Np) { return (size_t)Np * (Np + 2) / 2 + (size_t)NB * Np; }

void set_error(const char* what, hipError_t err);
// bcbf_refit_retry: the previous attempt's info[Bt] while the retry's launch is being set up on this host thread (NULL otherwise).
// Every refit kernel takes it as its last argument: an instance whose entry is 0 returns at once (info[b] = 0 again).
extern thread_local const int* g_refit_only_bad;
int check_launch(const char* what);
bool posterior_shared_fits(int N, int n, int m);   // regime-S MFMA kernel: staging + one W slab fit in LDS
bool posterior_shared64_fits(int N, int n, int m); // solution in registers (posterior_shared_reg.hip), fp64: N <= 512, n <= 4
bool posterior_shared_reg32_fits(int N, int n, int m);   // the same form in fp32
template <typename T>
int launch_posterior_shared_reg(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2,
                                const T* Bm, const T* M0, const T* xq, const T* jitter2, T* Mk, T* Bk, T* W, int nq,
                                int N, int n, int m, void* stream, int kind = 0);   // kind: 0 RBF, 1 Matern-5/2

// Data kernels: k(x, x') = s2 * shape(d2), d2 = sum_d ((x_d - x'_d) / ell_d)^2.
//   0  RBF               exp(-d2 / 2)                                   -- the reference's (ScaleKernel(RBFKernel(ard)))
//   1  Matern-5/2        (1 + a + a^2 / 3) exp(-a),  a = sqrt(5 d2)      -- opt-in
//   2  RBF x Matern-5/2  the PRODUCT of the two, one set of ARD length scales ("RBF x Matern", BASELINE.json north_star) -- opt-in
// The opt-in kinds have no reference counterpart (no Matern in the reference); parity unpinned, formulas held to
// the CPU oracle and to finite differences.  `dshape` = -2 d shape / d(d2): d shape / d x_d = dshape (x'_d - x_d) / ell_d^2,
// d shape / d ell_d = dshape z_d^2 / ell_d; at x' = x: d2 shape / dx_d dx'_d = kernel_kxx(kind) / ell_d^2.
constexpr int BCBF_KIND_RBF = 0, BCBF_KIND_MATERN52 = 1, BCBF_KIND_RBF_MATERN52 = 2, BCBF_KINDS = 3;
template <typename T, typename EXPF>
__host__ __device__ inline void kernel_shape(int kind, T d2, EXPF expf_, T& shape, T& dshape) {
    if (kind == BCBF_KIND_RBF) { shape = expf_(T(-0.5) * d2); dshape = shape; return; }
    const T a5 = (T)sqrt((double)(T(5) * d2));
    const T poly = T(1) + a5 + T(5) / T(3) * d2, dpoly = T(5) / T(3) * (T(1) + a5);
    if (kind == BCBF_KIND_MATERN52) { const T e5 = expf_(-a5); shape = poly * e5; dshape = dpoly * e5; return; }
    const T e = expf_(-a5 - T(0.5) * d2);
    shape = poly * e;
    dshape = shape + dpoly * e;
}
__host__ __device__ inline double kernel_kxx(int kind) { return kind == BCBF_KIND_RBF ? 1.0 : kind == BCBF_KIND_MATERN52 ? 5.0 / 3.0 : 8.0 / 3.0; }

// exp(-x) for x >= 0 in full double precision: k = rint(x log2 e), r = k ln2 - x in [-ln2/2, ln2/2] (two-part ln2),
// degree-12 Taylor polynomial (truncation 1.7e-16), scaled by 2^-k.  Half the instructions of the library exp (no
// special cases: the argument is a squared distance).
__device__ inline double exp_neg64(double x) {
    const double kf = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(kf, 0.6931471803691238, -x);
    r = __builtin_fma(kf, 1.9082149292705877e-10, r);
    double p = 1.0 / 479001600.0;
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, -(int)kf);
}
template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace bcbf
