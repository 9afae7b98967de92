// Regime S in fp64: many queries against ONE GP on v_mfma_f64_16x16x4_f64 (the reference's unicycle module is fp64,
// unicycle_move_to_pose.py:50; custom_predict with b test points, control_affine_model.py:536, 1051-1091).
//
// Same blocked forward substitution as posterior_shared.hip (fp32), rebuilt around what the fp64 MFMA gives:
//   * one wave = 4 queries = 16 right-hand-side columns (4 per query, unused ones zero);
//   * the fp64 accumulator holds row g + 4r (lane group g = lane >> 4, register r) of a 16-row tile in column
//     j = lane & 15 -- and the B operand of the next MFMA wants B[k = g][n = j]: with k-step s = (tile u, register r)
//     covering block rows 16u + 4r + g, an accumulator register IS a B operand, with no interleaving of rows at all;
//   * so W = L^-1 Phi never leaves the registers: wreg[K][s] (8 doubles per 32-row block, 16 blocks = 256 VGPRs at
//     N = 512; one wave per SIMD has 512) is written by the diagonal step of block K and read as the B operand of every
//     later tile (I, K).  No LDS slab (it would be 64 KB per wave in fp64: one wave per CU), no LDS traffic in the loop;
//     the loops over blocks are fully unrolled so that every wreg index is a compile-time constant;
//   * A operands: one 8-byte buffer load per lane per (output tile, k-step) straight from the packed operator (16
//     consecutive rows of one column per lane group: 128-byte segments), the whole next tile in flight while the
//     current one multiplies;  diagonal step: A = the stored inverse (full-tile copy), B = Phi - acc.
// X, UH*B and Vw are staged in LDS once per workgroup (doubles).  N <= 32 * PS64_MAXBLK; beyond that, and for few queries,
// the streaming kernel (posterior_step.hip, two queries per workgroup) answers.
#include "bcbf_common.h"

namespace bcbf {

using f64x4s = __attribute__((__vector_size__(4 * sizeof(double)))) double;
using u32x2q = __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned;

constexpr int PS64_MAXBLK = 16;          // N <= 512

// compile-time loop: the body sees a constant index (every wreg[][] subscript must be one, or the array leaves the
// register file for scratch memory -- "#pragma unroll" alone is a request the optimizer declines for 2000-MFMA bodies)
template <int I> struct Ic { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(Ic<B>{}); static_for<B + 1, E>(f); }
}

// quad broadcast of a double: lane c of every quad -> all four lanes (two DPP moves)
template <int CTRL> __device__ inline double dpp_qd(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

template <int C, int NS>
__global__ void __launch_bounds__(256, 1)
posterior_shared64_kernel(const double* __restrict__ Lop, const double* __restrict__ Vw, const double* __restrict__ X,
                          const double* __restrict__ UHB, const double* __restrict__ ell, const double* __restrict__ s2p,
                          const double* __restrict__ Bm, const double* __restrict__ M0, const double* __restrict__ xq,
                          const double* __restrict__ jitter2, double* __restrict__ Mk, double* __restrict__ Bk,
                          double* __restrict__ Wout, int nq, int N, int Np, int n) {
    constexpr int V = 2, QW = 4;
    extern __shared__ double smem64[];
    const int nwave = blockDim.x >> 6, wave = threadIdx.x >> 6;
    double* Xs = smem64;                               // [Np][NS]  (state dim padded to NS with zeros)
    double* Us = Xs + (size_t)Np * NS;                 // [Np][C]   (rows >= N are zero: padded rows contribute nothing)
    double* Vs = Us + (size_t)Np * C;                  // [Np][NS]
    for (int i = threadIdx.x; i < Np * NS; i += blockDim.x) {
        const int row = i / NS, d = i - row * NS;
        const bool ok = row < N && d < n;
        Xs[i] = ok ? X[(size_t)row * n + d] : 0.0;
        Vs[i] = ok ? Vw[(size_t)row * n + d] : 0.0;
    }
    for (int i = threadIdx.x; i < Np * C; i += blockDim.x) Us[i] = i < N * C ? UHB[i] : 0.0;
    __syncthreads();

    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int ql = j >> 2, c = j & 3;                  // query slot in the wave, component
    const int wq0 = (blockIdx.x * nwave + wave) * QW;
    if (wq0 >= nq) return;                             // whole wave idle (no further barriers below)
    const int q = wq0 + ql;
    const bool qok = q < nq, cok = c < C;
    const int qq = qok ? q : nq - 1;

    double xqr[NS], iell[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        xqr[d] = d < n ? xq[(size_t)qq * n + d] : 0.0;
        iell[d] = d < n ? 1.0 / ell[d] : 0.0;
    }
    const double s2 = s2p[0];
    double gram[C], mk[NS];
#pragma unroll
    for (int a = 0; a < C; ++a) gram[a] = 0.0;
#pragma unroll
    for (int d = 0; d < NS; ++d) mk[d] = 0.0;
    const int cc = cok ? c : 0;
    const double cmask = cok ? 1.0 : 0.0;
    const int nblk = Np / NB;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(Lop), 0, (int)(lop_elems<V>(Np) * sizeof(double)), 0x00020000);
    // A operands of tile (I, K), 16 values per lane: a[u'][s] = L[32I + 16u' + j][32K + 4s + g]  (K < I: the packed
    // off-diagonal part, element (row, col) at lop_base(col) + row; K == I: the full-tile copy of inv(L_II), column
    // stride 32).  One per-lane offset, scalar offsets per (u', s).
    auto load_tile = [&](double (&a)[2][8], int I, int K) {
        const bool isdiag = K == I;
        const int stride = isdiag ? NB : Np - NB * (K + 1);                   // column stride inside block column K
        const int base = isdiag ? lop_dfull_block(I, Np) : lop_base<V>(K * NB, Np) + I * NB;   // element (row 0 of the tile, col 0)
        const int voff = (g * stride + j) * (int)sizeof(double);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int up = 0; up < 2; ++up) {
                const int soff = (base + 4 * s * stride + 16 * up) * (int)sizeof(double);
                const u32x2q v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
                a[up][s] = __builtin_bit_cast(double, v);
            }
    };

    double wreg[PS64_MAXBLK][8];                       // W_K: rows 16u + 4r + g of block K at index 4u + r, column j
    double acur[2][8], anxt[2][8];
    load_tile(acur, 0, 0);
    static_for<0, PS64_MAXBLK>([&](auto Ict) {
        constexpr int I = decltype(Ict)::value;
        if (I < nblk) {
            f64x4s acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            // ---- off-diagonal tiles (I, K), K < I: acc += L_IK W_K   (acur holds tile (I, 0) when I > 0, else the diagonal)
            static_for<0, I>([&](auto Kct) {
                constexpr int K = decltype(Kct)::value;
                load_tile(anxt, I, K + 1);             // K + 1 == I: the diagonal tile (inverse)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[0][s], wreg[K][s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[1][s], wreg[K][s], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) { acur[0][s] = anxt[0][s]; acur[1][s] = anxt[1][s]; }
            });
            // ---- next tile in flight: (I + 1, 0)  (past the last block the bounds check of the buffer returns zeros)
            load_tile(anxt, I + 1, 0);
            // ---- Phi tile of block I: phi[4u + r] = k(x_q, X_row) (UH B)[row][c], row = 32I + 16u + 4r + g.  The exp of
            //      (query, row) is evaluated once, by the lane whose component equals r, and broadcast inside the quad
            double phi[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row_m = I * NB + 16 * u + 4 * c + g;
                double d2 = 0.0;
#pragma unroll
                for (int d = 0; d < NS; ++d) { const double z = (Xs[row_m * NS + d] - xqr[d]) * iell[d]; d2 += z * z; }
                const double kmine = s2 * exp(-0.5 * d2);
                const double kk[4] = {dpp_qd<0x00>(kmine), dpp_qd<0x55>(kmine), dpp_qd<0xAA>(kmine), dpp_qd<0xFF>(kmine)};
#pragma unroll
                for (int r = 0; r < 4; ++r) phi[4 * u + r] = kk[r] * Us[(I * NB + 16 * u + 4 * r + g) * C + cc] * cmask;
            }
            // ---- diagonal step: W_I = inv(L_II) (Phi_I - acc)   (inv(L_II) is lower triangular: tile 0 needs k < 16 only)
            f64x4s w0 = {0.0, 0.0, 0.0, 0.0}, w1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const double b_ = phi[s] - ((s >> 2) ? acc1[s & 3] : acc0[s & 3]);
                if (s < 4) w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[0][s], b_, w0, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[1][s], b_, w1, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double v = (e >> 2) ? w1[e & 3] : w0[e & 3];
                wreg[I][e] = v;
                const int row = I * NB + 16 * (e >> 2) + 4 * (e & 3) + g;
                if (Wout != nullptr && qok && cok) Wout[((size_t)q * Np + row) * C + c] = v;
                // Gram row / mean column of this lane's query
                const double vb[4] = {dpp_qd<0x00>(v), dpp_qd<0x55>(v), dpp_qd<0xAA>(v), dpp_qd<0xFF>(v)};
#pragma unroll
                for (int a_ = 0; a_ < C; ++a_) gram[a_] += v * vb[a_];             // lane c: G[c][a]
#pragma unroll
                for (int d = 0; d < NS; ++d) mk[d] += Vs[row * NS + d] * v;        // lane c: (Vw'W)[d][c]
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) { acur[0][s] = anxt[0][s]; acur[1][s] = anxt[1][s]; }
        }
    });

    // ---- the four lane groups hold different rows: combine, then write Mk[q][d][c], Bk[q][c][a]
#pragma unroll
    for (int a = 0; a < C; ++a) { gram[a] += __shfl_xor(gram[a], 16, 64); gram[a] += __shfl_xor(gram[a], 32, 64); }
#pragma unroll
    for (int d = 0; d < NS; ++d) { mk[d] += __shfl_xor(mk[d], 16, 64); mk[d] += __shfl_xor(mk[d], 32, 64); }
    if (g == 0 && qok && cok) {
#pragma unroll
        for (int d = 0; d < NS; ++d)
            if (d < n) Mk[((size_t)q * n + d) * C + c] = M0[c * n + d] + mk[d];
#pragma unroll
        for (int a = 0; a < C; ++a) {
            double v = s2 * Bm[c * C + a] - gram[a];
            if (a == c && jitter2 != nullptr) v += jitter2[(size_t)q * C + c];
            Bk[((size_t)q * C + c) * C + a] = v;
        }
    }
}

static int padded_state_dim64(int n) { return n <= 2 ? 2 : (n <= 4 ? n : 8); }   // instantiated widths

// N <= 512 and the staged copies fit in LDS
bool posterior_shared64_fits(int N, int n, int m) {
    const size_t Np = round_up(N, NB);
    return Np <= (size_t)NB * PS64_MAXBLK && Np * (2 * padded_state_dim64(n) + m + 1) * sizeof(double) <= 160 * 1024;
}

template <int C, int NS>
static void launch_shared64(dim3 grid, dim3 block, size_t lds, hipStream_t st, const double* Lop, const double* Vw,
                            const double* X, const double* UHB, const double* ell, const double* s2, const double* Bm,
                            const double* M0, const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                            int nq, int N, int Np, int n) {
    static int opt_in[64] = {0};               // largest dynamic LDS size opted into, per device
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    int& lds_opt_in = opt_in[dev_ & 63];
    if (lds > 64 * 1024 && (int)lds > lds_opt_in) {
        (void)hipFuncSetAttribute((const void*)posterior_shared64_kernel<C, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_opt_in = (int)lds;
    }
    hipLaunchKernelGGL((posterior_shared64_kernel<C, NS>), grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2,
                       Mk, Bk, W, nq, N, Np, n);
}

template <int C>
static void launch_shared64_c(int NSp, dim3 grid, dim3 block, size_t lds, hipStream_t st, const double* Lop,
                              const double* Vw, const double* X, const double* UHB, const double* ell, const double* s2,
                              const double* Bm, const double* M0, const double* xq, const double* jitter2, double* Mk,
                              double* Bk, double* W, int nq, int N, int Np, int n) {
#define BCBF_PSH64(NSV) launch_shared64<C, NSV>(grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n)
    switch (NSp) {
        case 2: BCBF_PSH64(2); break;
        case 3: BCBF_PSH64(3); break;
        case 4: BCBF_PSH64(4); break;
        default: BCBF_PSH64(8); break;
    }
#undef BCBF_PSH64
}

}  // namespace bcbf

extern "C" int bcbf_posterior_shared_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                         const double* ell, const double* s2, const double* Bm, const double* M0,
                                         const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                         int nq, int N, int n, int m, void* stream) {
    using namespace bcbf;
    if (nq <= 0) return BCBF_OK;
    if (!Lop || !Vw || !X || !UHB || !ell || !s2 || !Bm || !M0 || !xq || !Mk || !Bk) return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > 3) return BCBF_EINVAL;
    if (!posterior_shared64_fits(N, n, m)) return BCBF_EINVAL;
    const int Np = round_up(N, NB), C = m + 1, NSp = padded_state_dim64(n);
    const size_t lds = (size_t)Np * (2 * NSp + C) * sizeof(double);
    const int waves = (nq + 3) / 4;
    const int nwave = waves < 4 ? waves : 4;           // one wave per SIMD: the W tiles of a wave take 256 VGPRs
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((waves + nwave - 1) / nwave), block(64 * nwave);
    switch (m) {
        case 1: launch_shared64_c<2>(NSp, grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n); break;
        case 2: launch_shared64_c<3>(NSp, grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n); break;
        default: launch_shared64_c<4>(NSp, grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n); break;
    }
    return check_launch("posterior_shared64");
}
