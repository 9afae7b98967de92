// K1 + K2 on the matrix cores, fp64 (BASELINE configs[1]: batched kernel build + Cholesky, N = 256, fp64):
// the blocked left-looking Cholesky of refit_mfma.hip on v_mfma_f64_16x16x4_f64.
//
//   block column J, row tile I (32 rows).  The transposed 32x32 tile  S'[c][i] = K_b(i, 32J+c) - sum_k L_J[c][k] L_I[i][k]
//   is four 16x16 MFMA tiles (cb, ib); lane (g = lane/16, j = lane%16), register r of tile (cb, ib) holds
//   c = 16cb + 4r + g, i = 16ib + j  (the fp64 accumulator interleaves rows over the lane groups: probed on gfx950).
//     update step k..k+3:  A[c][k'] = L[32J+16cb+j][k+g],  B[k'][i] = L[32I+16ib+j][k+g]   (one f64 per lane each, read
//                          straight from the packed operator: 16 consecutive rows of one column per lane group)
//     panel  L_IJ' = inv(L_JJ) S':  the accumulator registers are the B operands again -- MFMA (cb, r) contracts over
//                          c = 16cb + 4r + g, so A[c'][k'=g] = inv(L_JJ)[16cb'+j][16cb+4r+g]  (from LDS)
//   and the result stores column-major as 128-byte segments.
// Wave 0 factors and inverts the 32x32 diagonal tile in registers (lane = row, v_readlane broadcasts) while waves 1-3
// run their update streams.  fp64 MFMA peak on MI355X is 78.6 TFLOP/s = the fp64 vector peak; as in fp32 the gain over
// the VALU kernel is operand traffic and instruction count.
#include "bcbf_common.h"
#include <stdlib.h>

namespace bcbf {

using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;

#ifndef BCBF_R64_MAXT
#define BCBF_R64_MAXT 1
#endif
#ifndef BCBF_R64_OCC
#define BCBF_R64_OCC 2
#endif
#ifndef BCBF_R64_KS
#define BCBF_R64_KS 4
#endif
constexpr int MT64 = 256;                  // threads
constexpr int MAXT64 = BCBF_R64_MAXT;      // row tiles a wave processes together (share the A operands)

__device__ inline double rlane64(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// v(lane) + v(lane ^ 32) in every lane, on the VALU: gfx950's v_permlane32_swap hands each lane the low-half and the
// high-half value of its column (a __shfl_xor would be an LDS round trip, 64 times per diagonal tile)
__device__ inline double half_sum64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double a = __builtin_bit_cast(double, ((long long)r1[0] << 32) | (long long)r0[0]);
    const double c = __builtin_bit_cast(double, ((long long)r1[1] << 32) | (long long)r0[1]);
    return a + c;
}
// 1/sqrt(p): hardware estimate + two Newton steps (full double precision; a divide per pivot would serialise wave 0)
__device__ inline double rsqrt_nr(double p) {
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);
    return y;
}

// NW waves per workgroup: 4 for batches (several workgroups per CU), more when only a few instances are in flight
// (one GP at a time, the reference's own use): the row tiles of a block column then run side by side.
template <bool FROM_DENSE, int NW>
__global__ void __launch_bounds__(64 * NW, (NW == 4 ? BCBF_R64_OCC : 2))
refit_mfma64_kernel(const double* __restrict__ X, const double* __restrict__ UH, const double* __restrict__ Bm,
                    const double* __restrict__ ell, const double* __restrict__ s2p, const double* __restrict__ jitter,
                    const double* __restrict__ Kdense, double* __restrict__ Lop, double* __restrict__ UHBout,
                    double* __restrict__ Ldense, int* __restrict__ info, int N, int Np, int n, int C, const int* only_bad) {
    constexpr int V = 2;
    __shared__ double dS[NB][NB + 1];                        // diagonal tile S_JJ (row c, col i)
    __shared__ double dinv[NB][NB + 1];                      // inv(L_JJ)[c'][c]
    __shared__ double colX[NB][BCBF_MAX_STATE_DIM];
    __shared__ double colUH[NB][BCBF_MAX_CTRL_DIM + 1];
    __shared__ double idg[NB];                               // 1 / L_JJ[c][c]
    __shared__ int fail;

    constexpr int MTT = 64 * NW;                              // threads
    const int b = blockIdx.x, tid = threadIdx.x;
    if (only_bad != nullptr && only_bad[b] == 0) { if (tid == 0) info[b] = 0; return; }     // bcbf_refit_retry: factored already
    const int wave = tid >> 6, lane = tid & 63, j16 = lane & 15, g = lane >> 4;
    double* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    const double* Xb = FROM_DENSE ? nullptr : X + (size_t)b * N * n;
    const double* UHb = FROM_DENSE ? nullptr : UH + (size_t)b * N * C;
    const double* Kb = FROM_DENSE ? Kdense + (size_t)b * N * N : nullptr;
    double* Ld = Ldense ? Ldense + (size_t)b * N * N : nullptr;
    double iell[BCBF_MAX_STATE_DIM], Bmr[(BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1)];
    double s2 = 0.0;
    if (!FROM_DENSE) {
        s2 = s2p[b];
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) iell[d] = d < n ? 1.0 / ell[(size_t)b * n + d] : 0.0;
#pragma unroll
        for (int a = 0; a < (BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1); ++a)
            Bmr[a] = a < C * C ? Bm[(size_t)b * C * C + a] : 0.0;
        for (int i = tid; i < N; i += MTT)
            for (int c = 0; c < C; ++c) {
                double s = 0.0;
                for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                UHBout[((size_t)b * N + i) * C + c] = s;
            }
    }
    if (tid == 0) fail = 0;
    if (Ld)
        for (int e = tid; e < N * N; e += MTT) { const int i = e / N, j = e - i * N; if (j > i) Ld[e] = 0.0; }
    __syncthreads();

    const int nblk = Np / NB;
    for (int J = 0; J < nblk; ++J) {
        const int col0 = J * NB;
        if (!FROM_DENSE) {
            for (int e = tid; e < NB * n; e += MTT) {
                const int c = e / n, d = e - c * n;
                colX[c][d] = (col0 + c < N) ? Xb[(size_t)(col0 + c) * n + d] : 0.0;
            }
            for (int e = tid; e < NB * C; e += MTT) {
                const int c = e / C, a = e - c * C;
                colUH[c][a] = (col0 + c < N) ? UHb[(size_t)(col0 + c) * C + a] : 0.0;
            }
        }
        __syncthreads();
        const int ntile = nblk - J;                          // row tiles I = J .. nblk-1
        // Tile schedule (as refit_mfma.hip).  Group 0: wave 0 takes ONLY the diagonal tile, then factors and inverts
        // it while waves 1-3 run the updates of tiles 1 .. 3*MAXT64.  Later groups: round-robin over all four waves.
        const int first = 1 + (NW - 1) * MAXT64;
        const int ngroups = ntile <= first ? 1 : 1 + (ntile - first + NW * MAXT64 - 1) / (NW * MAXT64);
        for (int gq = 0; gq < ngroups; ++gq) {                   // uniform trip count: barrier (B) is inside
            f64x4 acc[MAXT64][2][2];                             // [slot][cb][ib]
            int irow[MAXT64], tix[MAXT64];
            bool live[MAXT64];
#pragma unroll
            for (int q = 0; q < MAXT64; ++q) {
                int t;
                if (gq == 0) t = wave == 0 ? (q == 0 ? 0 : ntile) : 1 + (wave - 1) + (NW - 1) * q;
                else t = first + (gq - 1) * NW * MAXT64 + wave + NW * q;
                tix[q] = t;
                live[q] = t < ntile;
                irow[q] = (live[q] ? (J + t) * NB : col0) + j16;    // + 16*ib; dead slots shadow the diagonal tile (no stores)
                // ---- initial value: K_b'(c, i) for this lane's two rows and the 8 c's per row of its accumulators
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = irow[q] + 16 * ib;
                    double xi[BCBF_MAX_STATE_DIM], ub[BCBF_MAX_CTRL_DIM + 1], jit = 0.0;
                    if (!FROM_DENSE && i < N) {
#pragma unroll
                        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) xi[d] = d < n ? Xb[(size_t)i * n + d] : 0.0;
#pragma unroll
                        for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) {
                            double s = 0.0;
                            if (c < C) for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                            ub[c] = s;
                        }
                        jit = jitter ? jitter[(size_t)b * N + i] : 0.0;
                    }
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * cb + 4 * r + g, j = col0 + c;
                            double val;
                            if (i >= N || j >= N) val = (i == j) ? 1.0 : 0.0;          // padding: identity
                            else if (FROM_DENSE) val = (j <= i) ? Kb[(size_t)i * N + j] : Kb[(size_t)j * N + i];
                            else {
                                double d2 = 0.0, uu = 0.0;
#pragma unroll
                                for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d)
                                    if (d < n) { const double z = (xi[d] - colX[c][d]) * iell[d]; d2 += z * z; }
#pragma unroll
                                for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a)
                                    if (a < C) uu += ub[a] * colUH[c][a];
#ifdef BCBF_R64_ABL_NOEXP
                                val = s2 * (1.0 - 0.01 * d2) * uu + (i == j ? jit + 10.0 : 0.0);
#else
                                val = s2 * exp(-0.5 * d2) * uu + (i == j ? jit : 0.0);
#endif
                            }
                            acc[q][cb][ib][r] = val;
                        }
                }
            }
            // ---- S' -= L_J L_I'  over all previous columns: stages of KS k-steps (4 columns each); the next stage's
            //      operands are in flight while the current stage's MFMAs issue
            constexpr int KS = BCBF_R64_KS;
            const int kend = col0;                                            // multiple of 32
            double a_nxt[KS][2], b_nxt[KS][MAXT64][2];
            auto fetch = [&](int kk) {
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    const int base = lop_base<V>(kk + 4 * s_ + g, Np);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) a_nxt[s_][cb] = lop[base + col0 + 16 * cb + j16];
#pragma unroll
                    for (int q = 0; q < MAXT64; ++q)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) b_nxt[s_][q][ib] = lop[base + irow[q] + 16 * ib];
                }
            };
#ifndef BCBF_R64_ABL_NOKLOOP
            if (kend > 0) fetch(0);
            for (int kk = 0; kk < kend; kk += 4 * KS) {
                double a_cur[KS][2], b_cur[KS][MAXT64][2];
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) a_cur[s_][cb] = -a_nxt[s_][cb];        // D = (-A) B + C
#pragma unroll
                    for (int q = 0; q < MAXT64; ++q)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) b_cur[s_][q][ib] = b_nxt[s_][q][ib];
                }
                if (kk + 4 * KS < kend) fetch(kk + 4 * KS);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                    for (int q = 0; q < MAXT64; ++q)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib)
                                acc[q][cb][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[s_][cb], b_cur[s_][q][ib],
                                                                                      acc[q][cb][ib], 0, 0, 0);
            }
#endif
            if (gq == 0) {
                // ---- wave 0: diagonal tile -> LDS -> factor + invert in registers (lane = row); written and read by
                //      this wave only, so no workgroup barrier: LDS operations of one wave complete in order
                if (wave == 0) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dS[16 * cb + 4 * r + g][16 * ib + j16] = acc[0][cb][ib][r];
                    __builtin_amdgcn_wave_barrier();
                    // Left-looking 32x32 Cholesky and triangular inverse with lane = row (resp. column), entirely out of
                    // LDS with rolled loops: a register formulation (row[32], x[32] doubles, v_readlane broadcasts) costs
                    // 128+ VGPRs, which caps the kernel at one workgroup per CU, and is no faster.  S is symmetric, so
                    // row c of S' is column c of S; it is overwritten by column c of L as soon as it has been consumed.
                    const int ln = lane & (NB - 1), lh = lane >> 5;
                    int bad = 0;
#ifndef BCBF_R64_ABL_NOFACTOR
                    // Both 32-step loops are FULLY unrolled: with c a compile-time constant every LDS address is a per-lane
                    // base plus an immediate and no term needs a mask except the last one of an odd c.  Rolled, each
                    // dot-product term cost ~12 instructions of address arithmetic / selects plus an LDS round trip (a
                    // per-lane trip count even compiled to one read-wait-multiply per term): 31 k cycles per loop.
                    const double* rowL = &dS[lh][ln];            // L[ln][lh + 2t]  at  rowL[2t (NB+1)]
                    const double* colL = &dS[lh][0];             // L[c][lh + 2t]   at  colL[2t (NB+1) + c]
                    const double* rowX = &dinv[lh][ln];          // X[lh + 2t][ln]  at  rowX[2t (NB+1)]
#pragma unroll
                    for (int c = 0; c < NB; ++c) {
                        double v = lh ? 0.0 : dS[c][ln], v2 = 0.0;                        // S[lane][c]; two partial sums
#pragma unroll
                        for (int t = 0; 2 * t < c; ++t) {                                 // k = 2t + lh: even k in the low half,
                            const double term = rowL[2 * t * (NB + 1)] * colL[2 * t * (NB + 1) + c];   // odd k in the high
                            const double tm = (2 * t + 1 < c || !lh) ? term : 0.0;
                            if (t & 1) v2 -= tm; else v -= tm;
                        }
                        v = half_sum64(v + v2);
                        const double piv = rlane64(v, c);
                        if (!(piv > 0.0) && bad == 0) bad = col0 + c + 1;
                        const double inv = rsqrt_nr(piv > 0.0 ? piv : 1.0);
                        if (lane < NB) dS[c][lane] = lane == c ? piv * inv : (lane > c ? v * inv : 0.0);   // L[lane][c]
                        if (lane == c) idg[c] = inv;
                        __builtin_amdgcn_wave_barrier();       // other lanes read this column through LDS
                    }
                    if (Ld && lane < NB && col0 + lane < N) {
                        for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = dS[c][lane];
                    }
                    {
                        const int base = lop_dinv_block(J, Np) + lop_dinv_col(ln);       // lower triangle, packed
#pragma unroll
                        for (int i = 0; i < NB; ++i) {
                            double s_ = (ln == i && !lh) ? 1.0 : 0.0, s2 = 0.0;
#pragma unroll
                            for (int t = 0; 2 * t < i; ++t) {                             // L[i][k] X[k][lane], k = 2t + lh
                                const double term = colL[2 * t * (NB + 1) + i] * rowX[2 * t * (NB + 1)];
                                const double tm = (2 * t + 1 < i || !lh) ? term : 0.0;
                                if (t & 1) s2 -= tm; else s_ -= tm;
                            }
                            s_ = half_sum64(s_ + s2);
                            const double xi = s_ * idg[i];
                            if (lane < NB) { dinv[i][lane] = xi; if (i >= lane) lop[base + i] = xi; lop[lop_dfull(J, i, lane, Np)] = xi; }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
#else
                    if (lane < NB) {
                        const int base = lop_dinv_block(J, Np) + lop_dinv_col(lane);
                        for (int i = 0; i < NB; ++i) { const double xi = lane == i ? 1.0 : 1e-6 * dS[i][lane]; dinv[i][lane] = xi; if (i >= lane) lop[base + i] = xi; lop[lop_dfull(J, i, lane, Np)] = xi; }
                    }
#endif
                    if (lane < LOP_DB - 528) lop[lop_dinv_block(J, Np) + 528 + lane] = 0.0;      // the block's padding
                    if (lane == 0 && bad != 0 && bad <= N) fail = bad;
                }
                __syncthreads();          // (B) inv(L_JJ) visible
            }
            if (fail == 0) {
                // ---- panel:  L_IJ' = inv(L_JJ) S'   (accumulator registers of S' are the B operands)
                double ainv[2][2][4];                             // [cb'][cb][r]; inv(L_JJ) is lower triangular: (0,1) = 0
#pragma unroll
                for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                    for (int cb = 0; cb <= cbp; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ainv[cbp][cb][r] = dinv[16 * cbp + j16][16 * cb + 4 * r + g];
#pragma unroll
                for (int q = 0; q < MAXT64; ++q) {
                    const bool is_diag = tix[q] == 0;
                    if (!live[q] || is_diag) continue;
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        const int i = irow[q] + 16 * ib;
#pragma unroll
                        for (int cbp = 0; cbp < 2; ++cbp) {
                            f64x4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int cb = 0; cb <= cbp; ++cb)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    y = __builtin_amdgcn_mfma_f64_16x16x4f64(ainv[cbp][cb][r], acc[q][cb][ib][r], y, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 16 * cbp + 4 * r + g;
                                lop[lop_base<V>(col0 + c, Np) + i] = y[r];
                                if (Ld && i < N && col0 + c < N) Ld[(size_t)i * N + col0 + c] = y[r];
                            }
                        }
                    }
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        if (fail != 0) break;
    }
    if (tid == 0) info[b] = fail;
}

int launch_refit_wave64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                        const double* jitter, const double* Kdense, double* Lop, double* UHB, double* Ldense, int* info,
                        int Bt, int N, int Np, int n, int C, hipStream_t st);          // refit_wave64.hip
int launch_refit_pair64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                        const double* jitter, double* Lop, double* UHB, int* info, int Bt, int N, int Np, int n, int C,
                        hipStream_t st);                                               // refit_wave64.hip: two waves per instance


int launch_refit_team64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                        const double* jitter, const double* Kdense, double* Lop, double* UHB, double* Ldense, int* info, int Bt, int N,
                        int Np, int n, int C, int nw, hipStream_t st);                 // refit_wave64.hip: a team of eight (four) waves per instance

}  // namespace bcbf

extern "C" int bcbf_refit_mfma_f64(const double* X, const double* UH, const double* Bm, const double* ell,
                                   const double* s2, const double* jitter, const double* Kdense, double* Lop,
                                   double* UHB, double* Ldense, int* info, int Bt, int N, int n, int m, void* stream) {
    using namespace bcbf;
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !info || N < 1) return BCBF_EINVAL;
    const int Np = round_up(N, NB);
    hipStream_t st = (hipStream_t)stream;
    if (!Kdense) {
        if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
        if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    }
    // Batches: one wave per instance (refit_wave64.hip) -- every SIMD advances its own factorisation chain.
    // Measured (tools/bench_refit_forms.py, MI355X): the per-wave form wins from 1024 instances on at every N (N = 256:
    // 0.45 vs 0.82 ms at 1024, 1.75 vs 3.2 ms at 4096; N = 512: 2.1 vs 3.1 and 8.2 vs 12.0; N = 1024: 15.0 vs 15.7),
    // from 512 at N <= 256 and from 64 at N <= 128; below that the workgroup form, whose four waves share one
    // instance's update stream.  (The 4-column-blocked diagonal tile of the per-wave form was also tried here for wave 0:
    // it needs ~100 more registers than this kernel's 256 budget leaves, spills, and is 7 % slower than the 32-step
    // form below.)
    bool per_wave = Bt >= 1024 || (Bt >= 512 && Np <= 512) || (Bt >= 64 && Np <= 128);
    if (const char* e = getenv("BCBF_REFIT_WAVE")) per_wave = e[0] == '1';
    // Two waves per instance (refit_wave64.hip, round 3: the serial chain of diagonal tiles on one wave, every other tile on
    // the other) wins for small systems while one round of workgroups holds the batch -- measured (tools/bench_refit_forms.py,
    // ms workgroup / wave / two waves): 1024 x 256: 0.822 / 0.357 / 0.328, 64 x 256: 0.303 / 0.311 / 0.204, 1024 x 128: 0.319 /
    // 0.104 / 0.100, 64 x 128: 0.132 / 0.093 / 0.066; a tie at 4096 (x 256: 3.24 / 1.367 / 1.390) -- and loses at N = 512
    // (256 x 512: 1.12 / 1.50 / 1.40, 1024 x 512: 3.06 / 1.88 / 2.26), where the one-wave form (512 registers) keeps more of the
    // update stream in flight.  BCBF_REFIT_PAIR=0/1 forces the choice (N <= 512).
    // (ONE model, tools/dev/time_refit_one.py, us workgroup / two waves / team: N = 128: 133 / 61 / 65, 256: 308 / 199 / 147, 512:
    // 901 / 1193 / 423, 1024: 3565 / - / 2311.)
    bool pair = Bt <= 1024 && Np <= 256;
    if (const char* e = getenv("BCBF_REFIT_PAIR")) pair = e[0] == '1' && Np / NB <= 16;
    // A team of eight waves per instance (refit_wave64.hip: one chain wave, seven bulk waves; one workgroup per CU at a
    // time) while ONE round of workgroups holds the batch, two rounds from N = 512 on, four from 1024 (1024 x 1024 fp32: 6.57 / 6.90)
    // -- measured on 256 CUs
    // (tools/dev/check_refit_team.py, ms team / best other form), fp32: 1 x 512: 0.27 / 0.71, 256 x 512: 0.30 / 0.86, 512 x 512:
    // 0.61 / 0.93, 1024 x 512: 1.22 / 1.04, 1 x 1024: 1.31 / 2.81, 512 x 1024: 3.32 / 5.89, 256 x 256: 0.108 / 0.140,
    // 512 x 256: 0.21 / 0.16, 256 x 128: 0.047 / 0.045; fp64: 1 x 512: 0.43 / 0.90, 256 x 512: 0.52 / 1.12, 512 x 512:
    // 1.04 / 1.55, 1 x 1024: 2.31 / 3.57, 1 x 2048: 16.8 / 19.3, 256 x 256: 0.173 / 0.219.  BCBF_REFIT_TEAM=0/1 forces the
    // choice (N <= 8192)
    static int cus_ = 0;
    if (cus_ == 0) {
        int dev_ = 0, c_ = 256;
        (void)hipGetDevice(&dev_);
        (void)hipDeviceGetAttribute(&c_, hipDeviceAttributeMultiprocessorCount, dev_);
        cus_ = c_ > 0 ? c_ : 256;
    }
    bool team = Np / NB <= 64 &&       // (N <= 2048: beyond, no better than the workgroup form -- 1 x 4096 fp64: 128 / 125 ms)
                ((Np >= 256 && Bt <= cus_) || (Np >= 512 && Bt <= 2 * cus_) || (Np >= 1024 && Bt <= ((Kdense || Ldense) ? 4 : 2) * cus_));   // (1024 x 1024 fp64: team 13.2, one wave with super-panels 12.8 ms)
    // ... of four waves (two workgroups per CU) for cus < batch <= 2 cus at 256 <= N <= 512: 512 x 256 fp64 0.259 (two waves per
    // instance) / 0.210, fp32 0.162 / 0.136; 512 x 512 fp32 0.62 (team of eight, two rounds) / 0.53
    int team_nw = 8;
    if (Np / NB <= 16 && Np >= 256 && Bt > cus_ && Bt <= 2 * cus_ && !Kdense) { team = true; team_nw = 4; }
    if (getenv("BCBF_REFIT_WAVE") || getenv("BCBF_REFIT_PAIR")) team = false;      // (another form is being forced)
    if (const char* e = getenv("BCBF_REFIT_TEAM")) { team = e[0] == '1' && Np / NB <= 256; if (e[0] == '1' && e[1] == '4') team_nw = 4; else if (e[0] == '1' && e[1] == '8') team_nw = 8; }
    if (team) {
        if (!Kdense) {
            if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
            if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
        }
        if (launch_refit_team64(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, m + 1, team_nw, st) != 0) return BCBF_EINVAL;       // a size the form does not address
        return check_launch("refit_team64");
    }
    if (pair && !Kdense && !Ldense) {
        if (launch_refit_pair64(X, UH, Bm, ell, s2, jitter, Lop, UHB, info, Bt, N, Np, n, m + 1, st) != 0) return BCBF_EINVAL;       // a size the form does not address
        return check_launch("refit_pair64");
    }
    if (per_wave) {
        launch_refit_wave64(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, m + 1, st);
        return check_launch("refit_wave64");
    }
    const bool wide = Bt < 128 && N >= 256;  // few large instances: 8 waves per workgroup (16 would cap the VGPRs at 128: spills)
#define BCBF_REFIT_LAUNCH(DENSE, ...)                                                                   \
    do {                                                                                                \
        if (wide) hipLaunchKernelGGL((refit_mfma64_kernel<DENSE, 8>), dim3(Bt), dim3(512), __VA_ARGS__);                \
        else hipLaunchKernelGGL((refit_mfma64_kernel<DENSE, 4>), dim3(Bt), dim3(256), __VA_ARGS__);                     \
    } while (0)
    if (Kdense) {
        BCBF_REFIT_LAUNCH(true, 0, st, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, Kdense, Lop, nullptr, Ldense, info, N, Np, 0, 0, g_refit_only_bad);
    } else {
        BCBF_REFIT_LAUNCH(false, 0, st, X, UH, Bm, ell, s2, jitter, nullptr,
                           Lop, UHB, Ldense, info, N, Np, n, m + 1, g_refit_only_bad);
    }
    return check_launch("refit_mfma64");
}
