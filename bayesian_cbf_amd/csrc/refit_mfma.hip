// K1 + K2 on the matrix cores (fp32): blocked left-looking Cholesky, one 256-thread workgroup per
// instance, every rank-32 update and every panel solve as v_mfma_f32_32x32x2_f32 tiles whose operands
// come straight from the packed operator (column-major => a lane reads 32 consecutive rows of one
// column, fully coalesced) -- no LDS staging, no transposes:
//
//   block column J, row tile I (32 rows), transposed tile  S'[c][i] = K_b(i, 32J+c) - sum_kk L_J[c][kk] L_I[i][kk]
//     MFMA step kk..kk+1:  A[c][k] = L[32J+c][kk+k],  B[k][i] = L[32I+i][kk+k]   (one dword per lane each)
//   the accumulator of S' has the row index i on the lane and c in the registers, so
//     L_I,J' = inv(L_JJ) S'   is again an MFMA chain with the accumulator registers as B operands
//     (register r holds rows rho(r), rho(r)+4  ->  A_r[c][k] = inv(L_JJ)[c][rho(r) + 4k]),
//   and its result is stored column-major with 128-byte contiguous segments.
// The 32x32 diagonal tile goes through LDS once to wave 0, which factors and inverts it in registers.
// fp32 MFMA runs at the fp32 vector rate on gfx950: the gain is operand traffic / instruction count
// (2 loads + 1 MFMA per 2048 MACs instead of LDS broadcasts), not peak.
#include "bcbf_common.h"
#include "diag_tile64.h"
#include <stdlib.h>

#ifndef BCBF_R32_WAVE_MIN_BATCH
#define BCBF_R32_WAVE_MIN_BATCH 1024
#endif
namespace bcbf {

// v(lane) + v(lane ^ 32) in every lane on the VALU (gfx950 v_permlane32_swap; a __shfl_xor is an LDS round trip)
__device__ inline float half_sum32(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}


using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

constexpr int MT = 256;          // threads
#ifndef BCBF_R32_MAXT
#define BCBF_R32_MAXT 1
#endif
constexpr int MAXT = BCBF_R32_MAXT;   // row tiles a wave processes together (shares the A operand)

__device__ inline int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// broadcast of one lane's value through an SGPR (v_readlane_b32, compile-time lane): no VGPR, no LDS crossbar
__device__ inline float rlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

#ifndef BCBF_R32_OCC
#define BCBF_R32_OCC 4
#endif
// NW waves per workgroup: 4 for batches (several workgroups per CU), more when only a few instances are in flight
// (one GP at a time, the reference's own use): the row tiles of a block column then run side by side.
template <bool FROM_DENSE, int NW>
__global__ void __launch_bounds__(64 * NW, (NW == 4 ? BCBF_R32_OCC : 1))
refit_mfma_kernel(const float* __restrict__ X, const float* __restrict__ UH, const float* __restrict__ Bm,
                  const float* __restrict__ ell, const float* __restrict__ s2p, const float* __restrict__ jitter,
                  const float* __restrict__ Kdense, float* __restrict__ Lop, float* __restrict__ UHBout,
                  float* __restrict__ Ldense, int* __restrict__ info, int N, int Np, int n, int C, const int* only_bad) {
    constexpr int V = 4;
    __shared__ DiagTile<float> dt;                           // the diagonal tile's working set (diag_tile64.h)
    float (&dS)[NB][NB + 1] = dt.tile;                       // diagonal tile S_JJ (row c, col i); after the factorisation: L
    float (&dinv)[NB][NB + 1] = dt.xinv;                     // inv(L_JJ)[c][c']
    __shared__ float colX[NB][BCBF_MAX_STATE_DIM];
    __shared__ float colUH[NB][BCBF_MAX_CTRL_DIM + 1];
    __shared__ float idg[NB];                                // 1 / L_JJ[c][c]
    __shared__ int fail;

    constexpr int MTT = 64 * NW;                              // threads
    const int b = blockIdx.x, tid = threadIdx.x;
    if (only_bad != nullptr && only_bad[b] == 0) { if (tid == 0) info[b] = 0; return; }     // bcbf_refit_retry: factored already
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    float* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    const float* Xb = FROM_DENSE ? nullptr : X + (size_t)b * N * n;
    const float* UHb = FROM_DENSE ? nullptr : UH + (size_t)b * N * C;
    const float* Kb = FROM_DENSE ? Kdense + (size_t)b * N * N : nullptr;
    float* Ld = Ldense ? Ldense + (size_t)b * N * N : nullptr;
    float iell[BCBF_MAX_STATE_DIM], Bmr[(BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1)];
    float s2 = 0.f;
    if (!FROM_DENSE) {
        s2 = s2p[b];
#pragma unroll
        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) iell[d] = d < n ? 1.f / ell[(size_t)b * n + d] : 0.f;
#pragma unroll
        for (int a = 0; a < (BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1); ++a)
            Bmr[a] = a < C * C ? Bm[(size_t)b * C * C + a] : 0.f;
        for (int i = tid; i < N; i += MTT)
            for (int c = 0; c < C; ++c) {
                float s = 0.f;
                for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                UHBout[((size_t)b * N + i) * C + c] = s;
            }
    }
    if (tid == 0) fail = 0;
    if (Ld)
        for (int e = tid; e < N * N; e += MTT) { const int i = e / N, j = e - i * N; if (j > i) Ld[e] = 0.f; }
    __syncthreads();

    const int nblk = Np / NB;
    for (int J = 0; J < nblk; ++J) {
        const int col0 = J * NB;
        if (!FROM_DENSE) {
            for (int e = tid; e < NB * n; e += MTT) {
                const int c = e / n, d = e - c * n;
                colX[c][d] = (col0 + c < N) ? Xb[(size_t)(col0 + c) * n + d] : 0.f;
            }
            for (int e = tid; e < NB * C; e += MTT) {
                const int c = e / C, a = e - c * C;
                colUH[c][a] = (col0 + c < N) ? UHb[(size_t)(col0 + c) * C + a] : 0.f;
            }
        }
        __syncthreads();
        const int ntile = nblk - J;                          // row tiles I = J .. nblk-1
        // Tile schedule.  Group 0: wave 0 takes ONLY the diagonal tile, then factors and inverts it while
        // waves 1-3 run the updates of tiles 1 .. 3*MAXT -- the serial factorization hides behind their MFMA
        // streams.  Later groups: the remaining tiles round-robin over all four waves, MAXT per wave.
        const int first = 1 + (NW - 1) * MAXT;
        const int ngroups = ntile <= first ? 1 : 1 + (ntile - first + NW * MAXT - 1) / (NW * MAXT);
        for (int g = 0; g < ngroups; ++g) {                      // uniform trip count: barrier (B) is inside
            f32x16 acc[MAXT];
            int irow[MAXT], tix[MAXT];
            bool live[MAXT];
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                int t;
                if (g == 0) t = wave == 0 ? (q == 0 ? 0 : ntile) : 1 + (wave - 1) + (NW - 1) * q;
                else t = first + (g - 1) * NW * MAXT + wave + NW * q;
                tix[q] = t;
                const int I = J + t;
                live[q] = t < ntile;
                irow[q] = live[q] ? I * NB + li : col0 + li;         // dead slots shadow the diagonal tile (no stores)
                // ---- initial value: K_b'(c, i) for this lane's row i and the 16 c's of its accumulator rows
                const int i = irow[q];
                float xi[BCBF_MAX_STATE_DIM], ub[BCBF_MAX_CTRL_DIM + 1], jit = 0.f;
                if (!FROM_DENSE && i < N) {
#pragma unroll
                    for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d) xi[d] = d < n ? Xb[(size_t)i * n + d] : 0.f;
#pragma unroll
                    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) {
                        float s = 0.f;
                        if (c < C) for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                        ub[c] = s;
                    }
                    jit = jitter ? jitter[(size_t)b * N + i] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = acc_row(r, lh), j = col0 + c;
                    float val;
                    if (i >= N || j >= N) val = (i == j) ? 1.f : 0.f;          // padding: identity
                    else if (FROM_DENSE) val = (j <= i) ? Kb[(size_t)i * N + j] : Kb[(size_t)j * N + i];
                    else {
                        float d2 = 0.f, uu = 0.f;
#pragma unroll
                        for (int d = 0; d < BCBF_MAX_STATE_DIM; ++d)
                            if (d < n) { const float z = (xi[d] - colX[c][d]) * iell[d]; d2 += z * z; }
#pragma unroll
                        for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a)
                            if (a < C) uu += ub[a] * colUH[c][a];
                        val = s2 * expf(-0.5f * d2) * uu + (i == j ? jit : 0.f);
                    }
                    acc[q][r] = val;
                }
            }
            // ---- S' -= L_J L_I'  over all previous columns: stages of KS MFMA k-steps (2 columns each),
            //      the next stage's operands are in flight while the current stage's MFMAs issue
#ifndef BCBF_R32_KS
#define BCBF_R32_KS 4
#endif
            constexpr int KS = BCBF_R32_KS;
            const int kend = col0;                                            // multiple of 32
            float a_nxt[KS], b_nxt[KS][MAXT];
            auto fetch = [&](int kk) {
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    const int base = lop_base<V>(kk + 2 * s_ + lh, Np);
                    a_nxt[s_] = lop[base + col0 + li];
#pragma unroll
                    for (int q = 0; q < MAXT; ++q) b_nxt[s_][q] = lop[base + irow[q]];
                }
            };
#ifndef BCBF_ABL_SKIP_KLOOP
            if (kend > 0) fetch(0);
            for (int kk = 0; kk < kend; kk += 2 * KS) {
                float a_cur[KS], b_cur[KS][MAXT];
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    a_cur[s_] = -a_nxt[s_];                                   // D = (-A) B + C
#pragma unroll
                    for (int q = 0; q < MAXT; ++q) b_cur[s_][q] = b_nxt[s_][q];
                }
                if (kk + 2 * KS < kend) fetch(kk + 2 * KS);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                    for (int q = 0; q < MAXT; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s_], b_cur[s_][q], acc[q], 0, 0, 0);
            }
#endif
            // ---- diagonal tile -> LDS (tile I == J is slot q = 0 of wave 0, first group)
            if (g == 0) {
            // ---- wave 0: diagonal tile -> LDS -> factor + invert there; written and read by this wave only, so no
            //      workgroup barrier: LDS operations of one wave complete in order
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dS[acc_row(r, lh)][li] = acc[0][r];
                __builtin_amdgcn_wave_barrier();
                // Left-looking 32x32 Cholesky and triangular inverse out of LDS with rolled loops (lane = row resp. column,
                // even k in the low half of the wave, odd k in the high half): see refit_mfma64.hip.
                const int ln = lane & (NB - 1), lhh = lane >> 5;
                int bad = 0;
#if defined(BCBF_ABL_SKIP_FACTOR)
                if (lane < NB) {
                    const int base = lop_dinv_block(J, Np) + lop_dinv_col(lane);
                    for (int i = 0; i < NB; ++i) { const float xi = lane == i ? 1.f : 1e-6f * dS[i][lane]; dinv[i][lane] = xi; if (i >= lane) lop[base + i] = xi; lop[lop_dfull(J, i, lane, Np)] = xi; }
                }
#else
                if constexpr (NW == 8) {      // few instances in flight, 256-register budget: the 4-column-blocked tile routine
                bad = diag_factor_invert<float>(BCBF_LDS_TILE(float, dt), lane);
                if (bad != 0) bad += col0;
                __builtin_amdgcn_wave_barrier();
                if (Ld && lane < NB && col0 + lane < N) {
                    for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = dt.tile[lane][c];
                }
                {
                    const int bfull = lop_dfull_block(J, Np), bpack = lop_dinv_block(J, Np);
#pragma unroll
                    for (int t = 0; t < NB * NB / 64; ++t) {
                        const int e = lane + 64 * t, c = e >> 5, r = e & 31;
                        const float xv = dt.xinv[r][c];
                        lop[bfull + e] = xv;
                        if (r >= c) lop[bpack + lop_dinv_col(c) + r] = xv;
                    }
                }
                } else
                if constexpr (NW != 4) {      // few instances in flight: latency of one GP counts; the batch form (NW = 4, 128 VGPRs) would spill
                // Fully unrolled (see refit_mfma64.hip): constant LDS offsets, the half-wave sums on the VALU
                // (v_permlane32_swap), two partial sums per dot product.
                const float* rowL = &dS[lhh][ln];            // L[ln][lhh + 2t]  at  rowL[2t (NB+1)]
                const float* colL = &dS[lhh][0];             // L[c][lhh + 2t]   at  colL[2t (NB+1) + c]
                const float* rowX = &dinv[lhh][ln];          // X[lhh + 2t][ln]  at  rowX[2t (NB+1)]
                constexpr int LDW = sizeof(dS[0]) / sizeof(float);           // row pitch of dS / dinv
                static_assert(sizeof(dS[0]) == sizeof(dinv[0]), "one pitch for both tiles");
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    float v = lhh ? 0.f : dS[c][ln], v2 = 0.f;                            // S[lane][c]
#pragma unroll
                    for (int t = 0; 2 * t < c; ++t) {                                     // k = 2t + lhh
                        const float term = rowL[2 * t * LDW] * colL[2 * t * LDW + c];     // L[lane][k] L[c][k]
                        const float tm = (2 * t + 1 < c || !lhh) ? term : 0.f;
                        if (t & 1) v2 -= tm; else v -= tm;
                    }
                    v = half_sum32(v + v2);
                    const float piv = rlane(v, c);
                    if (!(piv > 0.f) && bad == 0) bad = col0 + c + 1;
                    const float inv = __builtin_amdgcn_rsqf(piv > 0.f ? piv : 1.f);      // 1-ulp rsq
                    if (lane < NB) dS[c][lane] = lane == c ? piv * inv : (lane > c ? v * inv : 0.f);   // L[lane][c]
                    if (lane == c) idg[c] = inv;
                    __builtin_amdgcn_wave_barrier();       // other lanes read this column through LDS
                }
                if (Ld && lane < NB && col0 + lane < N) {
                    for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = dS[c][lane];
                }
                {
                    const int base = lop_dinv_block(J, Np) + lop_dinv_col(ln);           // lower triangle, packed
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        float s_ = (ln == i && !lhh) ? 1.f : 0.f, s2 = 0.f;
#pragma unroll
                        for (int t = 0; 2 * t < i; ++t) {                                 // L[i][k] X[k][lane], k = 2t + lhh
                            const float term = colL[2 * t * LDW + i] * rowX[2 * t * LDW];
                            const float tm = (2 * t + 1 < i || !lhh) ? term : 0.f;
                            if (t & 1) s2 -= tm; else s_ -= tm;
                        }
                        s_ = half_sum32(s_ + s2);
                        const float xi = s_ * idg[i];
                        if (lane < NB) { dinv[i][lane] = xi; if (i >= lane) lop[base + i] = xi; lop[lop_dfull(J, i, lane, Np)] = xi; }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                } else {
                for (int c = 0; c < NB; ++c) {
                    float v = lhh ? 0.f : dS[c][ln];                                      // S[lane][c]
#pragma unroll 4
                    for (int k = lhh; k < c; k += 2) v -= dS[k][ln] * dS[k][c];           // L[lane][k] L[c][k]
                    v += __shfl_xor(v, 32, 64);
                    const float piv = rlane(v, c);
                    if (!(piv > 0.f) && bad == 0) bad = col0 + c + 1;
                    const float inv = __builtin_amdgcn_rsqf(piv > 0.f ? piv : 1.f);      // 1-ulp rsq
                    if (lane < NB) dS[c][lane] = lane == c ? piv * inv : (lane > c ? v * inv : 0.f);   // L[lane][c]
                    if (lane == c) idg[c] = inv;
                    __builtin_amdgcn_wave_barrier();       // other lanes read this column through LDS
                }
                if (Ld && lane < NB && col0 + lane < N) {
                    for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = dS[c][lane];
                }
                {
                    const int base = lop_dinv_block(J, Np) + lop_dinv_col(ln);           // lower triangle, packed
                    for (int i = 0; i < NB; ++i) {
                        float s_ = (ln == i && !lhh) ? 1.f : 0.f;
#pragma unroll 4
                        for (int k = lhh; k < i; k += 2) s_ -= dS[k][i] * dinv[k][ln];    // L[i][k] X[k][lane]
                        s_ += __shfl_xor(s_, 32, 64);
                        const float xi = s_ * idg[i];
                        if (lane < NB) { dinv[i][lane] = xi; if (i >= lane) lop[base + i] = xi; lop[lop_dfull(J, i, lane, Np)] = xi; }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                }
#endif
                if (lane < LOP_DB - 528) lop[lop_dinv_block(J, Np) + 528 + lane] = 0.f;      // the block's padding
                if (lane == 0 && bad != 0 && bad <= N) fail = bad;
            }
            __syncthreads();          // (B) inv(L_JJ) visible
            }
            if (fail == 0) {
                // ---- panel:  L_IJ' = inv(L_JJ) S'   (accumulator registers of S' are the B operands)
                float ainv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ainv[r] = dinv[li][acc_row(r, lh)];
#pragma unroll
                for (int q = 0; q < MAXT; ++q) {
                    const bool is_diag = tix[q] == 0;
                    if (!live[q] || is_diag) continue;
                    f32x16 y = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        y = __builtin_amdgcn_mfma_f32_32x32x2f32(ainv[r], acc[q][r], y, 0, 0, 0);
                    const int i = irow[q];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = acc_row(r, lh);
                        lop[lop_base<V>(col0 + c, Np) + i] = y[r];
                        if (Ld && i < N && col0 + c < N) Ld[(size_t)i * N + col0 + c] = y[r];
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

int launch_refit_wave32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                        const float* jitter, const float* Kdense, float* Lop, float* UHB, float* Ldense, int* info,
                        int Bt, int N, int Np, int n, int C, hipStream_t st);          // refit_wave64.hip
int launch_refit_pair32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                        const float* jitter, float* Lop, float* UHB, int* info, int Bt, int N, int Np, int n, int C,
                        hipStream_t st);                                               // refit_wave64.hip: two waves per instance


int launch_refit_slab32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                        const float* jitter, float* Lop, float* UHB, int* info, int Bt, int N, int Np, int n, int C,
                        hipStream_t st);                                               // refit_slab.hip: four waves per instance, operands staged through LDS
int launch_refit_team32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                        const float* jitter, const float* Kdense, float* Lop, float* UHB, float* Ldense, int* info, int Bt, int N,
                        int Np, int n, int C, int nw, hipStream_t st);                 // refit_wave64.hip: a team of eight (four) waves per instance

}  // namespace bcbf

extern "C" int bcbf_refit_mfma_f32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                                   const float* jitter, const float* Kdense, float* Lop, float* UHB, float* Ldense,
                                   int* info, int Bt, int N, int n, int m, void* stream) {
    using namespace bcbf;
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !info || N < 1) return BCBF_EINVAL;
    const int Np = round_up(N, NB);
    hipStream_t st = (hipStream_t)stream;
    // Batches: one wave per instance (refit_wave64.hip, also compiled for fp32); BCBF_REFIT_WAVE=0/1 forces a form
    bool per_wave = Bt >= BCBF_R32_WAVE_MIN_BATCH || (Bt >= 256 && Np <= 1024) || (Bt >= 128 && Np <= 512) || (Bt >= 64 && Np <= 128);
    if (const char* e = getenv("BCBF_REFIT_WAVE")) per_wave = e[0] == '1';
    // Two waves per instance (refit_wave64.hip): measured in fp32 (ms workgroup / wave / two waves): 1024 x 128: 0.172 /
    // 0.070 / 0.058, 4096 x 128: 0.67 / 0.20 / 0.22, 1024 x 256: 0.441 / 0.220 / 0.186, 4096 x 256: 1.76 / 0.69 / 0.84,
    // 256 x 512: 1.17 / 0.90 / 0.85, 512 x 512: 1.31 / 0.93 / 0.93, 1024 x 512: 1.69 / 1.04 / 1.18, 4096 x 512: 6.6 / 3.7 / 5.4,
    // 64 x 512: 0.72 / 0.87 / 0.75, 256 x 1024: 5.5 / 5.3 / -
    // (ONE model, tools/dev/time_refit_one.py, us workgroup / two waves / team: N = 128: 105 / 41 / 44, 256: 236 / 131 / 96, 512:
    // 705 / 741 / 264, 1024: 2820 / - / 1311.)
    bool pair = Bt <= 1024 && (Np <= 256 || (Np <= 512 && Bt >= 128 && Bt <= 256));
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
                ((Np >= 256 && Bt <= cus_) || (Np >= 512 && Bt <= 2 * cus_) || (Np >= 1024 && Bt <= ((Kdense || Ldense) ? 4 : 2) * cus_));   // (1024 x 1024 fp32: team 6.67, one wave with super-panels 5.29 ms)
    // ... of four waves (two workgroups per CU) for cus < batch <= 2 cus at 256 <= N <= 512: 512 x 256 fp64 0.259 (two waves per
    // instance) / 0.210, fp32 0.162 / 0.136; 512 x 512 fp32 0.62 (team of eight, two rounds) / 0.53
    int team_nw = 8;
    if (Np / NB <= 16 && Np >= 256 && Bt > cus_ && Bt <= 2 * cus_ && !Kdense) { team = true; team_nw = 4; }
    if (getenv("BCBF_REFIT_WAVE") || getenv("BCBF_REFIT_PAIR")) team = false;      // (another form is being forced)
    if (const char* e = getenv("BCBF_REFIT_TEAM")) { team = e[0] == '1' && Np / NB <= 256; if (e[0] == '1' && e[1] == '4') team_nw = 4; else if (e[0] == '1' && e[1] == '8') team_nw = 8; }
    // Large batches of N = 384 .. 512 (refit_slab.hip): BCBF_REFIT_SLAB=0/1 forces the choice
    bool slab = false;
    if (const char* e = getenv("BCBF_REFIT_SLAB")) slab = e[0] == '1';
    if (slab && !Kdense && !Ldense) {
        if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
        if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
        if (launch_refit_slab32(X, UH, Bm, ell, s2, jitter, Lop, UHB, info, Bt, N, Np, n, m + 1, st) == 0) return check_launch("refit_slab32");
    }
    if (team) {
        if (!Kdense) {
            if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
            if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
        }
        if (launch_refit_team32(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, m + 1, team_nw, st) != 0) return BCBF_EINVAL;       // a size the form does not address
        return check_launch("refit_team32");
    }
    if (pair && !Kdense && !Ldense) {
        if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
        if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
        if (launch_refit_pair32(X, UH, Bm, ell, s2, jitter, Lop, UHB, info, Bt, N, Np, n, m + 1, st) != 0) return BCBF_EINVAL;       // a size the form does not address
        return check_launch("refit_pair32");
    }
    if (per_wave) {
        if (!Kdense) {
            if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
            if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
        }
        launch_refit_wave32(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, m + 1, st);
        return check_launch("refit_wave32");
    }
    // few instances: 8 waves per workgroup with the 4-column-blocked diagonal tile (256-register budget) beat 16 waves with
    // the unrolled 32-step tile (128 registers: the blocked routine spills there) -- ONE model, fp32: 126 -> 107 us at N = 128,
    // 288 -> 235 at 256, 754 -> 702 at 512, 2991 -> 2815 at 1024.  BCBF_REFIT_WIDE8=0 keeps the 16-wave form
    static const int wide8 = [] { const char* e = getenv("BCBF_REFIT_WIDE8"); return e ? atoi(e) : 1; }();
    const bool wide = Bt < 128 && N >= 128;   // few instances: 16 waves per workgroup and the unrolled diagonal-tile code (Bt=1: 145 -> 117 us at N=128, 346 -> 273 at 256, 869 -> 725 at 512, 2115 -> 1990 at 1024)
#define BCBF_REFIT_LAUNCH(DENSE, ...)                                                                   \
    do {                                                                                                \
        if (wide && wide8) hipLaunchKernelGGL((refit_mfma_kernel<DENSE, 8>), dim3(Bt), dim3(512), __VA_ARGS__);       \
        else if (wide) hipLaunchKernelGGL((refit_mfma_kernel<DENSE, 16>), dim3(Bt), dim3(1024), __VA_ARGS__);         \
        else hipLaunchKernelGGL((refit_mfma_kernel<DENSE, 4>), dim3(Bt), dim3(256), __VA_ARGS__);                     \
    } while (0)
    if (Kdense) {
        BCBF_REFIT_LAUNCH(true, 0, st, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, Kdense, Lop, nullptr, Ldense, info, N, Np, 0, 0, g_refit_only_bad);
    } else {
        if (!X || !UH || !Bm || !ell || !s2 || !UHB) return BCBF_EINVAL;
        if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
        BCBF_REFIT_LAUNCH(false, 0, st, X, UH, Bm, ell, s2, jitter, nullptr,
                           Lop, UHB, Ldense, info, N, Np, n, m + 1, g_refit_only_bad);
    }
    return check_launch("refit_mfma");
}
