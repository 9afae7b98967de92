// K1 + K2, fp64, ONE WAVE PER INSTANCE (batches): the blocked left-looking Cholesky of refit_mfma64.hip re-parallelised.
//
// Why: in the workgroup-per-instance kernel the 32x32 diagonal tile of every block column is factored and inverted by
// ONE wave while the other three wait (at N = 256 that serial chain was 60 % of an instance's lifetime; MFMA pipe busy
// 15 %).  The chain cannot be shortened by more waves -- but a batch has thousands of independent chains.  Here an
// instance is owned by a single wave (no workgroup barriers, per-wave LDS), so every SIMD of the chip advances its own
// chain, and the chain itself is rebuilt around 4-column blocks:
//
//   diagonal tile S (32x32, symmetric), lane = (row r = lane & 31, column half hb = lane >> 5) holds S[r][16hb .. 16hb+15]
//   in registers.  8 steps, step s = columns 4s..4s+3:
//     publish the four raw columns (LDS) -> every lane factors AND inverts the 4x4 diagonal block redundantly (4 rsqrt
//     chains: the only serial part) -> lane r forms its row of the panel l[r][0..3] = P[r][:] inv(L4)' -> publish ->
//     rank-4 update of the lane's 16 elements.
//   inverse X = inv(L): lane = column c, 8 block steps; the two halves each form two of the four dot products of a
//     step and swap them (v_permlane32_swap), then multiply by the 4x4 inverse kept from the factor.
//   ~13 k cycles per tile instead of ~65 k.
//
// The rank-32 updates (v_mfma_f64_16x16x4_f64 on transposed tiles, operands one f64 per lane straight from the packed
// operator) and the panel solve (accumulator registers as B operands) are those of refit_mfma64.hip, run tile after tile
// by the owning wave.  Same outputs, same packed layout (bcbf_common.h), same info convention.
#include "bcbf_common.h"

namespace bcbf {

using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;

#ifndef BCBF_RW64_WPB
#define BCBF_RW64_WPB 2          // waves (= instances) per workgroup
#endif
#ifndef BCBF_RW64_OCC
#define BCBF_RW64_OCC 2          // waves per SIMD the register allocation aims at (256 VGPRs)
#endif
#ifndef BCBF_RW64_KS
#define BCBF_RW64_KS 4           // k-steps (of 4 columns) per software-pipeline stage of the update stream
#endif

// -DBCBF_RW64_PROF (development): per-section cycle counts of every instance land in Ldense[b][0][1..7]
#ifdef BCBF_RW64_PROF
#define RW_T0() long long t_ = __builtin_readcyclecounter()
#define RW_ACC(k) do { const long long n_ = __builtin_readcyclecounter(); prof[k] += n_ - t_; t_ = n_; } while (0)
#else
#define RW_T0() do {} while (0)
#define RW_ACC(k) do {} while (0)
#endif

constexpr int RW_WPB = BCBF_RW64_WPB;
constexpr int LS = NB + 1;       // padded row stride of the 32x32 LDS tile

struct RWShared {
    double tile[NB][LS];                      // S' (transfer), then L (row-major [r][c]), then X = inv(L) [row][col]
    double P[NB][4];                          // raw panel columns of the current 4-column step
    double Lp[NB][4];                         // solved panel rows l[r][0..3]
    double I4[NB / 4][12];                    // the eight 4x4 inverses (10 used)
    double colX[NB][BCBF_MAX_STATE_DIM];
    double colUH[NB][BCBF_MAX_CTRL_DIM + 1];
};

// both halves' values of v: .x = the value held by lane (l & 31), .y = by lane (l & 31) + 32
__device__ inline double2 halves64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    double2 o;
    o.x = __builtin_bit_cast(double, ((long long)r1[0] << 32) | (long long)r0[0]);
    o.y = __builtin_bit_cast(double, ((long long)r1[1] << 32) | (long long)r0[1]);
    return o;
}
__device__ inline double rsqrt_nr2(double p) {      // hardware estimate + two Newton steps: full double precision
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);
    return y;
}

template <bool FROM_DENSE>
__global__ void __launch_bounds__(64 * RW_WPB, BCBF_RW64_OCC)
refit_wave64_kernel(const double* __restrict__ X, const double* __restrict__ UH, const double* __restrict__ Bm,
                    const double* __restrict__ ell, const double* __restrict__ s2p, const double* __restrict__ jitter,
                    const double* __restrict__ Kdense, double* __restrict__ Lop, double* __restrict__ UHBout,
                    double* __restrict__ Ldense, int* __restrict__ info, int Bt, int N, int Np, int n, int C) {
    constexpr int V = 2;
    __shared__ RWShared shm[RW_WPB];
    // wave-uniform instance index: the per-instance pointers and hyper-parameters then live in SGPRs (scalar loads)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x * RW_WPB + wave;
    if (b >= Bt) return;                                      // whole wave: no workgroup barrier anywhere below
    RWShared& sh = shm[wave];
    const int j16 = lane & 15, g = lane >> 4;                 // MFMA roles
    const int rr = lane & 31, hb = lane >> 5;                 // diagonal-tile roles

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
        for (int i = lane; i < N; i += 64)
            for (int c = 0; c < C; ++c) {
                double s = 0.0;
                for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                UHBout[((size_t)b * N + i) * C + c] = s;
            }
    }
    if (Ld)
        for (int e = lane; e < N * N; e += 64) { const int i = e / N, j = e - i * N; if (j > i) Ld[e] = 0.0; }

    int fail = 0;
    const int nblk = Np / NB;
#ifdef BCBF_RW64_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    RW_T0();
    for (int J = 0; J < nblk && fail == 0; ++J) {
        const int col0 = J * NB;
        if (!FROM_DENSE) {
            for (int e = lane; e < NB * n; e += 64) {
                const int c = e / n, d = e - c * n;
                sh.colX[c][d] = (col0 + c < N) ? Xb[(size_t)(col0 + c) * n + d] : 0.0;
            }
            for (int e = lane; e < NB * C; e += 64) {
                const int c = e / C, a = e - c * C;
                sh.colUH[c][a] = (col0 + c < N) ? UHb[(size_t)(col0 + c) * C + a] : 0.0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        RW_ACC(0);                                             // column staging
        double ainv[2][2][4];                                  // inv(L_JJ) as panel-solve A operands, loaded after the diagonal tile

        for (int I = J; I < nblk; ++I) {
            const int irow = I * NB + j16;                        // + 16 ib
            f64x4 acc[2][2];                                      // [cb][ib]:  S'[c = 16cb + 4r + g][i = 16ib + j16]
            // ---- initial value K_b'(c, i)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const int i = irow + 16 * ib;
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
                                if (d < n) { const double z = (xi[d] - sh.colX[c][d]) * iell[d]; d2 += z * z; }
#pragma unroll
                            for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a)
                                if (a < C) uu += ub[a] * sh.colUH[c][a];
                            val = s2 * exp(-0.5 * d2) * uu + (i == j ? jit : 0.0);
                        }
                        acc[cb][ib][r] = val;
                    }
            }
            RW_ACC(1);                                            // K_b values
            // ---- S' -= L_J L_I'  over all previous columns (software pipelined: next stage's operands in flight)
            constexpr int KS = BCBF_RW64_KS;
            double a_nxt[KS][2], b_nxt[KS][2];
            auto fetch = [&](int kk) {
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    const int base = lop_base<V>(kk + 4 * s_ + g, Np);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) a_nxt[s_][cb] = lop[base + col0 + 16 * cb + j16];
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) b_nxt[s_][ib] = lop[base + irow + 16 * ib];
                }
            };
            if (col0 > 0) fetch(0);
            for (int kk = 0; kk < col0; kk += 4 * KS) {
                double a_cur[KS][2], b_cur[KS][2];
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) a_cur[s_][cb] = -a_nxt[s_][cb];            // D = (-A) B + C
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) b_cur[s_][ib] = b_nxt[s_][ib];
                }
                if (kk + 4 * KS < col0) fetch(kk + 4 * KS);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib)
                            acc[cb][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[s_][cb], b_cur[s_][ib], acc[cb][ib], 0, 0, 0);
            }

            RW_ACC(2);                                            // update stream
            if (I == J) {
                // =================== the diagonal tile: factor L_JJ, invert it ===================
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.tile[16 * cb + 4 * r + g][16 * ib + j16] = acc[cb][ib][r];
                __builtin_amdgcn_wave_barrier();
                double S[16];                                      // S[rr][16 hb + q]  (= S'[16 hb + q][rr])
#pragma unroll
                for (int q = 0; q < 16; ++q) S[q] = sh.tile[16 * hb + q][rr];
                __builtin_amdgcn_wave_barrier();                   // the tile buffer is free again (it becomes L)
                int bad = 0;
#pragma unroll
                for (int s = 0; s < NB / 4; ++s) {
                    constexpr int dummy = 0; (void)dummy;
                    const int c0 = 4 * s, hbs = c0 / 16, q0 = c0 % 16;
                    // (1) the four raw columns c0..c0+3, every row
                    if (hb == hbs) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) sh.P[rr][a] = S[q0 + a];
                    }
                    __builtin_amdgcn_wave_barrier();
                    // (2) 4x4 diagonal block: Cholesky factor and inverse, redundantly in every lane
                    const double d00 = sh.P[c0][0];
                    const double d10 = sh.P[c0 + 1][0], d11 = sh.P[c0 + 1][1];
                    const double d20 = sh.P[c0 + 2][0], d21 = sh.P[c0 + 2][1], d22 = sh.P[c0 + 2][2];
                    const double d30 = sh.P[c0 + 3][0], d31 = sh.P[c0 + 3][1], d32 = sh.P[c0 + 3][2], d33 = sh.P[c0 + 3][3];
                    const double p0 = d00;
                    if (!(p0 > 0.0) && bad == 0) bad = col0 + c0 + 1;
                    const double r0 = rsqrt_nr2(p0 > 0.0 ? p0 : 1.0);
                    const double l00 = p0 * r0, l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
                    const double p1 = d11 - l10 * l10;
                    if (!(p1 > 0.0) && bad == 0) bad = col0 + c0 + 2;
                    const double r1 = rsqrt_nr2(p1 > 0.0 ? p1 : 1.0);
                    const double l11 = p1 * r1, l21 = (d21 - l20 * l10) * r1, l31 = (d31 - l30 * l10) * r1;
                    const double p2 = d22 - l20 * l20 - l21 * l21;
                    if (!(p2 > 0.0) && bad == 0) bad = col0 + c0 + 3;
                    const double r2 = rsqrt_nr2(p2 > 0.0 ? p2 : 1.0);
                    const double l22 = p2 * r2, l32 = (d32 - l30 * l20 - l31 * l21) * r2;
                    const double p3 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
                    if (!(p3 > 0.0) && bad == 0) bad = col0 + c0 + 4;
                    const double r3 = rsqrt_nr2(p3 > 0.0 ? p3 : 1.0);
                    const double l33 = p3 * r3;
                    // inverse of the 4x4 factor (lower): i_aa = 1 / l_aa
                    const double i00 = r0, i11 = r1, i22 = r2, i33 = r3;
                    const double i10 = -(l10 * i00) * r1;
                    const double i20 = -(l20 * i00 + l21 * i10) * r2, i21 = -(l21 * i11) * r2;
                    const double i30 = -(l30 * i00 + l31 * i10 + l32 * i20) * r3, i31 = -(l31 * i11 + l32 * i21) * r3,
                                 i32 = -(l32 * i22) * r3;
                    if (lane == 0) {
                        double* I4 = sh.I4[s];
                        I4[0] = i00; I4[1] = i10; I4[2] = i11; I4[3] = i20; I4[4] = i21; I4[5] = i22;
                        I4[6] = i30; I4[7] = i31; I4[8] = i32; I4[9] = i33;
                    }
                    // (3) this lane's row of the panel: l = P[rr][:] inv(L4)'   (rows of the block itself: L4; rows above: 0)
                    const double pr0 = sh.P[rr][0], pr1 = sh.P[rr][1], pr2 = sh.P[rr][2], pr3 = sh.P[rr][3];
                    double l0 = pr0 * i00;
                    double l1 = pr0 * i10 + pr1 * i11;
                    double l2 = pr0 * i20 + pr1 * i21 + pr2 * i22;
                    double l3 = pr0 * i30 + pr1 * i31 + pr2 * i32 + pr3 * i33;
                    const int ra = rr - c0;                          // row inside the block (0..3), negative above
                    if (ra == 0) { l0 = l00; l1 = 0.0; l2 = 0.0; l3 = 0.0; }
                    if (ra == 1) { l0 = l10; l1 = l11; l2 = 0.0; l3 = 0.0; }
                    if (ra == 2) { l0 = l20; l1 = l21; l2 = l22; l3 = 0.0; }
                    if (ra == 3) { l0 = l30; l1 = l31; l2 = l32; l3 = l33; }
                    if (ra < 0) { l0 = 0.0; l1 = 0.0; l2 = 0.0; l3 = 0.0; }
                    if (hb == 0) {
                        sh.Lp[rr][0] = l0; sh.Lp[rr][1] = l1; sh.Lp[rr][2] = l2; sh.Lp[rr][3] = l3;
                        sh.tile[rr][c0] = l0; sh.tile[rr][c0 + 1] = l1; sh.tile[rr][c0 + 2] = l2; sh.tile[rr][c0 + 3] = l3;
                    }
                    __builtin_amdgcn_wave_barrier();
                    // (4) rank-4 update of the lane's 16 elements S[rr][16 hb + q] -= l[rr] . l[16 hb + q]
                    //     (columns already factored receive garbage: they are never read again)
                    if (s < NB / 4 - 1) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            if (16 + q <= c0 + 3) continue;          // dead for both halves (compile time)
                            const double* lj = sh.Lp[16 * hb + q];
                            S[q] -= l0 * lj[0] + l1 * lj[1] + l2 * lj[2] + l3 * lj[3];
                            // pin the update HERE: LLVM otherwise sinks the multiply-adds to the step that publishes
                            // this column and keeps the four loaded operands alive instead of the one result (500+
                            // spilled registers)
                            asm volatile("" : "+v"(S[q]));
                            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep 16 operand loads in flight, not 64
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                RW_ACC(3);                                        // factor
                if (bad != 0 && bad <= N) fail = bad;
                if (Ld && lane < NB && col0 + lane < N) {
                    for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = sh.tile[lane][c];
                }
                // ---- X = inv(L): lane = column rr; half hb forms rows 2hb, 2hb+1 of each 4-row step
                {
                    double Xc[NB];                                  // X[k][rr], both halves hold all of it
#pragma unroll
                    for (int s = 0; s < NB / 4; ++s) {
                        const int c0 = 4 * s;
                        double pa = 0.0, pb = 0.0;                   // rows c0 + 2hb, c0 + 2hb + 1
                        const double* La = &sh.tile[c0 + 2 * hb][0];
                        const double* Lb = &sh.tile[c0 + 2 * hb + 1][0];
#pragma unroll
                        for (int k = 0; k < c0; ++k) {
                            pa -= La[k] * Xc[k];
                            pb -= Lb[k] * Xc[k];
                            if ((k & 7) == 7) { asm volatile("" : "+v"(pa), "+v"(pb)); __builtin_amdgcn_sched_barrier(0); }
                        }
                        const double2 ha = halves64(pa), hbv = halves64(pb);
                        const double e0 = (rr == c0 ? 1.0 : 0.0) + ha.x, e1 = (rr == c0 + 1 ? 1.0 : 0.0) + hbv.x;
                        const double e2 = (rr == c0 + 2 ? 1.0 : 0.0) + ha.y, e3 = (rr == c0 + 3 ? 1.0 : 0.0) + hbv.y;
                        const double* I4 = sh.I4[s];
                        Xc[c0] = I4[0] * e0;
                        Xc[c0 + 1] = I4[1] * e0 + I4[2] * e1;
                        Xc[c0 + 2] = I4[3] * e0 + I4[4] * e1 + I4[5] * e2;
                        Xc[c0 + 3] = I4[6] * e0 + I4[7] * e1 + I4[8] * e2 + I4[9] * e3;
                        asm volatile("" : "+v"(Xc[c0]), "+v"(Xc[c0 + 1]), "+v"(Xc[c0 + 2]), "+v"(Xc[c0 + 3]));   // as above
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_wave_barrier();                 // every lane has finished reading L
                    if (hb == 0) {
                        const int base = lop_dinv_block(J, Np) + lop_dinv_col(rr);       // lower triangle, packed
#pragma unroll
                        for (int k = 0; k < NB; ++k) {
                            const double xk = k >= rr ? Xc[k] : 0.0;                       // exact zeros above the diagonal
                            sh.tile[k][rr] = xk;
                            if (k >= rr) lop[base + k] = xk;
                            lop[lop_dfull(J, k, rr, Np)] = xk;
                        }
                    }
                    if (lane < LOP_DB - 528) lop[lop_dinv_block(J, Np) + 528 + lane] = 0.0;      // the block's padding
                    __builtin_amdgcn_wave_barrier();
                }
                if (fail != 0) break;
#pragma unroll
                for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                    for (int cb = 0; cb <= cbp; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ainv[cbp][cb][r] = sh.tile[16 * cbp + j16][16 * cb + 4 * r + g];
                RW_ACC(4);                                        // inverse + stores
            } else {
                // ---- panel:  L_IJ' = inv(L_JJ) S'   (accumulator registers of S' are the B operands)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = irow + 16 * ib;
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp) {
                        f64x4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int cb = 0; cb <= cbp; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                y = __builtin_amdgcn_mfma_f64_16x16x4f64(ainv[cbp][cb][r], acc[cb][ib][r], y, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * cbp + 4 * r + g;
                            lop[lop_base<V>(col0 + c, Np) + i] = y[r];
                            if (Ld && i < N && col0 + c < N) Ld[(size_t)i * N + col0 + c] = y[r];
                        }
                    }
                }
                RW_ACC(5);                                        // panel solve + stores
            }
        }
        __threadfence_block();                 // this block column's panels are read back (by other lanes) from here on
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) info[b] = fail;
#ifdef BCBF_RW64_PROF
    if (Ld && lane == 0 && N > 8)
        for (int k = 0; k < 6; ++k) Ld[1 + k] = (double)prof[k];
#endif
}

// Called by bcbf_refit_mfma_f64 for batches: returns 0 after launching, or -1 when the shape is not taken here.
int launch_refit_wave64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                        const double* jitter, const double* Kdense, double* Lop, double* UHB, double* Ldense, int* info,
                        int Bt, int N, int Np, int n, int C, hipStream_t st) {
    const dim3 grid((Bt + RW_WPB - 1) / RW_WPB), block(64 * RW_WPB);
    if (Kdense)
        hipLaunchKernelGGL((refit_wave64_kernel<true>), grid, block, 0, st, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, Kdense, Lop, nullptr, Ldense, info, Bt, N, Np, 0, 0);
    else
        hipLaunchKernelGGL((refit_wave64_kernel<false>), grid, block, 0, st, X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB,
                           Ldense, info, Bt, N, Np, n, C);
    return 0;
}

}  // namespace bcbf
