// The 32x32 diagonal tile of the blocked Cholesky (refit_wave64.hip; fp64 and fp32): factor S = L L' and
// invert L, by ONE wave, out of LDS -- built around 4-column blocks so that the serial part of a step is four rsqrt
// chains instead of 32 dependent dot products:
//
//   factor   lane = (row rr = lane & 31, column half hb = lane >> 5) holds S[rr][16hb .. 16hb+15] in registers.  Step s
//            (columns 4s..4s+3): publish the four raw columns -> every lane factors AND inverts the 4x4 diagonal block
//            redundantly -> lane rr forms its panel row l[rr][0..3] = P[rr][:] inv(L4)' -> publish -> rank-4 update of
//            the lane's 16 elements.
//   inverse  lane = column rr; 8 block steps; the halves each form two of the four dot products of a step and swap
//            them (v_permlane32_swap), then multiply by the 4x4 inverse kept from the factor.  X lives in LDS.
//
// ~40 k cycles per tile measured in the kernel (the 32-step left-looking form it replaces: 65-77 k).
#pragma once
#include "bcbf_common.h"
#ifndef BCBF_DT_STAMP
#define BCBF_DT_STAMP(k) do {} while (0)      // development: refit_wave64.hip's -DBCBF_RP_TRACE time stamps
#endif

namespace bcbf {

constexpr int DT_LS = NB + 1;            // padded row stride of the LDS tiles

template <typename T> struct DiagTile {
    T tile[NB][DT_LS];                        // in: S' (tile[c][i] = S[i][c]);  out: L, row-major [r][c], zeros above
    T xinv[NB][DT_LS];                        // out: X = inv(L) [row][col], zeros above the diagonal
    T P[NB][4];                               // raw panel columns of the current 4-column step
    T Lp[NB][4];                              // solved panel rows l[r][0..3]
    T I4[NB / 4][12];                         // the eight 4x4 inverses (10 used)
};
using DiagTile64 = DiagTile<double>;

// both halves' values of v: .x = the value held by lane (l & 31), .y = by lane (l & 31) + 32
__device__ inline double2 halves64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
    unsigned lo2 = lo, hi2 = hi;
    asm volatile("" : "+v"(lo2), "+v"(hi2));  // distinct registers for the two in-place operands (see the float form)
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo2, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi2, false, false);
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
__device__ inline float rsqrt_nr2(float p) {        // hardware estimate (1 ulp) + one Newton step
    float y = __builtin_amdgcn_rsqf(p);
    y = y * (1.5f - 0.5f * p * y * y);
    return y;
}
__device__ inline float2 halves64(float v) {
    // (the builtin form of this swap, fine in the double overload above, came back with BOTH results equal to the lower
    // half's value here -- ROCm 7.2 clang, tools/dev/micro/halves_test.hip -- so the instruction is written out; the s_nop
    // covers the VALU-write -> permlane read hazard the compiler cannot see inside the asm)
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float2 o;
    o.x = __builtin_bit_cast(float, a);      // lanes 0-31 keep theirs, lanes 32-63 receive the lower half's
    o.y = __builtin_bit_cast(float, b);      // lanes 0-31 receive the upper half's, lanes 32-63 keep theirs
    return o;
}

// LDS-qualified views: a plain pointer to a __shared__ object is a GENERIC pointer once it crosses a function boundary
// or a dynamic index, and the accesses compile to flat_load / flat_store instead of ds_read / ds_write (measured:
// +27 % kernel time)
template <typename T> using DiagTileLds = __attribute__((address_space(3))) DiagTile<T>;
template <typename T> using LdsElem = __attribute__((address_space(3))) T;
using DiagTile64Lds = DiagTileLds<double>;
using LdsDouble = LdsElem<double>;
#define BCBF_LDS_TILE(T, obj) ((bcbf::DiagTileLds<T>*)&(obj))

// Whole wave (64 lanes) calls this with the tile's S' already in sh->tile (and visible: wave barrier done by the
// caller).  Returns 0, or 1 + the index (0..31) of the first non-positive pivot (the outputs are then garbage).
template <typename T> struct Vec2Of;
template <> struct Vec2Of<double> { using type = double2; };
template <> struct Vec2Of<float> { using type = float2; };

template <typename T> __device__ inline int diag_factor_invert(DiagTileLds<T>* sh, int lane) {
    using T2 = typename Vec2Of<T>::type;
    const int rr = lane & 31, hb = lane >> 5;
    T S[16];                                      // S[rr][16 hb + q]  (= S'[16 hb + q][rr])
#pragma unroll
    for (int q = 0; q < 16; ++q) S[q] = sh->tile[16 * hb + q][rr];
    __builtin_amdgcn_wave_barrier();                   // the tile buffer is free again (it becomes L)
    int bad = 0;
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) {
        constexpr int dummy = 0; (void)dummy;
        const int c0 = 4 * s, hbs = c0 / 16, q0 = c0 % 16;
        // (1) the four raw columns c0..c0+3, every row
        if (hb == hbs) {
#pragma unroll
            for (int a = 0; a < 4; ++a) sh->P[rr][a] = S[q0 + a];
        }
        __builtin_amdgcn_wave_barrier();
        BCBF_DT_STAMP(0);
        // (2) 4x4 diagonal block: Cholesky factor and inverse, redundantly in every lane
        const T d00 = sh->P[c0][0];
        const T d10 = sh->P[c0 + 1][0], d11 = sh->P[c0 + 1][1];
        const T d20 = sh->P[c0 + 2][0], d21 = sh->P[c0 + 2][1], d22 = sh->P[c0 + 2][2];
        const T d30 = sh->P[c0 + 3][0], d31 = sh->P[c0 + 3][1], d32 = sh->P[c0 + 3][2], d33 = sh->P[c0 + 3][3];
        const T p0 = d00;
        if (!(p0 > T(0.0)) && bad == 0) bad = c0 + 1;
        const T r0 = rsqrt_nr2(p0 > T(0.0) ? p0 : T(1.0));
        const T l00 = p0 * r0, l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
        const T p1 = d11 - l10 * l10;
        if (!(p1 > T(0.0)) && bad == 0) bad = c0 + 2;
        const T r1 = rsqrt_nr2(p1 > T(0.0) ? p1 : T(1.0));
        const T l11 = p1 * r1, l21 = (d21 - l20 * l10) * r1, l31 = (d31 - l30 * l10) * r1;
        const T p2 = d22 - l20 * l20 - l21 * l21;
        if (!(p2 > T(0.0)) && bad == 0) bad = c0 + 3;
        const T r2 = rsqrt_nr2(p2 > T(0.0) ? p2 : T(1.0));
        const T l22 = p2 * r2, l32 = (d32 - l30 * l20 - l31 * l21) * r2;
        const T p3 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
        if (!(p3 > T(0.0)) && bad == 0) bad = c0 + 4;
        const T r3 = rsqrt_nr2(p3 > T(0.0) ? p3 : T(1.0));
        const T l33 = p3 * r3;
        // inverse of the 4x4 factor (lower): i_aa = 1 / l_aa
        const T i00 = r0, i11 = r1, i22 = r2, i33 = r3;
        const T i10 = -(l10 * i00) * r1;
        const T i20 = -(l20 * i00 + l21 * i10) * r2, i21 = -(l21 * i11) * r2;
        const T i30 = -(l30 * i00 + l31 * i10 + l32 * i20) * r3, i31 = -(l31 * i11 + l32 * i21) * r3,
                     i32 = -(l32 * i22) * r3;
        if (lane == 0) {
            LdsElem<T>* I4 = sh->I4[s];
            I4[0] = i00; I4[1] = i10; I4[2] = i11; I4[3] = i20; I4[4] = i21; I4[5] = i22;
            I4[6] = i30; I4[7] = i31; I4[8] = i32; I4[9] = i33;
        }
        BCBF_DT_STAMP(1);
        // (3) this lane's row of the panel: l = P[rr][:] inv(L4)'   (rows of the block itself: L4; rows above: 0)
        const T pr0 = sh->P[rr][0], pr1 = sh->P[rr][1], pr2 = sh->P[rr][2], pr3 = sh->P[rr][3];
        T l0 = pr0 * i00;
        T l1 = pr0 * i10 + pr1 * i11;
        T l2 = pr0 * i20 + pr1 * i21 + pr2 * i22;
        T l3 = pr0 * i30 + pr1 * i31 + pr2 * i32 + pr3 * i33;
        const int ra = rr - c0;                          // row inside the block (0..3), negative above
        if (ra == 0) { l0 = l00; l1 = T(0.0); l2 = T(0.0); l3 = T(0.0); }
        if (ra == 1) { l0 = l10; l1 = l11; l2 = T(0.0); l3 = T(0.0); }
        if (ra == 2) { l0 = l20; l1 = l21; l2 = l22; l3 = T(0.0); }
        if (ra == 3) { l0 = l30; l1 = l31; l2 = l32; l3 = l33; }
        if (ra < 0) { l0 = T(0.0); l1 = T(0.0); l2 = T(0.0); l3 = T(0.0); }
        if (hb == 0) {
            sh->Lp[rr][0] = l0; sh->Lp[rr][1] = l1; sh->Lp[rr][2] = l2; sh->Lp[rr][3] = l3;
            sh->tile[rr][c0] = l0; sh->tile[rr][c0 + 1] = l1; sh->tile[rr][c0 + 2] = l2; sh->tile[rr][c0 + 3] = l3;
        }
        __builtin_amdgcn_wave_barrier();
        BCBF_DT_STAMP(2);
        // (4) rank-4 update of the lane's 16 elements S[rr][16 hb + q] -= l[rr] . l[16 hb + q]
        //     (columns already factored receive garbage: they are never read again)
        if (s < NB / 4 - 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (16 + q <= c0 + 3) continue;          // dead for both halves (compile time)
                const LdsElem<T>* lj = sh->Lp[16 * hb + q];
                S[q] -= l0 * lj[0] + l1 * lj[1] + l2 * lj[2] + l3 * lj[3];
                // pin the update HERE: LLVM otherwise sinks the multiply-adds to the step that publishes
                // this column and keeps the four loaded operands alive instead of the one result (500+
                // spilled registers)
                asm volatile("" : "+v"(S[q]));
                if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep 16 operand loads in flight, not 64
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        BCBF_DT_STAMP(3);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- X = inv(L): lane = column rr; half hb forms rows 2hb, 2hb+1 of each 4-row step.  X lives in LDS
    //      (xinv[k][rr]): rolled loops, a handful of registers (the register-resident form needed 200 more)
    {
#pragma unroll 1
        for (int s = 0; s < NB / 4; ++s) {
            const int c0 = 4 * s;
            T pa = T(0.0), pb = T(0.0);                   // rows c0 + 2hb, c0 + 2hb + 1
            const LdsElem<T>* La = &sh->tile[c0 + 2 * hb][0];
            const LdsElem<T>* Lb = La + DT_LS;
#pragma unroll 4
            for (int k = 0; k < c0; ++k) {
                const T xk = sh->xinv[k][rr];
                pa -= La[k] * xk;
                pb -= Lb[k] * xk;
            }
            const T2 ha = halves64(pa), hbv = halves64(pb);
            const T e0 = (rr == c0 ? T(1.0) : T(0.0)) + ha.x, e1 = (rr == c0 + 1 ? T(1.0) : T(0.0)) + hbv.x;
            const T e2 = (rr == c0 + 2 ? T(1.0) : T(0.0)) + ha.y, e3 = (rr == c0 + 3 ? T(1.0) : T(0.0)) + hbv.y;
            const LdsElem<T>* I4 = sh->I4[s];
            if (hb == 0) {                                // exact zeros above the diagonal (rows < column)
                sh->xinv[c0][rr] = c0 >= rr ? I4[0] * e0 : T(0.0);
                sh->xinv[c0 + 1][rr] = c0 + 1 >= rr ? I4[1] * e0 + I4[2] * e1 : T(0.0);
                sh->xinv[c0 + 2][rr] = c0 + 2 >= rr ? I4[3] * e0 + I4[4] * e1 + I4[5] * e2 : T(0.0);
                sh->xinv[c0 + 3][rr] = c0 + 3 >= rr ? I4[6] * e0 + I4[7] * e1 + I4[8] * e2 + I4[9] * e3 : T(0.0);
            }
            __builtin_amdgcn_wave_barrier();
            BCBF_DT_STAMP(4);
        }
    }
    return bad;
}


// ---- the same tile out of MFMA ACCUMULATOR registers, the rank-4 updates on the matrix cores (round 3) ----
// S arrives the way the update stream of refit_wave64.hip leaves it: S[cb][ib][r] in lane (j16 = lane & 15, g = lane >> 4) is
// the element (p = 2 midx(r, g) + cb, q = 2 j16 + ib) of the symmetric tile (midx: the accumulator row of register r in
// lane group g, g + 4r in fp64, 4g + r in fp32); the entries with p <= q are the ones read.  Step s (columns c0 = 4s ..):
//   publish  rows c0..c0+3 of S sit in ONE register index of two (fp64) / one (fp32) lane group(s): 2 values per lane and
//            column -> P[q][a]
//   pivots   every lane factors and inverts the 4x4 diagonal block redundantly (as above)
//   panel    lane (j16, g) forms l(q, c0 + g) for its two q from P[q][0..3] and row g of the 4x4 inverse -- exactly the A
//            (row 2 j16 + cb) and B (row 2 j16 + ib) operands of  S -= l l'  as FOUR 16x16x4 MFMAs: no second trip
//            through LDS, no 64 multiply-adds per lane
//   inverse  W (starts as the identity, same layout) takes the same eliminations: rows c0..c0+3 of W are published like
//            S's, X(c0 + g, q) = (row g of the 4x4 inverse) . W(c0.., q) is a row of inv(L) AND the B operand of
//            W -= l X: four more MFMAs, off the pivots' critical path.  The serial 8-step inverse pass above is gone.
// Outputs as above (sh->xinv = inv(L) [row][col] with zeros above the diagonal; sh->tile = L likewise when want_l (wave-uniform) -- the
// packed operator stores only the inverse of a diagonal tile).
template <typename T> __device__ inline int dt_midx(int r, int g) { return sizeof(T) == 8 ? g + 4 * r : 4 * g + r; }
__device__ inline __attribute__((__vector_size__(4 * sizeof(double)))) double
dt_mfma(double a, double b, __attribute__((__vector_size__(4 * sizeof(double)))) double c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ inline __attribute__((__vector_size__(4 * sizeof(float)))) float
dt_mfma(float a, float b, __attribute__((__vector_size__(4 * sizeof(float)))) float c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <typename T, typename ACC>
__device__ inline int diag_factor_invert_acc(DiagTileLds<T>* sh, ACC (&S)[2][2], int lane, bool want_l) {
    const int j16 = lane & 15, g = lane >> 4;
    ACC W[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) W[cb][ib][r] = (2 * dt_midx<T>(r, g) + cb == 2 * j16 + ib) ? T(1.0) : T(0.0);
    int bad = 0;
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) {
        const int c0 = 4 * s;
        // (1) rows c0..c0+3 of S and of W -> P / Lp  ([q][a])
        if (sizeof(T) == 8) {
            const int r0 = s >> 1, ga = (2 * s) & 3;
            if (g == ga || g == ga + 1) {
                const int a0 = g == ga ? 0 : 2;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    sh->P[2 * j16 + ib][a0] = S[0][ib][r0];  sh->P[2 * j16 + ib][a0 + 1] = S[1][ib][r0];
                    sh->Lp[2 * j16 + ib][a0] = W[0][ib][r0]; sh->Lp[2 * j16 + ib][a0 + 1] = W[1][ib][r0];
                }
            }
        } else {
            const int r0 = (2 * s) & 3;
            if (g == (s >> 1)) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    sh->P[2 * j16 + ib][0] = S[0][ib][r0];      sh->P[2 * j16 + ib][1] = S[1][ib][r0];
                    sh->P[2 * j16 + ib][2] = S[0][ib][r0 + 1];  sh->P[2 * j16 + ib][3] = S[1][ib][r0 + 1];
                    sh->Lp[2 * j16 + ib][0] = W[0][ib][r0];     sh->Lp[2 * j16 + ib][1] = W[1][ib][r0];
                    sh->Lp[2 * j16 + ib][2] = W[0][ib][r0 + 1]; sh->Lp[2 * j16 + ib][3] = W[1][ib][r0 + 1];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        BCBF_DT_STAMP(0);
        // (2) 4x4 diagonal block: Cholesky factor and inverse, redundantly in every lane
        const T d00 = sh->P[c0][0];
        const T d10 = sh->P[c0 + 1][0], d11 = sh->P[c0 + 1][1];
        const T d20 = sh->P[c0 + 2][0], d21 = sh->P[c0 + 2][1], d22 = sh->P[c0 + 2][2];
        const T d30 = sh->P[c0 + 3][0], d31 = sh->P[c0 + 3][1], d32 = sh->P[c0 + 3][2], d33 = sh->P[c0 + 3][3];
        const T p0 = d00;
        if (!(p0 > T(0.0)) && bad == 0) bad = c0 + 1;
        const T r0_ = rsqrt_nr2(p0 > T(0.0) ? p0 : T(1.0));
        const T l00 = p0 * r0_, l10 = d10 * r0_, l20 = d20 * r0_, l30 = d30 * r0_;
        const T p1 = d11 - l10 * l10;
        if (!(p1 > T(0.0)) && bad == 0) bad = c0 + 2;
        const T r1_ = rsqrt_nr2(p1 > T(0.0) ? p1 : T(1.0));
        const T l11 = p1 * r1_, l21 = (d21 - l20 * l10) * r1_, l31 = (d31 - l30 * l10) * r1_;
        const T p2 = d22 - l20 * l20 - l21 * l21;
        if (!(p2 > T(0.0)) && bad == 0) bad = c0 + 3;
        const T r2_ = rsqrt_nr2(p2 > T(0.0) ? p2 : T(1.0));
        const T l22 = p2 * r2_, l32 = (d32 - l30 * l20 - l31 * l21) * r2_;
        const T p3 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
        if (!(p3 > T(0.0)) && bad == 0) bad = c0 + 4;
        const T r3_ = rsqrt_nr2(p3 > T(0.0) ? p3 : T(1.0));
        const T l33 = p3 * r3_;
        const T i00 = r0_, i11 = r1_, i22 = r2_, i33 = r3_;
        const T i10 = -(l10 * i00) * r1_;
        const T i20 = -(l20 * i00 + l21 * i10) * r2_, i21 = -(l21 * i11) * r2_;
        const T i30 = -(l30 * i00 + l31 * i10 + l32 * i20) * r3_, i31 = -(l31 * i11 + l32 * i21) * r3_,
                     i32 = -(l32 * i22) * r3_;
        BCBF_DT_STAMP(1);
        // (3) row g of the 4x4 inverse / column g of the 4x4 factor, then this lane's two panel entries and inverse entries
        const T z = T(0.0);
        const T k0 = g == 0 ? i00 : g == 1 ? i10 : g == 2 ? i20 : i30;
        const T k1 = g == 0 ? z : g == 1 ? i11 : g == 2 ? i21 : i31;
        const T k2 = g < 2 ? z : g == 2 ? i22 : i32;
        const T k3 = g == 3 ? i33 : z;
        const T f0 = g == 0 ? l00 : z;
        const T f1 = g == 0 ? l10 : g == 1 ? l11 : z;
        const T f2 = g == 0 ? l20 : g == 1 ? l21 : g == 2 ? l22 : z;
        const T f3 = g == 0 ? l30 : g == 1 ? l31 : g == 2 ? l32 : l33;
        // (this lane's rows of both publications: read HERE, behind the pivots -- 16 more live values across the four
        //  rsqrt chains spill at 256 registers; the LDS round trip this exposes is a fifth of a microsecond per tile)
        __builtin_amdgcn_sched_barrier(0);
        T pr[2][4], pw[2][4];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int a = 0; a < 4; ++a) { pr[ib][a] = sh->P[2 * j16 + ib][a]; pw[ib][a] = sh->Lp[2 * j16 + ib][a]; }
        T la[2], xb[2];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int qa = 2 * j16 + ib - c0;              // row inside the block (0..3), negative above
            T v = k0 * pr[ib][0] + k1 * pr[ib][1] + k2 * pr[ib][2] + k3 * pr[ib][3];
            v = qa < 0 ? z : qa == 0 ? f0 : qa == 1 ? f1 : qa == 2 ? f2 : qa == 3 ? f3 : v;
            la[ib] = v;
            xb[ib] = k0 * pw[ib][0] + k1 * pw[ib][1] + k2 * pw[ib][2] + k3 * pw[ib][3];
            if (want_l) sh->tile[2 * j16 + ib][c0 + g] = v;
            sh->xinv[c0 + g][2 * j16 + ib] = xb[ib];
        }
        BCBF_DT_STAMP(2);
        // (4) S -= l l',  W -= l X  on the matrix cores
        if (s < NB / 4 - 1) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) S[cb][ib] = dt_mfma(-la[cb], la[ib], S[cb][ib]);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) W[cb][ib] = dt_mfma(-la[cb], xb[ib], W[cb][ib]);
        }
        BCBF_DT_STAMP(3);
    }
    __builtin_amdgcn_wave_barrier();
    return bad;
}

}  // namespace bcbf
