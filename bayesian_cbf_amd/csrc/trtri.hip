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
#include <type_traits>
#include <stdlib.h>

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
        // operands of block K + 1 are in flight while block K's MFMA chain runs (two register sets, swapped by the unrolled pair)
        float av[2][16], bv[2][16];
        auto load = [&](int K, float (&a)[16], float (&bq)[16]) {
            const int stride = Np - NB * (K + 1);
            const float* pa = lop + (lop_base<V>(K * NB, Np) + NB * (K + 1)) + (I - K - 1) * NB;      // (scalar base + 32-bit lane offsets: see the fp64 kernel)
            const float* pb = X + (size_t)K * NB * N + J * NB;
            const bool krow = (K + 1) * NB <= N;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = 2 * s + g;                           // column of L / row of X within block K
                a[s] = pa[__umul24(k, stride) + col];
                bq[s] = (vc && (krow || K * NB + k < N)) ? pb[__umul24(k, N) + col] : 0.0f;
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
        // addresses: wave-uniform bases (scalar registers) + 32-bit per-lane offsets.  Column c = 4 s + g of block K starts at lop_base(32 K) + c (Np - 32 (K + 1)):
        // one 24-bit multiply-add per s and K instead of the layout's full index arithmetic per load (20 vector instructions each, measured in the fp32 kernel)
        auto load = [&](int K, double (&a)[8][2], double (&bq)[8][2]) {
            const int stride = Np - NB * (K + 1);
            const double* pa = lop + (lop_base<V>(K * NB, Np) + NB * (K + 1)) + (I - K - 1) * NB;     // row 32 I of column 32 K (>= 0: the column stores rows >= 32 (K + 1))
            const double* pb = X + (size_t)K * NB * N + J * NB;
            const bool krow = (K + 1) * NB <= N;                  // (a block row clear of the padding: no per-row test)
#pragma unroll
            for (int s = 0; s < 8; ++s) {                          // k = 4 s + g within the 32-wide tile
                const unsigned ao = __umul24(4 * s + g, stride) + c16;
                const unsigned bo = __umul24(4 * s + g, N) + c16;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    a[s][h] = pa[ao + 16u * h];
                    const int gc = J * NB + 16 * h + c16;
                    bq[s][h] = (gc < N && (krow || K * NB + 4 * s + g < N)) ? pb[bo + 16u * h] : 0.0;
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

// RIGHT-LOOKING within a block column, N <= 512 (round 6): the form above reads every X_KJ back from the dense output it has just written -- a store, a fence and
// an exposed load round trip per block row, at ~220 registers (two waves per SIMD): 13 TFLOP/s.  In the forms below a wave keeps the accumulators of ALL the tiles
// under its columns in registers: as soon as X_KJ exists -- in the accumulator layout, which IS the B-operand layout -- it is applied to every pending row I > K,
// `acc_I += L_IK X_KJ`.  Nothing the wave loads depends on anything it computed: no read-back, no fence.  (Steps on the way, DESIGN 3.4 / DESIGN_NOTES: operands
// gathered straight from memory into a register ring -- drained per tile by the compiler's waits across basic blocks, 9.2 ms fp32 at 4096 x 512; a whole block
// column per wave with the LDS ring below, 4.2 ms; the pair form, 3.4 ms.)
constexpr int TR_MAXT = 15;

// compile-time loop: every index of the accumulator array is a constant, so the array lives in registers (with a run-time `break` in an unrolled loop the
// compiler kept it in scratch: 1 KB per lane)
template <int B, int E, typename F> __device__ inline void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// The operand stream goes through LDS by `buffer_load ... lds`.  Nothing a wave reads depends on what it computes, so the WHOLE read sequence is known up front:
// for K = J ..: [inv(L_KK) (K > J)], L_{K+1,K}, L_{K+2,K}, ...  -- every element a 32 x 32 tile.  A run-time cursor walks that sequence RING - 1 elements ahead
// of the (compile-time unrolled) consumption and copies each tile into a ring of LDS slots (no registers, no s_waitcnt the compiler places); a consumer step issues
// the next element's copy, waits with a COUNTED vmcnt for its own element (the younger copies stay in flight; past the end of the sequence the cursor issues
// out-of-range dummies so that the count holds), reads its A-operand registers from the slot and runs the MFMA chain.  The inverted diagonal tile is just another
// element of the stream (its full-tile copy is column-major like an off-diagonal tile's LDS image: one fragment addressing for both).  Workgroup -> (model, block
// column) is XCD-aware: the block columns of a model are dealt to ONE XCD (consecutive workgroups of that XCD), whose L2 then serves the K + 1 reads of tile (I, K).
// A wave holds the accumulators of HALF a block column (16 columns: 15 tiles x 16 registers in fp64, x 8 in fp32), so the two waves of a block column form ONE
// workgroup and share the ring -- each copies half of an element's eight 1 KB pieces, a workgroup barrier per element says that both halves have landed and
// that the slot about to be refilled has been read by both (8 KB tiles: the stream crosses the fabric once per block column, not once per wave).
constexpr int TD_RING64 = 8;
// (the same kernel serves fp32 -- `trtri_pair_kernel<float>` -- with the 16x16x4 fp32 MFMA: half a block column is then 120 accumulator registers and TWO waves share a
//  SIMD, which the copy path rewards: see launch_trtri_mfma_f32)
template <typename T> struct TP;
template <> struct TP<double> {
    using acc_t = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    static constexpr int OCC = 1, PIECES = 8, LQS = 4, RING = TD_RING64;      // 1 KB pieces per tile; lane = (column lane >> LQS, 16 bytes of rows) of a piece
    __device__ static acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static int row(int q, int g) { return 4 * q + g; }  // row (of 16) that accumulator register q holds in lane group g
};
template <> struct TP<float> {
    using acc_t = __attribute__((__vector_size__(4 * sizeof(float)))) float;
#ifndef BCBF_TP32_OCC
#define BCBF_TP32_OCC 3          // fp32 waves per SIMD (measured 4096 x 512, (waves, ring slots)): (2, 8) 3.56 ms, (2, 10) 3.60, (3, 6) 3.39 with 80 B of scratch -- the copy path rewards waves, not depth
#endif
#ifndef BCBF_TP32_RING
#define BCBF_TP32_RING 6
#endif
    static constexpr int OCC = BCBF_TP32_OCC, PIECES = 4, LQS = 3, RING = BCBF_TP32_RING;
    __device__ static acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static int row(int q, int g) { return 4 * g + q; }
};
template <typename T>
__global__ void __launch_bounds__(128, TP<T>::OCC)
trtri_pair_kernel(const T* __restrict__ Lop, T* __restrict__ Linv, int Bt, int N, int Np, int nblk) {
    using P = TP<T>;
    using f64x4 = typename P::acc_t;
    constexpr int V = Vec<T>::V, ES = (int)sizeof(T), RPL = 16 / ES;       // rows per lane and copy
    __shared__ __attribute__((aligned(16))) T ring[P::RING][NB * NB];
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int J = pos % nblk, b = (pos / nblk) * 8 + xcd;
    if (b >= Bt) return;                                           // (both waves)
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    T* __restrict__ X = Linv + (size_t)b * N * N;
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    const int lc = 16 * half + c16, gc = J * NB + lc;              // this lane's column (within the block column / global)
    const bool vc = gc < N;
    const int cnt = nblk - J - 1;
    for (int i = g; i < J * NB; i += 4)
        if (vc && i < N) X[(size_t)i * N + gc] = T(0.0);
    int rr[2][4];
    unsigned xoff[2][4];
#pragma unroll
    for (int hk = 0; hk < 2; ++hk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rr[hk][q] = 16 * hk + P::row(q, g);
            xoff[hk][q] = (unsigned)(rr[hk][q] * N + gc);
        }
    f64x4 xk[2], acc[TR_MAXT][2];
    {
        const T* dj = lop + lop_dfull_block(J, Np);
        T* xj = X + (size_t)J * NB * N;
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xk[hk][q] = dj[(unsigned)(NB * lc + rr[hk][q])];
                if (vc && J * NB + rr[hk][q] < N) xj[xoff[hk][q]] = xk[hk][q];
            }
    }
    static_for<0, TR_MAXT>([&](auto tc) { acc[decltype(tc)::value][0] = f64x4{0}; acc[decltype(tc)::value][1] = f64x4{0}; });

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(lop), 0, (unsigned)min((size_t)0xfffffff0u, lop_elems<V>(Np) * ES), 0x00020000);
    int cK = J, cI = J + 1, wslot = 0;
    if (cI >= nblk) { cK = J + 1; cI = cK; }
    const int lq = lane >> P::LQS, lr = lane & ((1 << P::LQS) - 1);      // a copy instruction moves 4 (fp32: 8) columns x 32 rows: lane = (column lq, rows RPL lr ..)
    auto issue_next = [&]() {                                      // this wave's four pieces (half, half + 2, ..) of the next element
        const bool valid = cK < nblk, diag = cI == cK;
        const int cs = Np - NB * (cK + 1);
        const int voff = !valid ? 0x7ffffff0 : diag ? 16 * lane : (lq * cs + RPL * lr) * ES;
        const int soff = diag ? lop_dfull_block(cK, Np) * ES : (lop_base<V>(cK * NB, Np) + NB * (cK + 1) + NB * (cI - cK - 1)) * ES;
        const int step = diag ? 1024 : 32 * cs;                   // (bytes between pieces: 1 KB of the contiguous diagonal tile / 4 (fp32: 8) columns)
        __attribute__((address_space(3))) T* dst = (__attribute__((address_space(3))) T*)&ring[wslot][0];
#pragma unroll
        for (int j4 = 0; j4 < P::PIECES / 2; ++j4) {
            const int j = half + 2 * j4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (1024 / ES) * j), 16, voff, valid ? soff + j * step : 0, 0, 0);
        }
        wslot = wslot + 1 == P::RING ? 0 : wslot + 1;
        ++cI;
        if (cI >= nblk) { ++cK; cI = cK; }
    };
    for (int a = 0; a < P::RING - 1; ++a) issue_next();
    int rslot = 0;
    // the next element as A-operand registers: fr[hk][q][hi] = tile[row 16 hi + c16][column rr[hk][q]]  (the accumulator's contraction order)
    auto next_frags = [&](T (&fr)[2][4][2]) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((P::PIECES / 2) * (P::RING - 2)) : "memory");      // this wave's pieces of the element have landed
        __builtin_amdgcn_s_barrier();                                                    // ... the other wave's too; the slot refilled next has been read by both
        asm volatile("" ::: "memory");
        issue_next();
        const __attribute__((address_space(3))) T* src = (const __attribute__((address_space(3))) T*)&ring[rslot][0] + P::row(0, g) * NB + c16;
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int hi = 0; hi < 2; ++hi) fr[hk][q][hi] = src[(16 * hk + P::row(q, 0)) * NB + 16 * hi];
        rslot = rslot + 1 == P::RING ? 0 : rslot + 1;
    };
    // (software pipeline as in the fp32 kernel: a consumer fetches its successor's element between its own MFMAs)
    T cur[2][4][2];
    next_frags(cur);
    auto copy_frags = [&](T (&d)[2][4][2], const T (&s_)[2][4][2]) {
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int hi = 0; hi < 2; ++hi) d[hk][q][hi] = s_[hk][q][hi];
    };
    static_for<0, TR_MAXT + 1>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        if (kk > cnt) return;
        const int K = J + kk;
        if constexpr (kk > 0) {
            T nxt[2][4][2];
            f64x4 out[2] = {f64x4{0}, f64x4{0}};
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) out[hi] = P::mfma(cur[0][0][hi], acc[kk - 1][0][0], out[hi]);
            __builtin_amdgcn_sched_barrier(0);
            next_frags(nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hk = 0; hk < 2; ++hk)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int hi = 0; hi < 2; ++hi)
                        if (hk + q > 0) out[hi] = P::mfma(cur[hk][q][hi], acc[kk - 1][hk][q], out[hi]);
            copy_frags(cur, nxt);
            T* xkp = X + (size_t)K * NB * N;
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xk[hi][q] = -out[hi][q];
                    if (vc && K * NB + rr[hi][q] < N) xkp[xoff[hi][q]] = xk[hi][q];
                }
        }
        static_for<0, TR_MAXT - kk>([&](auto pc) {
            constexpr int t = kk + decltype(pc)::value;
            if (t < cnt) {
                T nxt[2][4][2];
#pragma unroll
                for (int hi = 0; hi < 2; ++hi) acc[t][hi] = P::mfma(cur[0][0][hi], xk[0][0], acc[t][hi]);
                __builtin_amdgcn_sched_barrier(0);
                next_frags(nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int hk = 0; hk < 2; ++hk)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int hi = 0; hi < 2; ++hi)
                            if (hk + q > 0) acc[t][hi] = P::mfma(cur[hk][q][hi], xk[hk][q], acc[t][hi]);
                copy_frags(cur, nxt);
            }
        });
    });
}

// N <= 512: the pair form with the LDS ring (BCBF_TRTRI_DMA=0: the left-looking form, which every larger N takes)
template <typename T>
static int launch_trtri_mfma(const T* Lop, T* Linv, int Bt, int N, void* stream) {
    const int Np = round_up(N, NB), nblk = Np / NB;
    if ((long long)Bt * nblk > 0x7fffffffLL) return BCBF_EINVAL;
    static const int dma = [] { const char* e = getenv("BCBF_TRTRI_DMA"); return e ? atoi(e) : 1; }();         // (development)
    if (nblk - 1 <= TR_MAXT && dma) {
        const long long groups = ((long long)Bt + 7) / 8;
        if (groups * 8 * nblk > 0x7fffffffLL) return BCBF_EINVAL;
        hipLaunchKernelGGL(trtri_pair_kernel<T>, dim3((unsigned)(groups * 8 * nblk)), dim3(128), 0, (hipStream_t)stream, Lop, Linv, Bt, N, Np, nblk);
        return check_launch("trtri_pair");
    }
    if constexpr (sizeof(T) == 4) hipLaunchKernelGGL(trtri_mfma_kernel_f32, dim3(Bt * nblk), dim3(64), 0, (hipStream_t)stream, Lop, Linv, N, Np, nblk);
    else hipLaunchKernelGGL(trtri_mfma_kernel_f64, dim3(Bt * nblk), dim3(64), 0, (hipStream_t)stream, Lop, Linv, N, Np, nblk);
    return check_launch("trtri_mfma");
}
int launch_trtri_mfma_f32(const float* Lop, float* Linv, int Bt, int N, void* stream) { return launch_trtri_mfma<float>(Lop, Linv, Bt, N, stream); }
int launch_trtri_mfma_f64(const double* Lop, double* Linv, int Bt, int N, void* stream) { return launch_trtri_mfma<double>(Lop, Linv, Bt, N, stream); }

}  // namespace bcbf
