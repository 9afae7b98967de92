// K4+K5+K6+K7: per-instance GP posterior query -- the HBM-bound hot kernel of the control step.
//
// One workgroup per independent instance.  Thread t owns the mirrored pair of V-row blocks
// (t, nrb-1-t) of the forward substitution  W = L^-1 Phi,  so every lane streams the same
// number of bytes of the packed operator (include/bcbf.h "Lop": column-major, 32x32 diagonal
// blocks pre-inverted).  Per 32-column block J:
//   1. owners of rows in J publish their residual r_J to LDS,
//   2. wave 0 forms w_J = inv(L_JJ) r_J (a 32x32 triangular mat-vec, 16 columns per half-wave),
//      accumulates the Gram W'W (fp64) and Vw'W, and publishes w_J,
//   3. every lane subtracts L[rows below, J] w_J from its residuals: one 16-byte load per lane
//      per column (contiguous across the wave), w_J broadcast from LDS.
// Algorithmic HBM bytes per instance: sizeof(T) * [Np(Np+V)/2 + N(2n + C)]  (Lop + X + Vw + UHB).
#include "bcbf_common.h"
#include <type_traits>

#ifndef BCBF_PS_DUNR
#define BCBF_PS_DUNR 4     // unroll of the diagonal-block mat-vec
#endif
#ifndef BCBF_PS_WAVES
#define BCBF_PS_WAVES 2   // occupancy target (waves per SIMD) -> VGPR cap
#endif
#ifndef BCBF_PS_AUX
#define BCBF_PS_AUX 2      // cache-policy bits of the streaming loads: 2 = non-temporal (each byte is read once;
                           // measured +9 % over the default policy, 462 -> 424 us)
#endif
#ifndef BCBF_PS_PKASM
#define BCBF_PS_PKASM 1    // fp32: explicit v_pk_fma_f32 with op_sel broadcast of w (see consume)
#endif
#ifndef BCBF_PS_NQ
#define BCBF_PS_NQ 2       // queries per workgroup of the fp64 shared-GP form (1 = off)
#endif
#ifndef BCBF_PS_VWPF
#define BCBF_PS_VWPF 1     // fetch the Vw rows of a block one block ahead
#endif
#ifndef BCBF_PS_UNR8_MAXC
#define BCBF_PS_UNR8_MAXC 3     // C = 4 would spill at 8 columns per stage
#endif
#ifndef BCBF_PS_UNR
#define BCBF_PS_UNR 4      // columns per software-pipeline stage of the streaming loop
#endif
// jets instantiations (NJ > 0: the rel-degree-2 path, up to 12 right-hand-side columns): columns per pipeline stage and
// occupancy target per element type.  Their Gram / mean sums live in ONE matrix-core accumulator (see step 2b) instead
// of 126 VALU accumulators per lane, which is what lets two waves share a SIMD with 4 columns per stage in flight.
#ifndef BCBF_PQ_UNR
#define BCBF_PQ_UNR 4        // several queries per workgroup (NQ > 1), fp32: columns per pipeline stage (fp64: 2 -- 4 spill)
#endif
#ifndef BCBF_PQ_UNR64
#define BCBF_PQ_UNR64 2      // fp64, several queries per workgroup (shared model; the online path's query + append pair)
#endif
#ifndef BCBF_PJ_UNR32
#define BCBF_PJ_UNR32 4      // fp32, more than 6 right-hand-side columns (unicycle shape: 12); measured 4096 x 512, n=3, m=2:
                             // (UNR, waves/SIMD) = (2,1) 663 us, (2,2) 520, (4,1) 539, (4,2) 419, (8,1) 528
#endif
#ifndef BCBF_PJ_UNR32_NARROW
#define BCBF_PJ_UNR32_NARROW 8   // fp32, up to 6 columns (pendulum shape): 8 columns per stage still fit 2 waves/SIMD: 353 -> 344 us
#endif
#ifndef BCBF_PJ_UNR64
#define BCBF_PJ_UNR64 4      // fp64: (2,1) 633 us, (4,1) 567 (2048 x 512, n=3, m=2); 1024 x 256, n=2, m=1: 70 -> 58 us
#endif
#ifndef BCBF_PJ_WAVES32
#define BCBF_PJ_WAVES32 2
#endif
#ifndef BCBF_PJ_WAVES64
#define BCBF_PJ_WAVES64 1
#endif
#ifndef BCBF_PJ_ONE
#define BCBF_PJ_ONE 0        // fp32 jets with 12 right-hand-side columns (unicycle shape): ONE row block per thread (see ONE below).
                             // OFF -- measured round 5, 4096 x 512, n=3, m=2 (columns per stage, waves/SIMD): pair form (4,2) 0.438 ms;
                             // one row block per thread (8,2) 0.506, (4,2) 0.552, (16,2) 0.555 (36 B scratch), (16,1) 0.598, (8,3) 0.694
                             // (148 B scratch): two waves per instance meet at two barriers per block and both wait for wave 0's
                             // diagonal step -- the bytes in flight gained are lost to that chain
#endif
#ifndef BCBF_PJ_ONE_UNR
#define BCBF_PJ_ONE_UNR 8
#endif
#ifndef BCBF_PJ_ONE_WAVES
#define BCBF_PJ_ONE_WAVES 3
#endif
#ifndef BCBF_PS_UNR64_WIDE
#define BCBF_PS_UNR64_WIDE 2
#endif
#ifndef BCBF_PS_SKIP_DEAD_A
#define BCBF_PS_SKIP_DEAD_A 1
#endif
#ifndef BCBF_PX_UNR32
#define BCBF_PX_UNR32 8      // fp32 query + append column (4 right-hand-side columns): columns per pipeline stage
#endif
#ifndef BCBF_PX_VALU32
#define BCBF_PX_VALU32 1     // fp32 query + append column: per-lane Gram / mean sums instead of the matrix-core accumulator
#endif

namespace bcbf {

template <typename T> __device__ inline T texp(T x);
template <> __device__ inline float texp<float>(float x) { return expf(x); }
template <> __device__ inline double texp<double>(double x) { return exp(x); }

// Bounds-checked buffer loads: a lane whose byte offset is >= the descriptor's size gets zeros and
// generates no memory traffic, so the triangular structure needs no divergent branches (a branch
// around a load makes hipcc wait vmcnt(0) before every load, serialising the stream).
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned;
using u32x2 = __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned;
constexpr int OOB = 0x40000000;   // > any operator size; voffset + soffset stays below 2^31

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release / acquire over EVERY address space:
// hipcc waits for all outstanding global loads (s_waitcnt vmcnt(0)) in front of each barrier -- in this kernel that drained the loads
// of the next column group and the next diagonal block, which are issued a block ahead precisely so that they fly across the block's
// two barriers.  Nothing here passes data between threads through global memory; the threads meet in rbuf / wbuf only.
__device__ inline void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T> struct BufLoad;
template <> struct BufLoad<float> {
    static __device__ inline float4 vec(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, BCBF_PS_AUX);
        return __builtin_bit_cast(float4, v);
    }
    static __device__ inline float one(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        // diagonal-block values: default policy -- their lines are shared with the column's off-diagonal
        // rows that the streaming loads fetch a little later (non-temporal here costs 424 -> 458 us)
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
};
template <> struct BufLoad<double> {
    static __device__ inline double2 vec(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, BCBF_PS_AUX);
        return __builtin_bit_cast(double2, v);
    }
    static __device__ inline double one(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        return __builtin_bit_cast(double, v);
    }
};

// One 16x16 matrix-core accumulator: acc += a(16x4) b(4x16); lane l supplies a[l % 16][l / 16] and b[l / 16][l % 16]
// and holds acc[row(l / 16, r)][l % 16], r = 0..3, with row(g, r) = 4 g + r in fp32 and g + 4 r in fp64
// (tools/dev/probe/mfma64_layout.hip).  Exact IEEE arithmetic in both precisions.
template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
    using acc_t = __attribute__((__vector_size__(4 * sizeof(float)))) float;
    static __device__ inline acc_t mac(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int g, int r) { return 4 * g + r; }
};
template <> struct Mfma16<double> {
    using acc_t = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    static __device__ inline acc_t mac(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int g, int r) { return g + 4 * r; }
};

// NJ = 0: values only (the control-step kernel).  NJ = n > 0: also the first x-derivative jets --
// right-hand sides [Phi, dPhi/dx_1 .. dPhi/dx_n] (CT = C (1+n) columns of the same stream); outputs the full
// Gram Wj'Wj [CT,CT] and Vw'Wj [n,CT], from which the rel-degree-2 terms are formed.
// NQ > 1 (values only, shared GP): one workgroup answers NQ queries with the same stream of L -- NQ C right-hand-side
// columns -- so the factor crosses the L2 -> CU fabric once per NQ queries (the fp64 form of regime S, which has no
// matrix-core kernel, is bound by exactly that traffic).
// RHS (values only, NQ = 1): the same stream as a plain triangular solve -- the right-hand sides are GIVEN rows
// instead of kernel values: r_i = X[i][0..n) - sum_a UHB[i][a] M0[a][0..n)  (X := Xdot [N,n], UHB := UH [N,cu], M0 [cu,n]:
// the whitened targets Vw = L^-1 (Xdot - UH M0) of bcbf_potrs, control_affine_model.py:525-545), written to Wout as
// [N, n]; no Gram / mean.  C = number of solved columns (>= n; extra columns are zero).
// XC = 1 (values only, NQ = 1): ONE extra right-hand-side column rides along -- the column the online path's append needs,
// l = L^-1 (k(X, x2) o (UH B uh2)) for the new observation (x2, uh2): the bordered factor's new row is l itself, its pivot
// sqrt(kappa - l'l), the new whitened target (y - Vw'l) / d (gp_append_inplace_kernel).  Round 3 answered the control query and
// the append on one pass with TWO full queries (2 C = 6 columns at the unicycle shape): in fp64 that form fits two columns
// per pipeline stage and streamed at 4.0 TB/s at N >= 1024; C + 1 = 4 columns run the plain kernel's schedule.
// ONE row block per thread (instead of the mirrored pair): half the residual accumulators and half the bytes per load
// instruction per thread, twice the threads -- for the form whose 12 right-hand-side columns leave the pair form 256 registers,
// two waves per SIMD and four columns per pipeline stage: three waves per SIMD with eight columns per stage each put 1.5 x the
// bytes in flight per CU.  Lane l of wave w holds row block p = 32 w + (l & 31) (l < 32) or its mirror nrb - 1 - p (l >= 32): every
// wave keeps the pair form's balance between long and short columns.
template <typename T, int C, int NJ> __host__ __device__ constexpr bool ps_one_rowblock() {
    return BCBF_PJ_ONE && sizeof(T) == 4 && NJ == 3 && C == 3;
}

template <typename T, int C, int NS, int NJ, int NQ = 1, bool RHS = false, int XC = 0, bool ONEP = false, bool XK = false>
__global__ void __launch_bounds__((sizeof(T) == 8 && NJ == 0 ? 512 : 256),
                                   (NJ > 0 ? (ONEP ? BCBF_PJ_ONE_WAVES : sizeof(T) == 8 || C * (1 + NJ) > 12 || NJ > 3 ? BCBF_PJ_WAVES64 : BCBF_PJ_WAVES32) : BCBF_PS_WAVES))
posterior_step_kernel(const T* __restrict__ Lop, const T* __restrict__ Vw, const T* __restrict__ X,
                      const T* __restrict__ UHB, const T* __restrict__ ell, const T* __restrict__ s2p,
                      const T* __restrict__ Bm, const T* __restrict__ M0, const T* __restrict__ xq,
                      const T* __restrict__ jitter2, T* __restrict__ Mk, T* __restrict__ Bk,
                      T* __restrict__ Wout, T* __restrict__ Gfull, T* __restrict__ Mfull, int shared, int N, int Np,
                      int n, const T* __restrict__ lin, int nq, int Nl, int ldN, int kind, const T* __restrict__ xq2,
                      const T* __restrict__ uh2 = nullptr) {
    // Nl: the padded size the operator is LAID OUT for (>= Np; column lengths, block offsets, batch stride), ldN: rows
    // per instance of X / UH B / Vw.  Nl == Np, ldN == N: the packed layout of exactly N points; larger: capacity-
    // reserving storage of the online path (bcbf_gp_reserve), of which the first N points are live.
    // kind: data kernel -- 0 = RBF (the reference's), 1 = Matern-5/2 (opt-in; values only, the jets assume the RBF).
    // xq2 (NQ = 2, one GP per workgroup): the workgroup's second query, xq2[b] (its first is xq[b]) -- the online path's
    // control query and new observation share ONE pass over the instance's factor (bcbf_gp_append_reserved).
    static_assert(NQ == 1 || NJ == 0, "several queries per workgroup: values only");
    static_assert(!RHS || (NJ == 0 && NQ == 1), "right-hand-side mode: values only, one system per workgroup");
    static_assert(XC == 0 || (XC == 1 && NJ == 0 && NQ == 1 && !RHS), "extra column: values only, one query per workgroup");
    const int cu = RHS ? nq : 0;        // RHS mode: columns of UH (the launcher passes it in the nq slot)
    constexpr int V = Vec<T>::V;
    constexpr int CT = C * (1 + NJ) * NQ + XC;     // right-hand-side columns
    using VecT = typename Vec<T>::type;
    constexpr int RPB = NB / V;          // row blocks per diagonal block
    constexpr int CP = (CT + 3) / 4 * 4; // padded RHS count in LDS
    constexpr int CQ = (CT - XC) / NQ;                         // columns of one query
    constexpr int NGQ = NQ * (CQ * (CQ + 1) / 2);              // Gram entries kept: per query, upper triangle
    constexpr int NG = NGQ + XC;                               // (+ l'l of the append's column)
    // (a <= c, same query) -> slot; with NQ = 1 this is the upper triangle of the full CT x CT Gram
    auto gidx = [](int a, int c) {
        const int qi = a / CQ, a_ = a - qi * CQ, c_ = c - qi * CQ;
        return qi * (CQ * (CQ + 1) / 2) + a_ * CQ - a_ * (a_ - 1) / 2 + (c_ - a_);
    };
    // jets: a w_J row is widened to 16 entries [w (CT), Vw row (n), zeros]: the 32 x 16 tile is both operands of the
    // matrix-core product of step 2b
    // Gram / mean sums on the matrix cores: the forms with many right-hand-side columns (jets, several queries), and the
    // query + append column in fp64 (128-register cap at 512 threads).  In fp32 that form (4 columns, one wave per instance
    // at N <= 512) keeps the plain kernel's per-lane sums: the eight MFMAs + LDS reads per block sat in the one wave's serial
    // chain (4096 x 512: 442 us against the plain kernel's 331)
    constexpr bool MG = NJ > 0 || NQ > 1 || (XC > 0 && (sizeof(T) == 8 || !BCBF_PX_VALU32));
    // ... in one 16-column tile when CT + n <= 16 (every shape up to the unicycle's n=3, m=2), in two otherwise
    // (n=3 m=3, n=4 m=2, n=4 m=3: CT = 16, 15, 20): the product is then 2 x 2 accumulators
    constexpr int NEEDW = CT + (NJ > 0 ? NJ : NS);
    constexpr int NT = MG ? (NEEDW + 15) / 16 : 1;
    constexpr int CW = MG ? 16 * NT : CP;
    static_assert(NT <= 2, "[W, Vw] must fit 32 tile columns");
    __shared__ __attribute__((aligned(16))) T rbuf[NB][CP];
    __shared__ __attribute__((aligned(16))) T wbuf[NB][CW];
    // (XC: the solved column l [Np] and -- tail step -- all solved columns [Np][CT], kept for one burst of stores after the last block;
    //  dynamic shared memory, sized by launch_posterior_query_column_reserved)
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_dyn_smem[];
    T* const xc_l = reinterpret_cast<T*>(ps_dyn_smem);
    T* const xc_w = xc_l + (XC > 0 && Wout != nullptr ? Np : 0);

    const int b = blockIdx.x;
    const int gb = shared ? 0 : b;      // regime S: every query reads the one shared GP (instance 0)
    const int tid = threadIdx.x;
    const int nrb = Np / V;
    const int npairs = nrb / 2;
    constexpr bool ONE = ONEP;
    static_assert(!ONEP || (NQ == 1 && !RHS && XC == 0 && ps_one_rowblock<T, C, NJ>()), "one row block per thread: the fp32 unicycle jets");
    constexpr int NR = ONE ? 1 : 2;     // row blocks per thread
    const int pidx = ONE ? ((tid >> 6) * 32 + (tid & 31)) : tid;
    const bool live = pidx < npairs;
    const int rbA = ONE ? (((tid >> 5) & 1) ? nrb - 1 - pidx : pidx) : tid, rbB = nrb - 1 - tid;
    const T* __restrict__ lop = Lop + (size_t)gb * lop_elems<V>(Nl);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(lop), 0, (int)(lop_elems<V>(Nl) * sizeof(T)), 0x00020000);
    const T* __restrict__ Xb = X + (size_t)gb * ldN * n;
    const T* __restrict__ UHBb = UHB + (size_t)gb * ldN * (RHS ? cu : C);
    const T* __restrict__ Vwb = RHS ? nullptr : Vw + (size_t)gb * ldN * n;

    // ---- prologue: r = Phi rows owned by this thread:  phi_i = s2 exp(-1/2 |(x_i - xq)/ell|^2) * UHB_i
    T xqr[NQ][NS], iell[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {       // a query slot beyond the last query repeats the last one (not stored)
            const int qx = (NQ == 1 || xq2 != nullptr) ? b : min(b * NQ + qi, nq - 1);
            const T* __restrict__ qsrc = (NQ > 1 && qi == 1 && xq2 != nullptr) ? xq2 : xq;
            xqr[qi][d] = (!RHS && d < n) ? qsrc[(size_t)qx * n + d] : T(0);
        }
        iell[d] = (!RHS && d < n) ? T(1) / ell[(size_t)gb * n + d] : T(0);
    }
    T x2r[XC ? NS : 1], uh2r[XC ? C : 1];                      // the append's new observation (XC)
    if constexpr (XC > 0) {
#pragma unroll
        for (int d = 0; d < NS; ++d) x2r[d] = d < n ? xq2[(size_t)b * n + d] : T(0);
#pragma unroll
        for (int c = 0; c < C; ++c) uh2r[c] = uh2[(size_t)b * C + c];
    }
    const T s2 = RHS ? T(0) : s2p[gb];
    // optional linear part of the data kernel, k = s2 (exp(..) + lin x'x') (the CoGP comparator's RBF + Linear,
    // control_affine_model.py:1121-1122); lin == NULL -> 0
    const T linv = (!RHS && lin != nullptr) ? lin[gb] : T(0);
    T m0r[RHS ? BCBF_MAX_CTRL_DIM + 1 : 1][RHS ? C : 1];           // RHS mode: the prior mean M0 [cu, n]
    if constexpr (RHS) {
#pragma unroll
        for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a)
#pragma unroll
            for (int c = 0; c < C; ++c) m0r[a][c] = (a < cu && c < n) ? M0[((size_t)gb * cu + a) * n + c] : T(0);
    }
    // fp32 fast path: residuals held as register PAIRS over two consecutive rows, so that the update is one
    // v_pk_fma_f32 per (row pair, column) with w broadcast by op_sel -- the compiler's own packing mixes row- and
    // column-pairs and pays ~0.9 v_mov per packed multiply-add to shuffle the pairs (47 % of the loop's VALU issue).
    constexpr bool PK = BCBF_PS_PKASM && std::is_same<T, float>::value;
    using f32x2 = __attribute__((__vector_size__(2 * sizeof(float)))) float;
    T acc[PK ? 1 : NR][PK ? 1 : V][PK ? 1 : CT];
    f32x2 accp[PK ? NR : 1][PK ? V / 2 : 1][PK ? CT : 1];
#define BCBF_ACC(r, v, c) (*(PK ? reinterpret_cast<T*>(&accp[r][(v) >> 1][c]) + ((v) & 1) : &acc[PK ? 0 : (r)][PK ? 0 : (v)][PK ? 0 : (c)]))
    // The training-input loads of a row set (V rows) are issued together, bounds-checked (rows >= N, state dimensions
    // >= n and idle lanes read zeros without a branch), then the kernel values are formed: one exposed load latency
    // per row set instead of one per row.
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(Xb), 0, (int)((size_t)N * n * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(UHBb), 0, (int)((size_t)N * (RHS ? cu : C) * sizeof(T)), 0x00020000);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int rb = r == 0 ? rbA : rbB;
        if constexpr (RHS) {
            // given right-hand sides: r_i = Xdot_i - M0' uh_i  (rows >= N, columns >= n and idle lanes: zeros)
            T yv[V][C], uh[V][BCBF_MAX_CTRL_DIM + 1];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int i = rb * V + v;
#pragma unroll
                for (int c = 0; c < C; ++c)
                    yv[v][c] = BufLoad<T>::one(rsrc_x, (live && c < n) ? (i * n + c) * (int)sizeof(T) : OOB, 0);
#pragma unroll
                for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a)
                    uh[v][a] = BufLoad<T>::one(rsrc_u, (live && a < cu) ? (i * cu + a) * (int)sizeof(T) : OOB, 0);
            }
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    T y = yv[v][c];
#pragma unroll
                    for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a) y -= uh[v][a] * m0r[a][c];
                    BCBF_ACC(r, v, c) = y;
                }
            continue;
        }
        T xv[V][NS], uv[V][C];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int i = rb * V + v;
#pragma unroll
            for (int d = 0; d < NS; ++d)
                xv[v][d] = BufLoad<T>::one(rsrc_x, (live && d < n) ? (i * n + d) * (int)sizeof(T) : OOB, 0);
#pragma unroll
            for (int c = 0; c < C; ++c)
                uv[v][c] = BufLoad<T>::one(rsrc_u, live ? (i * C + c) * (int)sizeof(T) : OOB, 0);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                if constexpr (XC > 0) {                // the extra column: k(X_i, x2) (UH B)_i . uh2 (first: nothing of the query's own columns is live yet)
                    T e2 = T(0), ud = T(0);
#pragma unroll
                    for (int d = 0; d < NS; ++d) { const T z = (xv[v][d] - x2r[d]) * iell[d]; e2 += z * z; }
#pragma unroll
                    for (int c = 0; c < C; ++c) ud += uv[v][c] * uh2r[c];
                    T sh2, dsh2;                       // XK: the opt-in data kernels (its own instantiation: the RBF form keeps its registers)
                    if constexpr (XK) kernel_shape(kind, e2, [](T q_) { return texp<T>(q_); }, sh2, dsh2);
                    else sh2 = texp<T>(T(-0.5) * e2);
                    BCBF_ACC(r, v, CT - 1) = s2 * sh2 * ud;
                }
                T d2 = T(0), dot = T(0);
#pragma unroll
                for (int d = 0; d < NS; ++d) {         // d >= n: xv = xqr = iell = 0
                    const T z = (xv[v][d] - xqr[qi][d]) * iell[d];
                    d2 += z * z;
                    dot += xv[v][d] * xqr[qi][d];
                }
                T shape, dshape;                       // dshape: d shape / d x_q,d = dshape * (X_id - x_q,d) / ell_d^2
                if (kind != 0) {                       // opt-in: Matern-5/2 (1), RBF x Matern-5/2 (2) -- bcbf_common.h
                    kernel_shape(kind, d2, [](T q_) { return texp<T>(q_); }, shape, dshape);
                } else { shape = texp<T>(T(-0.5) * d2); dshape = shape; }
                const T k = s2 * (shape + linv * dot);
                const T kd = s2 * dshape;              // (jets: no linear part)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const T ub = uv[v][c];             // 0 for rows >= N and idle lanes: the row contributes nothing
                    BCBF_ACC(r, v, qi * C + c) = k * ub;
#pragma unroll
                    for (int d = 0; d < NJ; ++d) {     // d Phi / d x_d = -(x_d - X_id)/ell_d^2 * Phi
                        const T dz = (xv[v][d] - xqr[qi][d]) * iell[d] * iell[d];
                        BCBF_ACC(r, v, (1 + d) * C + c) = dz * kd * ub;
                    }
                }
            }
        }
    }

    // Gram W'W: per-lane partial sums in T (32 rows x nblk terms each), reduced and subtracted in fp64
    T gram[MG ? 1 : NG];
#pragma unroll
    for (int g = 0; g < (MG ? 1 : NG); ++g) gram[g] = T(0);
    T mk[MG ? 1 : NS][MG ? 1 : CT];
#pragma unroll
    for (int d = 0; d < (MG ? 1 : NS); ++d)
#pragma unroll
        for (int c = 0; c < (MG ? 1 : CT); ++c) mk[d][c] = T(0);
    typename Mfma16<T>::acc_t gacc[NT][NT];                         // jets: [W, Vw]' [W, Vw] summed over all rows
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) gacc[ti][tj] = typename Mfma16<T>::acc_t{T(0), T(0), T(0), T(0)};
    if constexpr (MG) {                                             // columns CT + n .. 15 of the tile stay zero
        for (int i = tid; i < NB * CW; i += blockDim.x) wbuf[i / CW][i % CW] = T(0);
    }

    const int nblk = Np / NB;
    // Streaming pipeline state.  Column groups of UNR columns form ONE stream over all blocks: the
    // loads of the next group (also across a block boundary) and the diagonal-block values of the
    // next block are in flight while the current group / the barriers / the diagonal mat-vec run.
    // fp32 with the packed update has registers to spare: 8 columns per stage (16-32 KB in flight per wave), +2.5 %
    constexpr int UNR = ONE ? BCBF_PJ_ONE_UNR : NJ > 0 ? (CT > 12 ? 2 : sizeof(T) == 8 ? BCBF_PJ_UNR64 : (CT <= 6 ? BCBF_PJ_UNR32_NARROW : BCBF_PJ_UNR32))
                               : (NQ > 1 ? (sizeof(T) == 8 ? BCBF_PQ_UNR64 : BCBF_PQ_UNR) : (XC > 0 && PK ? BCBF_PX_UNR32 : PK && C <= BCBF_PS_UNR8_MAXC ? 8
                                  // (fp64 with four columns or a wide state: two columns per stage -- at four the 256-register form spills
                                  //  108-268 B per lane and streams at 0.32-0.46 of HBM, round 5)
                                  : (sizeof(T) == 8 && XC == 0 && (C >= 4 || NS > 4)) ? BCBF_PS_UNR64_WIDE : BCBF_PS_UNR)), NGRP = NB / UNR, HALF = NB / 2;
    static_assert(NGRP % 2 == 0, "pipeline processes two groups per trip");
    VecT la0[UNR], lb0[UNR], la1[UNR], lb1[UNR];
    T dval[HALF];
    const int di = tid & 31, dh = (tid >> 5) & 1;
    // AL (compile-time): the wave still owns live A-side rows (its low row blocks); false: loads and updates of the A side are skipped
    auto issue = [&](VecT* la, VecT* lb, int J, int g, auto AL) __attribute__((always_inline)) {
        // column j of block column J stores rows 32 (J+1) .. Np-1: per-lane offset relative to the first stored row,
        // scalar offset = start of the column (lop_base(j) + 32 (J+1) >= 0: buffer offsets are unsigned)
        const int rbmin = (J + 1) * RPB;
        const int voffA = (live && rbA >= rbmin) ? (rbA - rbmin) * V * (int)sizeof(T) : OOB;
        const int voffB = (live && rbB >= rbmin) ? (rbB - rbmin) * V * (int)sizeof(T) : OOB;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int soff = (lop_base<V>(J * NB + g * UNR + u, Nl) + (J + 1) * NB) * (int)sizeof(T);
            if constexpr (decltype(AL)::value) la[u] = BufLoad<T>::vec(rsrc, voffA, soff);
            if constexpr (!ONE) lb[u] = BufLoad<T>::vec(rsrc, voffB, soff);
        }
    };
    auto consume = [&](const VecT* la, const VecT* lb, int jj0, auto AL) __attribute__((always_inline)) {
        constexpr bool ALV = decltype(AL)::value;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if constexpr (PK) {
                using f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
                f32x2 wp[CP / 2];                                     // w_J row, two columns per register pair
#pragma unroll
                for (int q = 0; q < CP / 4; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&wbuf[jj0 + u][4 * q]);
                    wp[2 * q] = __builtin_shufflevector(w4, w4, 0, 1);
                    wp[2 * q + 1] = __builtin_shufflevector(w4, w4, 2, 3);
                }
                const f32x4 a4 = (ALV || ONE) ? __builtin_bit_cast(f32x4, la[u]) : f32x4{}, b4 = __builtin_bit_cast(f32x4, lb[ONE ? 0 : u]);
                const f32x2 pa2[2] = {__builtin_shufflevector(a4, a4, 0, 1), __builtin_shufflevector(a4, a4, 2, 3)};
                const f32x2 pb2[2] = {__builtin_shufflevector(b4, b4, 0, 1), __builtin_shufflevector(b4, b4, 2, 3)};
#pragma unroll
                for (int vp = 0; vp < 2; ++vp)
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        // acc.{lo,hi} -= p.{lo,hi} * w[c]  (w[c] = low or high half of its pair, broadcast by op_sel)
                        if (c & 1) {
                            if constexpr (ALV || ONE)
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
                                : "+v"(accp[0][vp][c]) : "v"(pa2[vp]), "v"(wp[c >> 1]));
                            if constexpr (!ONE)
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
                                : "+v"(accp[NR - 1][vp][c]) : "v"(pb2[vp]), "v"(wp[c >> 1]));
                        } else {
                            if constexpr (ALV || ONE)
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
                                : "+v"(accp[0][vp][c]) : "v"(pa2[vp]), "v"(wp[c >> 1]));
                            if constexpr (!ONE)
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
                                : "+v"(accp[NR - 1][vp][c]) : "v"(pb2[vp]), "v"(wp[c >> 1]));
                        }
                    }
            } else {
                T wj[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c) wj[c] = wbuf[jj0 + u][c];
                const T* pa = reinterpret_cast<const T*>(&la[u]);
                const T* pb = reinterpret_cast<const T*>(&lb[u]);
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        if constexpr (ALV || ONE) acc[0][v][c] -= pa[v] * wj[c];
                        if constexpr (!ONE) acc[NR - 1][v][c] -= pb[v] * wj[c];
                    }
            }
        }
    };
    auto issue_diag = [&](int J) {      // inv(L_JJ)[di][dh*16 + q], wave 0 only (others, and the zeros above the
#pragma unroll                          // diagonal, which are not stored: out of range -> 0, no traffic)
        for (int q = 0; q < HALF; ++q) {
            const int jj = dh * HALF + q;
            const int voff = (tid < 64 && di >= jj) ? (lop_dinv_col(jj) + di) * (int)sizeof(T) : OOB;
            dval[q] = BufLoad<T>::one(rsrc, voff, lop_dinv_block(J, Nl) * (int)sizeof(T));
        }
    };
#if BCBF_PS_VWPF
    // whitened targets of the block's 32 rows (lanes 0-31 of wave 0), fetched one block ahead with the same
    // bounds-checked loads: the diagonal step is the serial part of the block, a global load there is exposed latency
    const __amdgpu_buffer_rsrc_t rsrc_vw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(RHS ? Xb : Vwb), 0, (int)((size_t)N * n * sizeof(T)), 0x00020000);
    T vwn[NS];
    auto issue_vw = [&](int J) {
        if constexpr (RHS) return;
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            const int row = J * NB + di;
            const int voff = (tid < 32 && d < n) ? (row * n + d) * (int)sizeof(T) : OOB;      // row >= N: out of range
            vwn[d] = BufLoad<T>::one(rsrc_vw, voff, 0);
        }
    };
    issue_vw(0);
#endif
    issue_diag(0);
    if (nblk > 1) issue(la0, lb0, 0, 0, std::true_type{});
    // One block of the forward substitution.  AL (compile-time): this wave still owns live A-side rows; the blocks after its last live
    // one run the B side only (BCBF_PS_SKIP_DEAD_A) -- in a SECOND loop, so that the A side's registers are dead there
    auto block_body = [&](const int J, auto AL) __attribute__((always_inline)) {
        const int row0 = J * NB;
        // 1. publish r_J
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int rb = r == 0 ? rbA : rbB;
            if (live && rb / RPB == J) {
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int c = 0; c < CT; ++c) rbuf[rb * V + v - row0][c] = BCBF_ACC(r, v, c);
            }
        }
        lds_barrier();
        // 2. diagonal block: w_J = inv(L_JJ) r_J  (wave 0; lane = (row di, column half dh))
        if (tid < 64) {
            T w[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) w[c] = T(0);
#pragma unroll
            for (int q = 0; q < HALF; ++q) {
#pragma unroll
                for (int c = 0; c < CT; ++c) w[c] += dval[q] * rbuf[dh * HALF + q][c];
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) w[c] += __shfl_xor(w[c], 32, 64);
            if (dh == 0) {
#pragma unroll
                for (int c = 0; c < CT; ++c) wbuf[di][c] = w[c];
                if constexpr (RHS) {                    // the solved rows ARE the output: Vw[N, n]
                    if (row0 + di < N) {
#pragma unroll
                        for (int c = 0; c < C; ++c) if (c < n) Wout[((size_t)b * ldN + row0 + di) * n + c] = w[c];
                    }
                } else if (XC > 0) {                            // the append's column l, a vector [Np] per instance
                    // (into LDS here, to memory after the last block: a global store in the diagonal step sits in the chain the whole
                    //  workgroup waits for -- stores and loads retire through ONE in-order counter, so every load issued after it
                    //  waited out the store's acknowledgement: 4096 x 472 fp32, 375 us with the stores here, 317 without)
                    if (Wout != nullptr) xc_l[row0 + di] = w[CT - 1];
                    if (Mfull != nullptr) {                     // ... and all CT solved columns [Np, CT] (the tail step, tail.hip)
#pragma unroll
                        for (int c = 0; c < CT; ++c) xc_w[(row0 + di) * CT + c] = w[c];
                    }
                } else if (NJ > 0 && Wout != nullptr) {        // jets: all CT columns [Phi, dPhi/dx_1 ..] of this row
#pragma unroll
                    for (int c = 0; c < CT; ++c) Wout[((size_t)b * Np + row0 + di) * CT + c] = w[c];
                } else if (Wout != nullptr) {
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        if (NQ == 1 || b * NQ + qi < nq) {
#pragma unroll
                            for (int c = 0; c < C; ++c) Wout[((size_t)(b * NQ + qi) * Np + row0 + di) * C + c] = w[qi * C + c];
                        }
                }
                int g = 0;
                if constexpr (!RHS && !MG) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
                    for (int a = 0; a < CQ; ++a)
#pragma unroll
                        for (int c = a; c < CQ; ++c) gram[g++] += w[qi * CQ + a] * w[qi * CQ + c];
                if constexpr (XC > 0) gram[NGQ] += w[CT - 1] * w[CT - 1];
                }
                (void)g;
                if constexpr (MG) {
#if BCBF_PS_VWPF
                    // the Vw rows of this block (fetched a block ahead; zeros beyond row N / state dimension n) complete
                    // the tile: columns CT .. CT + n - 1
#pragma unroll
                    for (int d = 0; d < NS; ++d)
                        if (CT + d < CW) wbuf[di][CT + d] = vwn[d];
#else
                    static_assert(!MG, "jets need the prefetched Vw rows (BCBF_PS_VWPF)");
#endif
                } else if constexpr (RHS) { /* no mean accumulation */ } else {
#if BCBF_PS_VWPF
#pragma unroll
                for (int d = 0; d < NS; ++d) {            // Vw rows of this block were fetched a block ahead (zeros
#pragma unroll                                            // beyond row N and state dimension n)
                    for (int c = 0; c < CT; ++c) mk[d][c] += vwn[d] * w[c];
                }
#else
                const int row = row0 + di;
                if (row < N) {
#pragma unroll
                    for (int d = 0; d < NS; ++d)
                        if (d < n) {
                            const T vw = Vwb[(size_t)row * n + d];
#pragma unroll
                            for (int c = 0; c < CT; ++c) mk[d][c] += vw * w[c];
                        }
                }
#endif
                }
            }
        }
        lds_barrier();
        if constexpr (MG) {
            // 2b. Gram and mean sums of the block on the matrix cores: with T_J = [w_J, Vw_J, 0] (32 x 16, in wbuf),
            // gacc += T_J' T_J -- rows / columns < CT: the Gram Wj'Wj, rows CT .. CT+n-1: Vw'Wj.  A and B operand of
            // a k-step are the SAME register (a[i][k] = T_J[k][i] = b[k][i]); 8 k-steps of 4 rows; the reads are
            // 64 consecutive words each.  One accumulator (4 registers) replaces CT (CT+1) / 2 + n CT VALU sums.
            // AFTER the barrier and spread over the workgroup's waves (wave w takes k-steps w, w + nw, ...; the partial
            // accumulators meet in the epilogue): inside wave 0's diagonal step these MFMAs were part of the serial chain
            // every other wave waits for -- at N >= 1024 in fp64 (6-8 waves) a sixth of a block's time.  wbuf is not written
            // again before the next block's first barrier.
            const int nw = blockDim.x >> 6, wv = tid >> 6, kq = (tid >> 4) & 3, ci = tid & 15;
            for (int s8 = wv; s8 < NB / 4; s8 += nw) {
                T a[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) a[t] = wbuf[4 * s8 + kq][16 * t + ci];
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) gacc[ti][tj] = Mfma16<T>::mac(a[ti], a[tj], gacc[ti][tj]);
            }
        }
        // 3. rows below the block:  r -= L[:, J] w_J   (group 0 of this block is already in flight)
        if (J + 1 < nblk) {
            issue_diag(J + 1);
#if BCBF_PS_VWPF
            issue_vw(J + 1);
#endif
            // the block's column groups; a wave whose A-side row blocks all lie above the rows this block column still touches runs
            // the B side only (BCBF_PS_SKIP_DEAD_A: the dead half cost its loads' issue slots and its multiply-adds -- a quarter of
            // the kernel's instructions over a pass, which the forms near their issue limit feel: jets, the append's column)
            auto stream_block = [&](auto AL) __attribute__((always_inline)) {
#pragma unroll 1
                for (int g = 0; g < NGRP; g += 2) {
                    issue(la1, lb1, J, g + 1, AL);
                    consume(la0, lb0, g * UNR, AL);
                    if (g + 2 < NGRP) issue(la0, lb0, J, g + 2, AL);
                    else if (J + 2 < nblk) issue(la0, lb0, J + 1, 0, AL);
                    consume(la1, lb1, (g + 1) * UNR, AL);
                }
            };
            stream_block(AL);
        }
    };
    {
        // a wave's A-side row blocks are [64 w, 64 w + 63]: live in block J while (J + 1) RPB <= the largest of them
        const int rbA_wave_max = __builtin_amdgcn_readfirstlane(tid) | 63;
        const int Jsplit = (BCBF_PS_SKIP_DEAD_A && !ONE) ? min(nblk, rbA_wave_max / RPB) : nblk;
        for (int J = 0; J < Jsplit; ++J) block_body(J, std::true_type{});
        for (int J = Jsplit; J < nblk; ++J) block_body(J, std::false_type{});
    }

    if constexpr (XC > 0) {
        // the append's column (and, for the tail step, every solved column): LDS -> memory, 16 bytes per lane, nothing waits for it
        lds_barrier();
        using VecT_ = typename Vec<T>::type;
        if (Wout != nullptr)
            for (int i = tid; i < Np / V; i += blockDim.x)
                reinterpret_cast<VecT_*>(Wout + (size_t)b * Np)[i] = reinterpret_cast<const VecT_*>(xc_l)[i];
        if (Mfull != nullptr)
            for (int i = tid; i < Np * CT / V; i += blockDim.x)
                reinterpret_cast<VecT_*>(Mfull + (size_t)b * Np * CT)[i] = reinterpret_cast<const VecT_*>(xc_w)[i];
    }
    // ---- epilogue: wave 0 reduces the Gram and Vw'W and writes Mk, Bk
    if constexpr (RHS) return;
    if constexpr (MG) {
        // the waves' partial accumulators -> wave 0 (through wbuf, free now: one accumulator register of every wave at a time)
        const int nw = blockDim.x >> 6;
        if (nw > 1) {
            T* red = &wbuf[0][0];
            static_assert(!MG || NB * CW >= 512, "reduction buffer: 8 waves x 64 lanes");
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        lds_barrier();
                        if (tid >= 64) red[tid] = gacc[ti][tj][r];
                        lds_barrier();
                        if (tid < 64) {
                            T sum = gacc[ti][tj][r];
                            for (int w_ = 1; w_ < nw; ++w_) sum += red[w_ * 64 + tid];
                            gacc[ti][tj][r] = sum;
                        }
                    }
        }
    }
    if constexpr (MG && NJ > 0) {
        // jets: every lane of wave 0 writes its four entries (i = row(l / 16, r), j = l % 16) of [W, Vw]'[W, Vw]
        if (tid < 64) {
            const int grp = tid >> 4;
            T* Gb = Gfull + (size_t)b * CT * CT;
            T* Mb = Mfull + (size_t)b * n * CT;
            const T* M0b = M0 + (size_t)gb * C * n;
            const T* Bmb = Bm + (size_t)gb * C * C;
            T* Mkb = Mk + (size_t)b * n * C;
            T* Bkb = Bk + (size_t)b * C * C;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + Mfma16<T>::row(grp, r), j = 16 * tj + (tid & 15);
                const T val = gacc[ti][tj][r];
                if (j < CT) {
                    if (i < CT) {
                        Gb[i * CT + j] = val;
                        if (i < C && j < C) Bkb[i * C + j] = (T)((double)s2 * (double)Bmb[i * C + j] - (double)val);
                    } else if (i < CT + n) {
                        const int d = i - CT;
                        Mb[d * CT + j] = val;
                        if (j < C) Mkb[d * C + j] = M0b[j * n + d] + val;
                    }
                }
            }
        }
        return;
    }
    if constexpr (MG && XC > 0) {
        // query + one column: the query's Gram block / mean as usual; row / column C of the accumulator is the column l:
        // (C, C) = l'l, (CT + d, C) = (Vw'l)_d -> Gfull[b][1 + n]
        if (tid < 64) {
            const int j = tid & 15, grp = tid >> 4;
            const T* M0b = M0 + (size_t)gb * C * n;
            const T* Bmb = Bm + (size_t)gb * C * C;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = Mfma16<T>::row(grp, r);
                const T val = gacc[0][0][r];
                if (j < C) {
                    if (i < C) {
                        double v = (double)s2 * (double)Bmb[i * C + j] - (double)val;
                        if (i == j && jitter2 != nullptr) v += (double)jitter2[(size_t)b * C + i];
                        Bk[(size_t)b * C * C + i * C + j] = (T)v;
                    } else if (i >= CT && i < CT + n) {
                        const int d = i - CT;
                        Mk[(size_t)b * n * C + d * C + j] = M0b[j * n + d] + val;
                    }
                } else if (j == C && Gfull != nullptr) {
                    if (i == C) Gfull[(size_t)b * (1 + n)] = val;
                    else if (i >= CT && i < CT + n) Gfull[(size_t)b * (1 + n) + 1 + (i - CT)] = val;
                }
            }
        }
        return;
    }
    if constexpr (MG && NQ > 1) {
        // several queries per workgroup: query qi owns columns qi C .. qi C + C - 1; its Gram block is the diagonal block of
        // the accumulator, its mean the rows CT .. CT + n - 1 of its columns
        if (tid < 64) {
            const int j = tid & 15, grp = tid >> 4;
            const int qi = j / C, c = j - qi * C;
            const int qx = b * NQ + qi;
            if (j < CT && (xq2 != nullptr || qx < nq)) {
                const T* M0b = M0 + (size_t)gb * C * n;
                const T* Bmb = Bm + (size_t)gb * C * C;
                double kss = (double)s2;                      // k(xq, xq) = s2 (1 + lin |xq|^2)
                if (lin != nullptr) {
                    double q2 = 0.0;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (q == qi) {
#pragma unroll
                            for (int d = 0; d < NS; ++d) q2 += (double)xqr[q][d] * (double)xqr[q][d];
                        }
                    kss *= 1.0 + (double)linv * q2;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = Mfma16<T>::row(grp, r);
                    const T val = gacc[0][0][r];
                    if (i >= qi * C && i < qi * C + C) {
                        const int a = i - qi * C;
                        double v = kss * (double)Bmb[a * C + c] - (double)val;
                        if (a == c && jitter2 != nullptr) v += (double)jitter2[(size_t)qx * C + a];
                        Bk[(size_t)qx * C * C + a * C + c] = (T)v;
                    } else if (i >= CT && i < CT + n) {
                        const int d = i - CT;
                        Mk[(size_t)qx * n * C + d * C + c] = M0b[c * n + d] + val;
                    }
                }
            }
        }
        return;
    }
    if (tid < 64) {
        double gsum[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) gsum[g] = wave_sum((double)gram[g]);
#pragma unroll
        for (int d = 0; d < NS; ++d)
            if (d < n) {
#pragma unroll
                for (int c = 0; c < CT; ++c) mk[d][c] = wave_sum(mk[d][c]);
            }
        if (tid == 0) {
            if (NJ > 0 && Gfull != nullptr && Mfull != nullptr) {
                T* Gb = Gfull + (size_t)b * CT * CT;
                T* Mb = Mfull + (size_t)b * n * CT;
#pragma unroll
                for (int a = 0; a < CT; ++a)
#pragma unroll
                    for (int c = a; c < CT; ++c) { Gb[a * CT + c] = (T)gsum[gidx(a, c)]; Gb[c * CT + a] = (T)gsum[gidx(a, c)]; }
#pragma unroll
                for (int d = 0; d < NS; ++d)
                    if (d < n) {
#pragma unroll
                        for (int c = 0; c < CT; ++c) Mb[d * CT + c] = mk[d][c];
                    }
            }
            if constexpr (XC > 0) {                              // the append's column: l'l and Vw'l -> Gfull[b][1 + n]
                if (Gfull != nullptr) {
                    Gfull[(size_t)b * (1 + n)] = (T)gsum[NGQ];
#pragma unroll
                    for (int d = 0; d < NS; ++d)
                        if (d < n) Gfull[(size_t)b * (1 + n) + 1 + d] = mk[d][CT - 1];
                }
            }
            const T* M0b = M0 + (size_t)gb * C * n;
            const T* Bmb = Bm + (size_t)gb * C * C;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const int qx = b * NQ + qi;                      // this slot's query
                if (NQ > 1 && qx >= nq) break;
                T* Mkb = Mk + (size_t)qx * n * C;
#pragma unroll
                for (int d = 0; d < NS; ++d)
                    if (d < n) {
#pragma unroll
                        for (int c = 0; c < C; ++c) Mkb[d * C + c] = M0b[c * n + d] + mk[d][qi * C + c];
                    }
                T* Bkb = Bk + (size_t)qx * C * C;
                double kss = (double)s2;                      // k(xq, xq) = s2 (1 + lin |xq|^2)
                if (lin != nullptr) {
                    double q2 = 0.0;
#pragma unroll
                    for (int d = 0; d < NS; ++d) q2 += (double)xqr[qi][d] * (double)xqr[qi][d];
                    kss *= 1.0 + (double)linv * q2;
                }
#pragma unroll
                for (int a = 0; a < C; ++a)
#pragma unroll
                    for (int c = a; c < C; ++c) {
                        const double G = gsum[gidx(qi * C + a, qi * C + c)];
                        double v1 = kss * (double)Bmb[a * C + c] - G;
                        double v2 = kss * (double)Bmb[c * C + a] - G;
                        if (a == c && jitter2 != nullptr) { v1 += (double)jitter2[(size_t)qx * C + a]; v2 = v1; }
                        Bkb[a * C + c] = (T)v1;
                        Bkb[c * C + a] = (T)v2;
                    }
            }
        }
    }
}

int launch_posterior_jets_mfma(const float* Lop, const float* Vw, const float* X, const float* UHB, const float* ell, const float* s2,
                               const float* Bm, const float* M0, const float* xq, float* Mk, float* Bk, float* Wout, float* Gfull, float* Mfull,
                               int shared, int Bt, int N, int n, int m, int kind, hipStream_t st);       // jets_mfma.hip
bool posterior_jets_mfma_preferred(int N, int n, int m);
template <typename T>
static int launch_posterior_step(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2,
                                 const T* Bm, const T* M0, const T* xq, const T* jitter2, T* Mk, T* Bk,
                                 T* Wout, int shared, int Bt, int N, int n, int m, void* stream,
                                 T* Gfull = nullptr, T* Mfull = nullptr, const T* lin = nullptr, int Ncap = 0, int kind = 0,
                                 const T* xq2 = nullptr) {
    if (Bt <= 0) return BCBF_OK;
    if (!Lop || !Vw || !X || !UHB || !ell || !s2 || !Bm || !M0 || !xq || !Mk || !Bk) return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;
    constexpr int V = Vec<T>::V;
    const int Np = round_up(N, NB);
    if (Ncap != 0 && Ncap < N) return BCBF_EINVAL;
    const int Nl = Ncap ? round_up(Ncap, NB) : Np, ldN = Ncap ? Ncap : N;     // capacity-reserving storage (bcbf_gp_reserve)
    const int npairs = Np / V / 2;
    const int threads = round_up(npairs, 64);
    if (threads > (sizeof(T) == 8 && Gfull == nullptr ? 512 : 256)) return BCBF_EINVAL;   // N <= 2048 (fp64 jets: 1024)
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(Bt), block(threads);
#define BCBF_PS_LAUNCH(CC, NSS) hipLaunchKernelGGL((posterior_step_kernel<T, CC, NSS, 0>), grid, block, 0, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, Wout, nullptr, nullptr, shared, N, Np, n, lin, Bt, Nl, ldN, kind, (const T*)nullptr)
#define BCBF_PJ_LAUNCH(CC, NN) hipLaunchKernelGGL((posterior_step_kernel<T, CC, 4, NN>), grid, block, 0, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, Wout, Gfull, Mfull, shared, N, Np, n, (const T*)nullptr, Bt, Nl, ldN, kind, (const T*)nullptr)
    if (xq2 != nullptr) {
        // two queries per instance (xq[b], xq2[b]) on one pass over its factor: outputs interleaved, Mk / Bk [Bt, 2, ..],
        // Wout [Bt, 2, Np, C]
        if (shared || Gfull || lin || n > 4 || m > 2) return BCBF_EINVAL;
#define BCBF_P2_LAUNCH(CC) hipLaunchKernelGGL((posterior_step_kernel<T, CC, 4, 0, 2>), grid, block, 0, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, Wout, nullptr, nullptr, 0, N, Np, n, lin, 2 * Bt, Nl, ldN, kind, xq2)
        if (m == 1) BCBF_P2_LAUNCH(2);
        else BCBF_P2_LAUNCH(3);
#undef BCBF_P2_LAUNCH
        return check_launch("posterior_pair");
    }
    if (Gfull != nullptr) {      // jets: (n, m) combinations compiled in
        if (!Mfull || lin || kind < 0 || kind >= BCBF_KINDS) return BCBF_EINVAL;
        if (n > 4) return BCBF_EINVAL;                  // (the rel-degree-2 terms kernel holds n <= 4 too)
        if constexpr (sizeof(T) == 4) {
            // fp32, N <= 512, twelve right-hand-side columns (n = 3, m = 2): on the matrix cores, the operator through an LDS-DMA ring (jets_mfma.hip).
            // BCBF_JETS_MFMA=0: the streaming kernel; =2: every shape the form takes (n = 2 / 3, m = 1 / 2)
            const char* e_ = getenv("BCBF_JETS_MFMA");
            const int jm = e_ ? atoi(e_) : 1;
            if (jm && (jm == 2 || posterior_jets_mfma_preferred(N, n, m)) && Ncap == 0 && launch_posterior_jets_mfma(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, Mk, Bk, Wout, Gfull, Mfull, shared, Bt, N, n, m, kind, st) == 0)
                return check_launch("posterior_jets_mfma");
        }
        switch (10 * n + m) {                           // every (n <= 4, m <= 3): C = 1 + m columns x (1 + n) jets
            case 11: BCBF_PJ_LAUNCH(2, 1); break;
            case 12: BCBF_PJ_LAUNCH(3, 1); break;
            case 13: BCBF_PJ_LAUNCH(4, 1); break;
            case 21: BCBF_PJ_LAUNCH(2, 2); break;
            case 22: BCBF_PJ_LAUNCH(3, 2); break;
            case 23: BCBF_PJ_LAUNCH(4, 2); break;
            case 31: BCBF_PJ_LAUNCH(2, 3); break;
            case 32:
                if constexpr (ps_one_rowblock<T, 3, 3>()) {
                    if (2 * npairs <= 256) {            // (N <= 1024: one row block per thread, twice the threads)
                        block = dim3(round_up(2 * npairs, 64));
                        hipLaunchKernelGGL((posterior_step_kernel<T, 3, 4, 3, 1, false, 0, true>), grid, block, 0, st, Lop, Vw, X, UHB, ell,
                                           s2, Bm, M0, xq, jitter2, Mk, Bk, Wout, Gfull, Mfull, shared, N, Np, n, (const T*)nullptr, Bt,
                                           Nl, ldN, kind, (const T*)nullptr);
                        break;
                    }
                }
                BCBF_PJ_LAUNCH(3, 3);
                break;
            case 33: BCBF_PJ_LAUNCH(4, 3); break;
            case 41: BCBF_PJ_LAUNCH(2, 4); break;
            case 42: BCBF_PJ_LAUNCH(3, 4); break;
            case 43: BCBF_PJ_LAUNCH(4, 4); break;
            default: return BCBF_EINVAL;
        }
    } else if (BCBF_PS_NQ > 1 && shared && sizeof(T) == 8 && n <= 4 && m <= 2 && Bt >= 2 * BCBF_PS_NQ) {
      if constexpr (sizeof(T) == 8) {
        // fp64, one model, many queries: BCBF_PS_NQ queries per workgroup share the stream of L
        const dim3 gridq((Bt + BCBF_PS_NQ - 1) / BCBF_PS_NQ);
#define BCBF_PQ_LAUNCH(CC) hipLaunchKernelGGL((posterior_step_kernel<T, CC, 4, 0, BCBF_PS_NQ>), gridq, block, 0, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, Wout, nullptr, nullptr, shared, N, Np, n, lin, Bt, Nl, ldN, kind, (const T*)nullptr)
        if (m == 1) BCBF_PQ_LAUNCH(2);
        else BCBF_PQ_LAUNCH(3);
#undef BCBF_PQ_LAUNCH
      }
    } else if (n <= 4) {
        switch (m) {
            case 1: BCBF_PS_LAUNCH(2, 4); break;
            case 2: BCBF_PS_LAUNCH(3, 4); break;
            case 3: BCBF_PS_LAUNCH(4, 4); break;
            default: return BCBF_EINVAL;
        }
    } else {
        switch (m) {
            case 1: BCBF_PS_LAUNCH(2, BCBF_MAX_STATE_DIM); break;
            case 2: BCBF_PS_LAUNCH(3, BCBF_MAX_STATE_DIM); break;
            case 3: BCBF_PS_LAUNCH(4, BCBF_MAX_STATE_DIM); break;
            default: return BCBF_EINVAL;
        }
    }
#undef BCBF_PS_LAUNCH
#undef BCBF_PJ_LAUNCH
    return check_launch("posterior_step");
}

// Forward solve Vw = L^-1 (Xdot - UH M0) on the streaming structure above (bcbf_potrs without alpha, n <= 4):
// returns BCBF_OK after launching, or 1 when the shape is not taken here (the caller falls back to potrs_kernel).
template <typename T>
int launch_forward_stream(const T* Lop, const T* Xdot, const T* UH, const T* M0, T* Vw, int Bt, int N, int n, int cu,
                          void* stream) {
    constexpr int V = Vec<T>::V;
    const int Np = round_up(N, NB);
    const int threads = round_up(Np / V / 2, 64);
    if (n < 1 || n > 4 || cu < 1 || cu > BCBF_MAX_CTRL_DIM + 1 || threads > (sizeof(T) == 8 ? 512 : 256)) return 1;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(Bt), block(threads);
#define BCBF_FS_LAUNCH(CC) hipLaunchKernelGGL((posterior_step_kernel<T, CC, 4, 0, 1, true>), grid, block, 0, st, Lop, (const T*)nullptr, Xdot, UH, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr, M0, (const T*)nullptr, (const T*)nullptr, (T*)nullptr, (T*)nullptr, Vw, (T*)nullptr, (T*)nullptr, 0, N, Np, n, (const T*)nullptr, cu, Np, N, 0, (const T*)nullptr)
    switch (n) {
        case 1: case 2: BCBF_FS_LAUNCH(2); break;
        case 3: BCBF_FS_LAUNCH(3); break;
        default: BCBF_FS_LAUNCH(4); break;
    }
#undef BCBF_FS_LAUNCH
    return check_launch("potrs_stream");
}
template int launch_forward_stream<float>(const float*, const float*, const float*, const float*, float*, int, int, int, int, void*);
template int launch_forward_stream<double>(const double*, const double*, const double*, const double*, double*, int, int, int, int, void*);

}  // namespace bcbf

extern "C" int bcbf_posterior_step_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                       const float* ell, const float* s2, const float* Bm, const float* M0,
                                       const float* xq, const float* jitter2, float* Mk, float* Bk,
                                       int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, nullptr, 0, Bt, N, n, m, stream);
}
extern "C" int bcbf_posterior_step_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                       const double* ell, const double* s2, const double* Bm, const double* M0,
                                       const double* xq, const double* jitter2, double* Mk, double* Bk,
                                       int Bt, int N, int n, int m, void* stream) {
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, nullptr, 0, Bt, N, n, m, stream);
}

// Query variant: `shared` != 0 -> all Bt queries go against ONE GP (the reference's own batched
// API, custom_predict with b test points, control_affine_model.py:536,1051); W (optional,
// [Bt, Np, C] with Np = N rounded up to 32) = L^-1 Phi(x_b) lets the caller form cross-covariances
// between different queries, B_k(x,x') = k(x,x') Bm - W(x)'W(x')  (:586, :1079-1088).
extern "C" int bcbf_posterior_query_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                        const float* ell, const float* s2, const float* Bm, const float* M0,
                                        const float* xq, const float* jitter2, float* Mk, float* Bk, float* W,
                                        int shared, int Bt, int N, int n, int m, void* stream) {
    // many queries of one GP: the factor stays in cache and the solve is compute bound -> matrix-core kernel
    if (shared && Bt >= 16 && bcbf::posterior_shared_fits(N, n, m))
        return bcbf_posterior_shared_f32(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, Bt, N, n, m, stream);
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, shared, Bt, N, n, m, stream);
}
extern "C" int bcbf_posterior_query_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                        const double* ell, const double* s2, const double* Bm, const double* M0,
                                        const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                        int shared, int Bt, int N, int n, int m, void* stream) {
    // fp64 matrix cores with the solution kept in registers (posterior_shared_reg.hip): N <= 512, n <= 4
    if (shared && Bt >= 16 && bcbf::posterior_shared64_fits(N, n, m))
        return bcbf_posterior_shared_f64(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, Bt, N, n, m, stream);
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, shared, Bt, N, n, m, stream);
}

// Same query with the data kernel k = s2 (exp(-1/2 |(x-x')/ell|^2) + lin x'x'), lin[Bt] (or [1] when shared): the
// RBF + Linear kernel of the reference's CoGP comparator (ControlAffineVectorGP, control_affine_model.py:1106-1126),
// whose (N n)-sample system is this kernel's MVGP structure with expanded inputs.
extern "C" int bcbf_posterior_query_rbflin_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                               const float* ell, const float* s2, const float* lin, const float* Bm,
                                               const float* M0, const float* xq, const float* jitter2, float* Mk,
                                               float* Bk, float* W, int shared, int Bt, int N, int n, int m,
                                               void* stream) {
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, shared, Bt, N, n, m, stream, nullptr, nullptr, lin);
}
extern "C" int bcbf_posterior_query_rbflin_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                               const double* ell, const double* s2, const double* lin, const double* Bm,
                                               const double* M0, const double* xq, const double* jitter2, double* Mk,
                                               double* Bk, double* W, int shared, int Bt, int N, int n, int m,
                                               void* stream) {
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, shared, Bt, N, n, m, stream, nullptr, nullptr, lin);
}

// Jets: value + first x-derivatives of the posterior factors at one query per instance (or per query of
// a shared GP).  CT = (1+m)(1+n) columns [Phi, dPhi/dx_1..dPhi/dx_n]:  G[Bt,CT,CT] = Wj'Wj,
// Mj[Bt,n,CT] = Vw'Wj (so Mk = M0' + Mj[:, :C], dMk/dx_d = Mj[:, (1+d)C:(2+d)C]).  Feeds bcbf_cbc2_terms.
// Wj (optional, [Bt, Np, CT]) = L^-1 [Phi, dPhi/dx_d]: lets the caller form the derivative kernels between two
// DIFFERENT states, d/dx d/dx' B_k(x, x') = d2k/dxdx' Bm - dW_d(x)'dW_e(x')  (GradientGP.knl(x, x'), gp_algebra.py:355-393).
// Replaces autograd through custom_predict in GradientGP (gp_algebra.py:340-402).  Every (n <= 4, m <= 3).
extern "C" int bcbf_posterior_jets_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                       const float* ell, const float* s2, const float* Bm, const float* M0,
                                       const float* xq, float* Mk, float* Bk, float* G, float* Mj, float* Wj,
                                       int shared, int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!G || !Mj) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, nullptr, Mk, Bk, Wj, shared, Bt, N, n, m, stream, G, Mj);
}
extern "C" int bcbf_posterior_jets_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                       const double* ell, const double* s2, const double* Bm, const double* M0,
                                       const double* xq, double* Mk, double* Bk, double* G, double* Mj, double* Wj,
                                       int shared, int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!G || !Mj) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, nullptr, Mk, Bk, Wj, shared, Bt, N, n, m, stream, G, Mj);
}

// Queries on capacity-reserving storage (bcbf_gp_reserve: the operator laid out for Ncap points, X / UH B / Vw with Ncap
// rows per instance, the first N live): one query per instance, the streaming kernel.  W (optional) [Bt, Np, 1+m] with
// Np = N rounded up to 32.  The online path's forward solve and its control-step posterior.
extern "C" int bcbf_posterior_query_reserved_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                                 const float* ell, const float* s2, const float* Bm, const float* M0,
                                                 const float* xq, const float* jitter2, float* Mk, float* Bk, float* W,
                                                 int Bt, int N, int Ncap, int n, int m, void* stream) {
    if (Ncap < N) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, 0, Bt, N, n, m, stream,
                                              nullptr, nullptr, nullptr, Ncap);
}
extern "C" int bcbf_posterior_query_reserved_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                                 const double* ell, const double* s2, const double* Bm, const double* M0,
                                                 const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                                 int Bt, int N, int Ncap, int n, int m, void* stream) {
    if (Ncap < N) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, 0, Bt, N, n, m, stream,
                                               nullptr, nullptr, nullptr, Ncap);
}

// ... with the opt-in data kernels (kernel_kind of bcbf_common.h: 0 RBF, 1 Matern-5/2, 2 RBF x Matern-5/2) -- states grown by
// bcbf_gp_append_reserved_kind / bcbf_gp_tail_step_kind
extern "C" int bcbf_posterior_query_reserved_kind_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                                      const float* ell, const float* s2, const float* Bm, const float* M0,
                                                      const float* xq, const float* jitter2, float* Mk, float* Bk, float* W,
                                                      int Bt, int N, int Ncap, int n, int m, int kernel_kind, void* stream) {
    if (Ncap < N || kernel_kind < 0 || kernel_kind >= bcbf::BCBF_KINDS) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, 0, Bt, N, n, m, stream,
                                              nullptr, nullptr, nullptr, Ncap, kernel_kind);
}
extern "C" int bcbf_posterior_query_reserved_kind_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                                      const double* ell, const double* s2, const double* Bm, const double* M0,
                                                      const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                                      int Bt, int N, int Ncap, int n, int m, int kernel_kind, void* stream) {
    if (Ncap < N || kernel_kind < 0 || kernel_kind >= bcbf::BCBF_KINDS) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, 0, Bt, N, n, m, stream,
                                               nullptr, nullptr, nullptr, Ncap, kernel_kind);
}

// The same queries with the Matern-5/2 data kernel  k = s2 (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r),  r^2 = sum_d ((x_d - x'_d) / ell_d)^2
// (gpytorch MaternKernel(nu = 2.5, ard) under ScaleKernel).  OPT-IN and parity unpinned: the reference has no Matern
// (and the derivative jets of the same kernel, bcbf_posterior_jets_matern52: d k / d x_d = -(5/3) s2 (1 + a) exp(-a) (x_d - x'_d) / ell_d^2)
extern "C" int bcbf_posterior_jets_matern52_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                                const float* ell, const float* s2, const float* Bm, const float* M0,
                                                const float* xq, float* Mk, float* Bk, float* G, float* Mj, float* Wj,
                                                int shared, int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!G || !Mj) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, nullptr, Mk, Bk, Wj, shared, Bt, N, n, m, stream, G, Mj,
                                              nullptr, 0, 1);
}
extern "C" int bcbf_posterior_jets_matern52_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                                const double* ell, const double* s2, const double* Bm, const double* M0,
                                                const double* xq, double* Mk, double* Bk, double* G, double* Mj, double* Wj,
                                                int shared, int Bt, int N, int n, int m, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!G || !Mj) return BCBF_EINVAL;
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, nullptr, Mk, Bk, Wj, shared, Bt, N, n, m, stream, G, Mj,
                                               nullptr, 0, 1);
}
// kernel (its data kernels are RBF and RBF + Linear); offered because the task statement names an "RBF x Matern" kernel
// build.  Streaming kernel only (shared != 0: one GP, many queries, from cache); K_b: bcbf_kb_build_matern52 + bcbf_potrf.
extern "C" int bcbf_posterior_query_matern52_f32(const float* Lop, const float* Vw, const float* X, const float* UHB,
                                                 const float* ell, const float* s2, const float* Bm, const float* M0,
                                                 const float* xq, const float* jitter2, float* Mk, float* Bk, float* W,
                                                 int shared, int Bt, int N, int n, int m, void* stream) {
    // many queries of one model: the matrix-core kernels (as bcbf_posterior_query does for the RBF kernel)
    if (shared && Bt >= 16 && bcbf::posterior_shared_fits(N, n, m))
        return bcbf_posterior_shared_matern52_f32(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, Bt, N, n, m, stream);
    return bcbf::launch_posterior_step<float>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, shared, Bt, N, n, m, stream,
                                              nullptr, nullptr, nullptr, 0, 1);
}
extern "C" int bcbf_posterior_query_matern52_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                                 const double* ell, const double* s2, const double* Bm, const double* M0,
                                                 const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                                 int shared, int Bt, int N, int n, int m, void* stream) {
    if (shared && Bt >= 16 && bcbf::posterior_shared64_fits(N, n, m))
        return bcbf_posterior_shared_matern52_f64(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, Bt, N, n, m, stream);
    return bcbf::launch_posterior_step<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, shared, Bt, N, n, m, stream,
                                               nullptr, nullptr, nullptr, 0, 1);
}

// The same two entry points for the PRODUCT kernel RBF x Matern-5/2 (kind 2 of bcbf_common.h: kernel_shape; opt-in, no
// reference counterpart)
#define BCBF_RBFM52_ENTRIES(T, SUF)                                                                                              \
    extern "C" int bcbf_posterior_jets_rbfm52_##SUF(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2, \
                                                    const T* Bm, const T* M0, const T* xq, T* Mk, T* Bk, T* G, T* Mj, T* Wj,     \
                                                    int shared, int Bt, int N, int n, int m, void* stream) {                   \
        if (Bt <= 0) return BCBF_OK;                                                                                           \
        if (!G || !Mj) return BCBF_EINVAL;                                                                                     \
        return bcbf::launch_posterior_step<T>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, nullptr, Mk, Bk, Wj, shared, Bt, N, n, m,    \
                                              stream, G, Mj, nullptr, 0, 2);                                                   \
    }                                                                                                                          \
    extern "C" int bcbf_posterior_query_rbfm52_##SUF(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell,          \
                                                     const T* s2, const T* Bm, const T* M0, const T* xq, const T* jitter2,      \
                                                     T* Mk, T* Bk, T* W, int shared, int Bt, int N, int n, int m, void* stream) { \
        if (shared && Bt >= 16 && (sizeof(T) == 8 ? bcbf::posterior_shared64_fits(N, n, m) : bcbf::posterior_shared_fits(N, n, m))) \
            return bcbf_posterior_shared_rbfm52_##SUF(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, Bt, N, n, m, stream); \
        return bcbf::launch_posterior_step<T>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, shared, Bt, N, n, m,     \
                                              stream, nullptr, nullptr, nullptr, 0, 2);                                        \
    }
BCBF_RBFM52_ENTRIES(float, f32)
BCBF_RBFM52_ENTRIES(double, f64)
#undef BCBF_RBFM52_ENTRIES

// Two queries per instance on ONE pass over its factor (reserved storage): xq[Bt,n] and xq2[Bt,n] -> Mk2[Bt,2,n,1+m],
// Bk2[Bt,2,1+m,1+m], W2[Bt,2,Np,1+m] (slot 0 = xq, slot 1 = xq2).  Internal to bcbf_gp_append_reserved (solve.hip).
namespace bcbf {
template <typename T>
int launch_posterior_pair_reserved(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2, const T* Bm,
                                   const T* M0, const T* xq, const T* xq2, T* Mk2, T* Bk2, T* W2, int Bt, int N, int Ncap, int n,
                                   int m, void* stream, int kind) {
    return launch_posterior_step<T>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, nullptr, Mk2, Bk2, W2, 0, Bt, N, n, m, stream, nullptr,
                                    nullptr, nullptr, Ncap, kind, xq2);
}
// The online path's fused pass: posterior (Mk, Bk) at xq AND the append's column l = L^-1 (k(X, x_new) o (UH B uh_new)) with its
// sums (lsum[Bt, 1 + n] = l'l, Vw'l) on ONE pass over every instance's factor (reserved storage, capacity Ncap)
template <typename T>
int launch_posterior_query_column_reserved(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2,
                                           const T* Bm, const T* M0, const T* xq, const T* x_new, const T* uh_new, T* Mk, T* Bk,
                                           T* lvec, T* lsum, int Bt, int N, int Ncap, int n, int m, void* stream, T* Wfull, int Lcap,
                                           int kind) {
    // Wfull (optional, [Bt, Np, 1 + m + 1]): every solved column of the pass, row-major per row (what the tail step continues from)
    // Lcap (0: Ncap): the capacity the OPERATOR is laid out for where it differs from the arrays' -- a packed operator of exactly N
    // points (bcbf_refit's output) beside arrays of Ncap rows: the tail step's window, which never grows in place
    if (Bt <= 0) return BCBF_OK;
    if (n < 1 || n > 4 || m < 1 || m > BCBF_MAX_CTRL_DIM || Ncap < N || kind < 0 || kind >= BCBF_KINDS) return BCBF_EINVAL;
    constexpr int V = Vec<T>::V;
    const int Np = round_up(N, NB), Nl = round_up(Lcap ? Lcap : Ncap, NB);
    const int threads = round_up(Np / V / 2, 64);
    if (threads > (sizeof(T) == 8 ? 512 : 256) || Nl < Np) return BCBF_EINVAL;
    const size_t px_lds = ((lvec ? Np : 0) + (Wfull ? (size_t)Np * (m + 2) : 0)) * sizeof(T);      // (the columns kept in LDS until the pass ends)
    if (px_lds > 100 * 1024) return BCBF_EINVAL;                    // (beside ~1 KB of static LDS; gfx950: 160 KB per workgroup)
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(Bt), block(threads);
#define BCBF_PX_LAUNCH_(CC, XKV) do { if (px_lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)posterior_step_kernel<T, CC, 4, 0, 1, false, 1, false, XKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)px_lds); hipLaunchKernelGGL((posterior_step_kernel<T, CC, 4, 0, 1, false, 1, false, XKV>), grid, block, ((lvec ? Np : 0) + (Wfull ? (size_t)Np * (CC + 1) : 0)) * sizeof(T), st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, (const T*)nullptr, Mk, Bk, lvec, lsum, Wfull, 0, N, Np, n, (const T*)nullptr, Bt, Nl, Ncap, kind, x_new, uh_new); } while (0)
#define BCBF_PX_LAUNCH(CC) if (kind != 0) BCBF_PX_LAUNCH_(CC, true); else BCBF_PX_LAUNCH_(CC, false)
    switch (m) {
        case 1: BCBF_PX_LAUNCH(2); break;
        case 2: BCBF_PX_LAUNCH(3); break;
        default: BCBF_PX_LAUNCH(4); break;
    }
#undef BCBF_PX_LAUNCH
#undef BCBF_PX_LAUNCH_
    return check_launch("posterior_query_column");
}
template int launch_posterior_query_column_reserved<float>(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, int, int, int, int, int, void*, float*, int, int);
template int launch_posterior_query_column_reserved<double>(const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*, double*, double*, int, int, int, int, int, void*, double*, int, int);
template int launch_posterior_pair_reserved<float>(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int, int, int, void*, int);
template int launch_posterior_pair_reserved<double>(const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*, double*, int, int, int, int, int, void*, int);
}  // namespace bcbf
