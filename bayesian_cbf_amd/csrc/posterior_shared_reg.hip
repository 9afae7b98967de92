// Regime S with the solution held in registers: many queries against ONE GP on the matrix cores, fp64
// (v_mfma_f64_16x16x4_f64; the reference's unicycle module is fp64, unicycle_move_to_pose.py:50) and fp32
// (v_mfma_f32_16x16x4_f32).  Replaces custom_predict with b test points (control_affine_model.py:536-602, 1051-1091).
//
// Blocked forward substitution W = L^-1 Phi with 32-row blocks, one wave = 4 queries = 16 right-hand-side columns (4 per
// query, unused ones zero), one wave per SIMD, four waves per workgroup:
//   * the accumulator of a 16x16x4 MFMA holds, in lane (j = lane & 15, g = lane >> 4), register r, the element
//     (row g + 4r [fp64] | 4g + r [fp32], column j) of a 16-row tile -- and the B operand of the next MFMA wants
//     B[k = g][n = j].  With k-step s = (tile u, register r) covering the block rows blkrow(s, g) = 16u + 4r + g [fp64] |
//     16u + 4g + r [fp32], an accumulator register IS a B operand: no shuffle, no LDS round trip;
//   * so W never leaves the registers: wreg[K][s] (8 values per 32-row block, pinned to the accumulation registers, of
//     which a wave alone on its SIMD has 256) is written by the diagonal step of block K and read as the B operand of
//     every later tile (I, K).  fp64 at N = 512 would need all 256: blocks 12-14 (6 tiles read them) sit in a per-lane
//     LDS slab instead.  The loops over blocks are compile-time loops: every wreg index is a constant;
//   * A operands: every 32x32 tile of the packed operator is fetched ONCE per workgroup (16-byte buffer loads) into a
//     ring of three LDS tiles and read by all four waves with explicit ds_read, issued a whole tile ahead of the MFMAs
//     that consume them (bank-conflict-free image: see gload);  diagonal step: A = the stored inverse (full-tile copy),
//     B = Phi - acc.  One barrier per tile.
//   History, fp64, N = 512, 4096 queries: streaming VALU kernel 0.27 ms -> per-wave 8-byte operand loads from L2 0.145
//   -> tiles shared through LDS 0.137 -> W pinned to AGPRs (no spills) + next tile's operands prefetched 0.095 -> reads
//   chained between the MFMAs, Phi tile beside the previous tile's MFMAs 0.092 ms
//   (a bare loop of the fp64 MFMA sustains 31.4 ns per instruction with one wave per SIMD: 66 us for the 2112 of a wave).
// The explicit prefetch (asm issue, asm wait) is only sound while the register allocator does not spill an operand
// between the two: build.py compiles this file with -Rpass-analysis=kernel-resource-usage and refuses a build whose
// one-wave kernels report a non-zero scratch size.  fp32 with more queries than one wave per SIMD holds takes a second
// instantiation (OCC = 2: 252 registers, plain operand loads at their use, two workgroups per CU): 63 -> 74 TFLOP/s at
// 16384 queries.  State dimensions n <= 4 (wider ones stream); N <= 512; beyond that, and for
// few queries, posterior_shared.hip (fp32, W slab in LDS) or the streaming kernel (posterior_step.hip) answer.
#include "bcbf_common.h"
#include "diag_tile64.h"      // LdsDouble

namespace bcbf {

using f64x4r = __attribute__((__vector_size__(4 * sizeof(double)))) double;
using f32x4r = __attribute__((__vector_size__(4 * sizeof(float)))) float;
using u32x4r = __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned;

#ifndef BCBF_PSR_OCC2
#define BCBF_PSR_OCC2 1
#endif
#ifndef BCBF_PSR_QW5
#define BCBF_PSR_QW5 1
#endif
constexpr int PSR_MAXBLK = 16;           // N <= 512

// compile-time loop: the body sees a constant index (every wreg[][] subscript must be one, or the array leaves the
// register file for scratch memory -- "#pragma unroll" alone is a request the optimizer declines for 2000-MFMA bodies)
template <int I> struct Ic { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(Ic<B>{}); static_for<B + 1, E>(f); }
}

// block row of tile t in the order (0,0), (1,0), (1,1), (2,0), ...
constexpr int tile_row(int t) { int I = 0; while ((I + 1) * (I + 2) / 2 <= t) ++I; return I; }

template <typename T> struct PSR;
template <> struct PSR<double> {
    using acc_t = f64x4r;
    static constexpr int REGBLK = 12;                 // W blocks in registers (16 VGPRs each); 12..14 in the LDS slab
    static constexpr int PIECES = 2;                  // 16-byte pieces of a tile per thread
    static constexpr int PPC = 16;                    // pieces per tile column
    static constexpr int TILE_BYTES = NB * NB * 8;
    // column of the tile that k-step s reads in lane group g (= block row of the W value in register s): colstep(s) + lanecol * g
    __host__ __device__ static constexpr int colstep(int s) { return 4 * s; }
    static constexpr int LANECOL = 1;
    __device__ static acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static double exp_(double x) { return exp(x); }
};
template <> struct PSR<float> {
    using acc_t = f32x4r;
    static constexpr int REGBLK = PSR_MAXBLK - 1;     // all of them (8 VGPRs each)
    static constexpr int PIECES = 1;
    static constexpr int PPC = 8;
    static constexpr int TILE_BYTES = NB * NB * 4;
    __host__ __device__ static constexpr int colstep(int s) { return 16 * (s >> 2) + (s & 3); }
    static constexpr int LANECOL = 4;
    __device__ static acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static float exp_(float x) { return __expf(x); }
};

// A operands of one tile: 16 explicit LDS reads (a[u'][s] at byte offset OFF + column colstep(s) from the lane's two
// bases), and the wait that makes their results usable
template <int OFF> __device__ __forceinline__ void lds_get16(double (&a)[2][8], unsigned a0, unsigned a1) {
#define BCBF_RD(u_, s_, addr) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[u_][s_]) : "v"(addr), "n"(OFF + NB * 8 * PSR<double>::colstep(s_)))
    BCBF_RD(0, 0, a0); BCBF_RD(1, 0, a1); BCBF_RD(0, 1, a0); BCBF_RD(1, 1, a1);
    BCBF_RD(0, 2, a0); BCBF_RD(1, 2, a1); BCBF_RD(0, 3, a0); BCBF_RD(1, 3, a1);
    BCBF_RD(0, 4, a0); BCBF_RD(1, 4, a1); BCBF_RD(0, 5, a0); BCBF_RD(1, 5, a1);
    BCBF_RD(0, 6, a0); BCBF_RD(1, 6, a1); BCBF_RD(0, 7, a0); BCBF_RD(1, 7, a1);
#undef BCBF_RD
}
template <int OFF> __device__ __forceinline__ void lds_get16(float (&a)[2][8], unsigned a0, unsigned a1) {
#define BCBF_RD(u_, s_, addr) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[u_][s_]) : "v"(addr), "n"(OFF + NB * 4 * PSR<float>::colstep(s_)))
    BCBF_RD(0, 0, a0); BCBF_RD(1, 0, a1); BCBF_RD(0, 1, a0); BCBF_RD(1, 1, a1);
    BCBF_RD(0, 2, a0); BCBF_RD(1, 2, a1); BCBF_RD(0, 3, a0); BCBF_RD(1, 3, a1);
    BCBF_RD(0, 4, a0); BCBF_RD(1, 4, a1); BCBF_RD(0, 5, a0); BCBF_RD(1, 5, a1);
    BCBF_RD(0, 6, a0); BCBF_RD(1, 6, a1); BCBF_RD(0, 7, a0); BCBF_RD(1, 7, a1);
#undef BCBF_RD
}
// ... the same reads a quarter at a time (k-steps 2G, 2G+1 of both tiles), CHAINED to an accumulator: the first read
// names `acc` as an in/out operand (an accumulation register tuple, "+a"), so it is ordered after the MFMA that last
// wrote that accumulator and before the next one that reads it -- the source order of reads and MFMAs below is the issue
// order.  (Volatile asm statements keep their order among themselves only: left unchained, the scheduler runs a tile's 16
// MFMAs first and sinks all 16 reads of the next tile to the end of the step, against the wait.  sched_barrier does pin
// them but splits the scheduling region: 1500 spills in fp64.  Measured gain of the chaining: 3 % fp32, 1 % fp64.)
template <int OFF, int G, typename A> __device__ __forceinline__ void lds_get4(double (&a)[2][8], unsigned a0, unsigned a1, A& acc) {
#define BCBF_RD(u_, s_, addr) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[u_][s_]) : "v"(addr), "n"(OFF + NB * 8 * PSR<double>::colstep(s_)))
    asm volatile("ds_read_b64 %0, %2 offset:%3" : "=v"(a[0][2 * G]), "+a"(acc) : "v"(a0), "n"(OFF + NB * 8 * PSR<double>::colstep(2 * G)));
    BCBF_RD(1, 2 * G, a1); BCBF_RD(0, 2 * G + 1, a0); BCBF_RD(1, 2 * G + 1, a1);
#undef BCBF_RD
}
template <int OFF, int G, typename A> __device__ __forceinline__ void lds_get4(float (&a)[2][8], unsigned a0, unsigned a1, A& acc) {
#define BCBF_RD(u_, s_, addr) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[u_][s_]) : "v"(addr), "n"(OFF + NB * 4 * PSR<float>::colstep(s_)))
    asm volatile("ds_read_b32 %0, %2 offset:%3" : "=v"(a[0][2 * G]), "+a"(acc) : "v"(a0), "n"(OFF + NB * 4 * PSR<float>::colstep(2 * G)));
    BCBF_RD(1, 2 * G, a1); BCBF_RD(0, 2 * G + 1, a0); BCBF_RD(1, 2 * G + 1, a1);
#undef BCBF_RD
}
template <typename A> __device__ __forceinline__ void chain(A& acc) { asm volatile("" : "+a"(acc)); }
template <typename T, typename A> __device__ __forceinline__ void lds_wait16(T (&a)[2][8], A& acc) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[0][6]), "+v"(a[0][7]),
                   "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[1][4]), "+v"(a[1][5]), "+v"(a[1][6]), "+v"(a[1][7]),
                   "+a"(acc));
}
template <typename T> __device__ __forceinline__ void lds_wait16(T (&a)[2][8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[0][6]), "+v"(a[0][7]),
                   "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[1][4]), "+v"(a[1][5]), "+v"(a[1][6]), "+v"(a[1][7]));
}

// quad broadcast: lane c of every quad -> all four lanes (one DPP move per 32 bits)
template <int CTRL> __device__ inline float dpp_bc(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ inline double dpp_bc(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

// QW: queries per wave -- 4 (one per lane quad, 4 columns each: component 3 of the unicycle's C = 3 is a zero column, two of
// the pendulum's C = 2) or the dense packing, C adjacent columns per query: 5 queries for C = 3 (column 15 idle: 15 of the 16
// MFMA columns carry a query instead of 12), 8 for C = 2 (16 instead of 8).  Chosen by the launcher when the queries no longer
// fit one wave per SIMD at QW = 4 (a fifth / half fewer waves for the same queries; with one round of waves either way the
// quads' cheaper broadcasts win).
template <typename T, int C, int NS, int OCC, int QW = 4, int KIND = 0>
__global__ void __launch_bounds__(256, OCC)
posterior_shared_reg_kernel(const T* __restrict__ Lop, const T* __restrict__ Vw, const T* __restrict__ X,
                            const T* __restrict__ UHB, const T* __restrict__ ell, const T* __restrict__ s2p,
                            const T* __restrict__ Bm, const T* __restrict__ M0, const T* __restrict__ xq,
                            const T* __restrict__ jitter2, T* __restrict__ Mk, T* __restrict__ Bk,
                            T* __restrict__ Wout, int nq, int N, int Np, int n) {
    // KIND: data kernel -- 0 = RBF (the reference's), 1 = Matern-5/2 (opt-in, bcbf_posterior_shared_matern52).  A template
    // parameter: as a run-time switch it cost the RBF instantiations 6-10 % (fp64 4096 queries 0.0846 -> 0.0895 ms); the Matern
    // form is compiled for the base packing only (one wave per SIMD, four queries per wave)
    using P = PSR<T>;
    using acc_t = typename P::acc_t;
    constexpr int V = Vec<T>::V, ES = (int)sizeof(T);
    static_assert(QW == 4 || (QW == 5 && C == 3) || (QW == 8 && C == 2), "dense packing: C adjacent columns per query");
    constexpr int STR = QW == 4 ? 4 : C;               // columns from one query to the next
    constexpr int REGBLK = P::REGBLK, SLABBLK = PSR_MAXBLK - 1 - REGBLK;
    constexpr int TILE = NB * NB;                      // elements per tile buffer
    extern __shared__ double smem_psr[];
    const int wave = threadIdx.x >> 6;                 // blockDim.x == 256: all four waves stage tiles
    T* Ts = reinterpret_cast<T*>(smem_psr);            // three 32x32 operator tiles (ring) first: 16-byte aligned
    T* Xs = Ts + 3 * TILE;                             // [Np][NS]  (state dim padded to NS with zeros)
    T* Us = Xs + (size_t)Np * NS;                      // [Np][C]   (rows >= N are zero: padded rows contribute nothing)
    T* Vs = Us + (size_t)Np * C;                       // [Np][NS]
    T* Wl = Vs + (size_t)Np * NS + (size_t)wave * (SLABBLK * 8 * 64) + (threadIdx.x & 63);   // this lane's slots of W blocks >= REGBLK
    for (int i = threadIdx.x; i < Np * NS; i += blockDim.x) {
        const int row = i / NS, d = i - row * NS;
        const bool ok = row < N && d < n;
        Xs[i] = ok ? X[(size_t)row * n + d] : T(0);
        Vs[i] = ok ? Vw[(size_t)row * n + d] : T(0);
    }
    for (int i = threadIdx.x; i < Np * C; i += blockDim.x) Us[i] = i < N * C ? UHB[i] : T(0);

    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int ql = j / STR, c = j - STR * ql;          // query slot in the wave, component
    const bool slot = j < QW * STR;                    // (QW = 5: column 15 belongs to no query)
    const int q = (blockIdx.x * 4 + wave) * QW + ql;
    const bool qok = slot && q < nq, cok = slot && c < C;   // (a wave past the end still stages tiles and meets the barriers)
    // value of the lane holding component A of this lane's query: a quad broadcast (QW = 4); among three adjacent lanes
    // (QW = 5) the row shift by A - c
    auto gb = [&](auto ac, T v) -> T {
        constexpr int A = decltype(ac)::value;
        if constexpr (QW == 4) return dpp_bc<0x55 * A>(v);
        else if constexpr (STR == 2) {
            if constexpr (A == 0) { const T m1 = dpp_bc<0x111>(v); return c == 0 ? v : m1; }
            else { const T p1 = dpp_bc<0x101>(v); return c == 0 ? p1 : v; }
        }
        else if constexpr (A == 0) { const T m1 = dpp_bc<0x111>(v), m2 = dpp_bc<0x112>(v); return c == 0 ? v : (c == 1 ? m1 : m2); }
        else if constexpr (A == 1) { const T p1 = dpp_bc<0x101>(v), m1 = dpp_bc<0x111>(v); return c == 0 ? p1 : (c == 1 ? v : m1); }
        else { const T p2 = dpp_bc<0x102>(v), p1 = dpp_bc<0x101>(v); return c == 0 ? p2 : (c == 1 ? p1 : v); }
    };
    const int qq = qok ? q : nq - 1;

    T xqr[NS], iell[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        xqr[d] = d < n ? xq[(size_t)qq * n + d] : T(0);
        iell[d] = d < n ? T(1) / ell[d] : T(0);
    }
    const T s2 = s2p[0];
    T gram[C], mk[NS];
#pragma unroll
    for (int a = 0; a < C; ++a) gram[a] = T(0);
#pragma unroll
    for (int d = 0; d < NS; ++d) mk[d] = T(0);
    const int cc = cok ? c : 0;
    const T cmask = cok ? T(1) : T(0);
    const int nblk = Np / NB;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(Lop), 0, (int)(lop_elems<V>(Np) * sizeof(T)), 0x00020000);
    // Tile (I, K) of the operator, L[32I + r][32K + col] (K < I: the packed off-diagonal part, element (row, col) at
    // lop_base(col) + row; K == I: the full-tile copy of inv(L_II), column stride 32), is fetched ONCE per workgroup in
    // 16-byte pieces (PPC per column; thread -> column tid / PPC (+ 16 for its second piece in fp64), piece tid % PPC).
    // LDS image: column-major, 32 elements per column, with the two 16-row halves of a column swapped when the lane
    // group that reads it is odd -- the 32 lanes an LDS read serves in one cycle (lane groups g, g+1: two columns, 16
    // rows each) then cover every bank exactly once (ds_read_b64: 64 banks; ds_read_b32: 32).
    constexpr int PPC = P::PPC, EPP = 16 / ES;        // elements per piece
    const int pc0 = threadIdx.x / PPC, pp = threadIdx.x % PPC;
    auto gload = [&](u32x4r (&r)[P::PIECES], int I, int K) {
        const bool isdiag = K == I;
        const int stride = isdiag ? NB : Np - NB * (K + 1);                   // column stride inside block column K
        const int base = isdiag ? lop_dfull_block(I, Np) : lop_base<V>(K * NB, Np) + I * NB;   // element (row 0 of the tile, col 0)
        const int voff = (pc0 * stride + EPP * pp) * ES;
#pragma unroll
        for (int h = 0; h < P::PIECES; ++h)
            r[h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (base + 16 * h * stride) * ES, 0);
    };
    // the group that reads column col is (col / LANECOL) & 3; its parity selects the half swap
    const int wr_off = pc0 * NB + ((pp ^ (((pc0 / P::LANECOL) & 1) * (PPC / 2))) * EPP);   // (second piece: 16 columns further)
    auto lds_put = [&](T* tb, const u32x4r (&r)[P::PIECES]) {
#pragma unroll
        for (int h = 0; h < P::PIECES; ++h) *reinterpret_cast<u32x4r*>(tb + wr_off + 16 * NB * h) = r[h];
    };
    // a[u'][s] = tile[row 16u' + j][col colstep(s) + LANECOL g]
    const int rd_off0 = P::LANECOL * g * NB + (j ^ ((g & 1) << 4));
    const int rd_off1 = P::LANECOL * g * NB + ((16 | j) ^ ((g & 1) << 4));
    const unsigned ts_a0 = (unsigned)(size_t)(__attribute__((address_space(3))) T*)(Ts + rd_off0);
    const unsigned ts_a1 = (unsigned)(size_t)(__attribute__((address_space(3))) T*)(Ts + rd_off1);

    T wreg[REGBLK][8];                                 // W_K: block rows colstep(s) + LANECOL g at index s, column j
    auto wget = [&](auto Kc, int s_) -> T {
        constexpr int K_ = decltype(Kc)::value;
        if constexpr (K_ < REGBLK) return wreg[K_][s_];
        else return Wl[((K_ - REGBLK) * 8 + s_) * 64];
    };
    // Pipeline, step t = tile t (tiles in the order (0,0), (1,0), (1,1), (2,0), ...; three LDS buffers):
    //   write tile t+2 (fetched during step t-1) into buffer (t+2) % 3  |  fetch tile t+3 into registers  |
    //   read the A operands of tile t+1 from buffer (t+1) % 3  |  multiply tile t (operands read a step ago)  |  barrier
    auto fetch = [&](u32x4r (&r)[P::PIECES], auto tc) {
        constexpr int t_ = decltype(tc)::value;
        constexpr int I_ = tile_row(t_);
        gload(r, I_, t_ - I_ * (I_ + 1) / 2);
    };
    u32x4r stage[P::PIECES];
    T acur[2][8], anxt[2][8], pend[8];
    fetch(stage, Ic<0>{});
    lds_put(Ts, stage);
    fetch(stage, Ic<1>{});
    lds_put(Ts + TILE, stage);
    fetch(stage, Ic<2>{});
    __syncthreads();                                   // staging of X / UH B / Vw and tiles 0, 1 visible
    if constexpr (OCC == 1) {
        lds_get16<0>(acur, ts_a0, ts_a1);
        lds_wait16(acur);
    }
    // Gram row / mean column of this lane's query from the W tile of block Ib held in pend[]
    auto epilogue = [&](int Ib) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const T v = pend[e];
            const int row = Ib * NB + P::colstep(e) + P::LANECOL * g;
            // lane (component c): gram[k] += W[row][c] W[row][c - k] -- the neighbour k lanes down is component c - k of the SAME
            // query for k <= c (quads and the three-lane groups alike); entries with k > c are never read
            static_for<0, C>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k == 0) gram[0] += v * v;
                else gram[k] += v * dpp_bc<0x110 + k>(v);          // row_shr:k
            });
#pragma unroll
            for (int d = 0; d < NS; ++d) mk[d] += Vs[row * NS + d] * v;        // lane c: (Vw'W)[d][c]
        }
    };
    // Phi tile of block I: phi[s] = k(x_q, X_row) (UH B)[row][c], row = 32I + blkrow(s, g).  The exp of (query, row) is
    // evaluated once, by the lane whose component equals the register index, and broadcast inside the quad: lane c
    // evaluates the row that register 4u + c holds.  It does not depend on the accumulation, so it is formed during the
    // LAST off-diagonal tile of the row, beside that tile's MFMAs (in the diagonal step every MFMA waits for it)
    T phi[8];
    // shape of the data kernel at squared scaled distance d2: exp(-d2 / 2), or Matern-5/2 (1 + a + a^2 / 3) exp(-a), a = sqrt(5 d2)
    auto kshape = [&](T d2) -> T {
        if constexpr (KIND != 0) {                     // Matern-5/2 (1), RBF x Matern-5/2 (2): bcbf_common.h
            T shape, dshape_;
            kernel_shape(KIND, d2, [](T q_) { return P::exp_(q_); }, shape, dshape_);
            return shape;
        }
        return P::exp_(T(-0.5) * d2);
    };
    auto phi_tile = [&](auto Ic_) {
        constexpr int I = decltype(Ic_)::value;
        if constexpr (QW != 4) {
            // lane c evaluates the rows of registers c, c + STR, c + 2 STR, ... (the last one clamped: register 7 twice)
            constexpr int NE = (8 + STR - 1) / STR;
            T km[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int e = min(c + STR * i, 7);
                const int row_m = I * NB + P::colstep(e) + P::LANECOL * g;
                T d2 = T(0);
#pragma unroll
                for (int d = 0; d < NS; ++d) { const T z = (Xs[row_m * NS + d] - xqr[d]) * iell[d]; d2 += z * z; }
                km[i] = s2 * kshape(d2);
            }
            static_for<0, 8>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                phi[e] = gb(Ic<e % STR>{}, km[e / STR]) * Us[(I * NB + P::colstep(e) + P::LANECOL * g) * C + cc] * cmask;
            });
            return;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row_m = I * NB + P::colstep(4 * u) + P::LANECOL * g + (P::colstep(1) - P::colstep(0)) * c;
            T d2 = T(0);
#pragma unroll
            for (int d = 0; d < NS; ++d) { const T z = (Xs[row_m * NS + d] - xqr[d]) * iell[d]; d2 += z * z; }
            const T kmine = s2 * kshape(d2);
            const T kk[4] = {dpp_bc<0x00>(kmine), dpp_bc<0x55>(kmine), dpp_bc<0xAA>(kmine), dpp_bc<0xFF>(kmine)};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                phi[4 * u + r] = kk[r] * Us[(I * NB + P::colstep(4 * u + r) + P::LANECOL * g) * C + cc] * cmask;
        }
    };
    static_for<0, PSR_MAXBLK>([&](auto Ict) {
        constexpr int I = decltype(Ict)::value;
        if (I < nblk) {
            acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            // ---- off-diagonal tiles (I, K), K < I: acc += L_IK W_K
            static_for<0, I>([&](auto Kct) {
                constexpr int K = decltype(Kct)::value;
                constexpr int t = I * (I + 1) / 2 + K;
                lds_put(Ts + TILE * ((t + 2) % 3), stage);
                fetch(stage, Ic<t + 3>{});
                if constexpr (OCC == 1) {
                    // the next tile's operand reads ride behind this tile's first four MFMAs on acc0, four reads each
                    static_for<0, 8>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        const T wk = wget(Kct, s);
                        acc0 = P::mfma(acur[0][s], wk, acc0);
                        if constexpr (s < 4) lds_get4<((t + 1) % 3) * P::TILE_BYTES, s>(anxt, ts_a0, ts_a1, acc0);
                        if constexpr (s == 4) chain(acc0);                    // (bounds how far the last reads can sink)
                        acc1 = P::mfma(acur[1][s], wk, acc1);
                    });
                    if constexpr (K == 0 && I > 0) epilogue(I - 1);            // VALU work beside this row's first MFMAs
                    if constexpr (K == I - 1) phi_tile(Ict);                   // ... and beside its last ones
                    lds_wait16(anxt, acc0);
#pragma unroll
                    for (int s = 0; s < 8; ++s) { acur[0][s] = anxt[0][s]; acur[1][s] = anxt[1][s]; }
                } else {
                    // two waves per SIMD: plain operand loads at their use (the other wave's MFMAs cover the round trip),
                    // 16 registers fewer and nothing for the register allocator to get wrong
                    const T* tb = Ts + TILE * (t % 3);
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const T wk = wget(Kct, s);
                        acc0 = P::mfma(tb[rd_off0 + NB * P::colstep(s)], wk, acc0);
                        acc1 = P::mfma(tb[rd_off1 + NB * P::colstep(s)], wk, acc1);
                    }
                    if constexpr (K == 0 && I > 0) epilogue(I - 1);
                }
                __syncthreads();
            });
            constexpr int t = I * (I + 1) / 2 + I;
            lds_put(Ts + TILE * ((t + 2) % 3), stage);
            fetch(stage, Ic<t + 3>{});
            if constexpr (OCC == 1) lds_get16<((t + 1) % 3) * P::TILE_BYTES>(anxt, ts_a0, ts_a1);
            if constexpr (I == 0 || OCC == 2) phi_tile(Ict);   // (one wave per SIMD, later blocks: formed beside the MFMAs of tile (I, I-1))
            if constexpr (OCC == 2) {
                const T* tb = Ts + TILE * (t % 3);
#pragma unroll
                for (int s = 0; s < 8; ++s) { acur[0][s] = tb[rd_off0 + NB * P::colstep(s)]; acur[1][s] = tb[rd_off1 + NB * P::colstep(s)]; }
            }
            // ---- diagonal step: W_I = inv(L_II) (Phi_I - acc)   (inv(L_II) is lower triangular: tile 0 needs k < 16 only)
            acc_t w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const T b_ = phi[s] - ((s >> 2) ? acc1[s & 3] : acc0[s & 3]);
                if (s < 4) w0 = P::mfma(acur[0][s], b_, w0);
                w1 = P::mfma(acur[1][s], b_, w1);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const T v = (e >> 2) ? w1[e & 3] : w0[e & 3];
                pend[e] = v;
                if constexpr (I < REGBLK) { wreg[I][e] = v; asm volatile("" : "+a"(wreg[I][e])); }   // lives in the accumulation registers
                else if constexpr (I < PSR_MAXBLK - 1) Wl[((I - REGBLK) * 8 + e) * 64] = v;
            }
            if (Wout != nullptr && qok && cok) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    Wout[((size_t)q * Np + I * NB + P::colstep(e) + P::LANECOL * g) * C + c] = pend[e];
            }
            if constexpr (OCC == 1) {
                lds_wait16(anxt);
#pragma unroll
                for (int s = 0; s < 8; ++s) { acur[0][s] = anxt[0][s]; acur[1][s] = anxt[1][s]; }
            }
            if (I == nblk - 1) epilogue(I);
            __syncthreads();
        }
    });

    // ---- the four lane groups hold different rows: combine, then write Mk[q][d][c], Bk[q][c][a]
#pragma unroll
    for (int a = 0; a < C; ++a) { gram[a] += __shfl_xor(gram[a], 16, 64); gram[a] += __shfl_xor(gram[a], 32, 64); }
    // gram[k] = G[c][c - k] (k <= c).  Row c of the symmetric block: G[c][a] = gram[c - a] for a <= c, and for a > c the entry
    // G[a][c] = gram[a - c] of the lane a - c further up
    T grow[C];
#pragma unroll
    for (int a = 0; a < C; ++a) grow[a] = T(0);
    static_for<0, C>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const T up = k == 0 ? gram[0] : __shfl(gram[k], (lane + k) & 63, 64);      // lane j + k holds G[c + k][c]
#pragma unroll
        for (int a = 0; a < C; ++a) {
            if (a == c + k) grow[a] = up;
            if (k > 0 && a == c - k) grow[a] = gram[k];
        }
    });
#pragma unroll
    for (int d = 0; d < NS; ++d) { mk[d] += __shfl_xor(mk[d], 16, 64); mk[d] += __shfl_xor(mk[d], 32, 64); }
    if (g == 0 && qok && cok) {
#pragma unroll
        for (int d = 0; d < NS; ++d)
            if (d < n) Mk[((size_t)q * n + d) * C + c] = M0[c * n + d] + mk[d];
#pragma unroll
        for (int a = 0; a < C; ++a) {
            T v = s2 * Bm[c * C + a] - grow[a];
            if (a == c && jitter2 != nullptr) v += jitter2[(size_t)q * C + c];
            Bk[((size_t)q * C + c) * C + a] = v;
        }
    }
}

static int psr_state_dim(int n) { return n <= 2 ? 2 : n; }   // instantiated widths: 2, 3, 4

template <typename T> static size_t psr_lds_bytes(int Np, int n, int m) {
    constexpr int SLABBLK = PSR_MAXBLK - 1 - PSR<T>::REGBLK;
    return ((size_t)3 * NB * NB + (size_t)Np * (2 * psr_state_dim(n) + m + 1) +
            (Np > NB * PSR<T>::REGBLK ? (size_t)4 * SLABBLK * 512 : 0)) * sizeof(T);
}

// N <= 512, n <= 4 (wider states need more registers than the explicit operand prefetch leaves) and the staged copies
// fit in LDS
template <typename T> static bool psr_fits(int N, int n, int m) {
    const int Np = round_up(N, NB);
    return n <= 4 && Np <= NB * PSR_MAXBLK && psr_lds_bytes<T>(Np, n, m) <= 160 * 1024;
}
template <typename T, int C, int NS>
static void launch_psr(size_t lds, hipStream_t st, const T* Lop, const T* Vw, const T* X, const T* UHB,
                       const T* ell, const T* s2, const T* Bm, const T* M0, const T* xq, const T* jitter2, T* Mk, T* Bk,
                       T* W, int nq, int N, int Np, int n, int kind) {
    // fp32 with enough queries for two waves per SIMD (more than 16 per CU-SIMD: > 4096 on 256 CUs): the 256-register
    // allocation, two workgroups per CU, each wave's stalls under the other's MFMAs.  fp64 needs the full file.
    int dev_ = 0, cus = 256;
    (void)hipGetDevice(&dev_);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_);
    // Queries per wave (4, or 5 when C = 3) and waves per SIMD (1, or 2 in fp32): the launch runs in rounds of 4 (8) waves
    // per CU and a round costs the same however full it is, so the cheapest of the four combinations by
    //   rounds x (time of one round relative to the four-query one-wave form: 1.0 | 1.05 with five queries, x 1.7 at two waves per SIMD)
    // (measured, N = 512, us per round: fp32 48 / 50 / 82 / 86, fp64 84 / 88): five queries per wave pay when they save a round --
    // 5120 queries: fp32 0.050 ms (80 TFLOP/s) against 0.082, fp64 0.088 (46 TFLOP/s) against 0.168; 20480: fp32 0.171 against
    // 0.245 -- and lose a twentieth when they do not.
    static const int qw_force = [] { const char* e = getenv("BCBF_PSR_QW"); return e ? atoi(e) : 0; }();      // (development)
    constexpr int QWD = C == 3 ? 5 : C == 2 ? 8 : 4;   // queries per wave of the dense packing
    const bool can5 = QWD != 4 && BCBF_PSR_QW5, can2 = sizeof(T) == 4 && BCBF_PSR_OCC2 && lds <= 64 * 1024;
    bool five = false, two = false;
    double best = 1e30;
    for (int f = 0; f <= (can5 ? 1 : 0); ++f)
        for (int t = 0; t <= (can2 ? 1 : 0); ++t) {
            if (qw_force && (qw_force != 4) != (f == 1) && can5) continue;
            const long w = f ? (nq + QWD - 1) / QWD : (nq + 3) / 4, slots = (long)(t ? 8 : 4) * cus;
            const double cost = (double)((w + slots - 1) / slots) * (f ? 1.05 : 1.0) * (t ? 1.7 : 1.0);
            if (cost < best) { best = cost; five = f; two = t; }
        }
    if (kind != 0) { five = false; two = false; }      // opt-in kernels: the base packing only
    const int waves = five ? (nq + QWD - 1) / QWD : (nq + 3) / 4;
    const dim3 grid((waves + 3) / 4);                  // 256 threads per workgroup, four waves
    auto go_kind = [&](auto kc) {
        constexpr int KD = decltype(kc)::value;
        static int opt_in_m[64] = {0};
        int& lds_opt_in = opt_in_m[dev_ & 63];
        if (lds > 64 * 1024 && (int)lds > lds_opt_in) {
            (void)hipFuncSetAttribute((const void*)posterior_shared_reg_kernel<T, C, NS, 1, 4, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            lds_opt_in = (int)lds;
        }
        hipLaunchKernelGGL((posterior_shared_reg_kernel<T, C, NS, 1, 4, KD>), grid, dim3(256), lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0,
                           xq, jitter2, Mk, Bk, W, nq, N, Np, n);
    };
    if (kind == 1) { go_kind(Ic<1>{}); return; }
    if (kind == 2) { go_kind(Ic<2>{}); return; }
    auto go5 = [&](auto occ, auto qw) {
        constexpr int OCC = decltype(occ)::value, QW = decltype(qw)::value;
        static int opt_in[2][64] = {{0}};           // largest dynamic LDS size opted into, per device
        int& lds_opt_in = opt_in[QW != 4][dev_ & 63];
        if (lds > 64 * 1024 && (int)lds > lds_opt_in) {
            (void)hipFuncSetAttribute((const void*)posterior_shared_reg_kernel<T, C, NS, OCC, QW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            lds_opt_in = (int)lds;
        }
        hipLaunchKernelGGL((posterior_shared_reg_kernel<T, C, NS, OCC, QW>), grid, dim3(256), lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0,
                           xq, jitter2, Mk, Bk, W, nq, N, Np, n);
    };
    auto go = [&](auto occ) {
        if constexpr (QWD != 4 && BCBF_PSR_QW5) { if (five) { go5(occ, Ic<QWD>{}); return; } }
        go5(occ, Ic<4>{});
    };
    if constexpr (sizeof(T) == 4 && BCBF_PSR_OCC2) {
        if (two) go(Ic<2>{}); else go(Ic<1>{});
    } else {
        go(Ic<1>{});
    }
}

// The 27 kernel instantiations take minutes to compile in one translation unit: build.py compiles this file once per
// (precision, component count) with -DBCBF_PSR_PART_T=<0 double | 1 float> -DBCBF_PSR_PART_C=<2|3|4> -- each part
// instantiates launch_psr_c<T, C> and its kernels -- and once with -DBCBF_PSR_PART_BASE for the dispatchers and C entry
// points, which only see the declaration below.  No part macro: everything in one unit.
template <typename T, int C>
void launch_psr_c(int NSp, size_t lds, hipStream_t st, const T* Lop, const T* Vw, const T* X,
                  const T* UHB, const T* ell, const T* s2, const T* Bm, const T* M0, const T* xq, const T* jitter2,
                  T* Mk, T* Bk, T* W, int nq, int N, int Np, int n, int kind);
#ifndef BCBF_PSR_PART_BASE
template <typename T, int C>
void launch_psr_c(int NSp, size_t lds, hipStream_t st, const T* Lop, const T* Vw, const T* X,
                  const T* UHB, const T* ell, const T* s2, const T* Bm, const T* M0, const T* xq, const T* jitter2,
                  T* Mk, T* Bk, T* W, int nq, int N, int Np, int n, int kind) {
#define BCBF_PSR(NSV) launch_psr<T, C, NSV>(lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n, kind)
    switch (NSp) {
#ifndef BCBF_PSR_DEV              // (development: the C = 3, NS = 3 instantiations only)
        case 2: BCBF_PSR(2); break;
        case 4: BCBF_PSR(4); break;
#endif
        case 3: BCBF_PSR(3); break;
        default: break;
    }
#undef BCBF_PSR
}
#endif
#if defined(BCBF_PSR_PART_T) && defined(BCBF_PSR_PART_C)
#if BCBF_PSR_PART_T == 0
#define BCBF_PSR_PT double
#else
#define BCBF_PSR_PT float
#endif
template void launch_psr_c<BCBF_PSR_PT, BCBF_PSR_PART_C>(int, size_t, hipStream_t, const BCBF_PSR_PT*, const BCBF_PSR_PT*,
                                                         const BCBF_PSR_PT*, const BCBF_PSR_PT*, const BCBF_PSR_PT*, const BCBF_PSR_PT*,
                                                         const BCBF_PSR_PT*, const BCBF_PSR_PT*, const BCBF_PSR_PT*, const BCBF_PSR_PT*,
                                                         BCBF_PSR_PT*, BCBF_PSR_PT*, BCBF_PSR_PT*, int, int, int, int, int);
#endif

#if !(defined(BCBF_PSR_PART_T) && defined(BCBF_PSR_PART_C))
bool posterior_shared64_fits(int N, int n, int m) { return psr_fits<double>(N, n, m); }
bool posterior_shared_reg32_fits(int N, int n, int m) { return psr_fits<float>(N, n, m); }

template <typename T>
int launch_posterior_shared_reg(const T* Lop, const T* Vw, const T* X, const T* UHB, const T* ell, const T* s2,
                                const T* Bm, const T* M0, const T* xq, const T* jitter2, T* Mk, T* Bk, T* W, int nq,
                                int N, int n, int m, void* stream, int kind) {
    if (nq <= 0) return BCBF_OK;
    if (!Lop || !Vw || !X || !UHB || !ell || !s2 || !Bm || !M0 || !xq || !Mk || !Bk) return BCBF_EINVAL;
    if (N < 1 || n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > 3) return BCBF_EINVAL;
    if (!psr_fits<T>(N, n, m)) return BCBF_EINVAL;
    const int Np = round_up(N, NB), NSp = psr_state_dim(n);
    const size_t lds = psr_lds_bytes<T>(Np, n, m);
    hipStream_t st = (hipStream_t)stream;
    switch (m) {
#ifdef BCBF_PSR_DEV
        case 2: launch_psr_c<T, 3>(NSp, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n, kind); break;
        default: break;
#else
        case 1: launch_psr_c<T, 2>(NSp, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n, kind); break;
        case 2: launch_psr_c<T, 3>(NSp, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n, kind); break;
        default: launch_psr_c<T, 4>(NSp, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n, kind); break;
#endif
    }
    return check_launch("posterior_shared_reg");
}
template int launch_posterior_shared_reg<float>(const float*, const float*, const float*, const float*, const float*, const float*,
                                                const float*, const float*, const float*, const float*, float*, float*, float*,
                                                int, int, int, int, void*, int);

#endif
}  // namespace bcbf

#if !(defined(BCBF_PSR_PART_T) && defined(BCBF_PSR_PART_C))
extern "C" int bcbf_posterior_shared_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                         const double* ell, const double* s2, const double* Bm, const double* M0,
                                         const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                         int nq, int N, int n, int m, void* stream) {
    return bcbf::launch_posterior_shared_reg<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, n, m, stream, 0);
}
// The same matrix-core query with the opt-in Matern-5/2 data kernel (bcbf_posterior_query_matern52; parity unpinned: the
// reference has no Matern kernel)
extern "C" int bcbf_posterior_shared_matern52_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                                  const double* ell, const double* s2, const double* Bm, const double* M0,
                                                  const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                                  int nq, int N, int n, int m, void* stream) {
    return bcbf::launch_posterior_shared_reg<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, n, m, stream, 1);
}
// ... and with the product kernel RBF x Matern-5/2 (kind 2)
extern "C" int bcbf_posterior_shared_rbfm52_f64(const double* Lop, const double* Vw, const double* X, const double* UHB,
                                                const double* ell, const double* s2, const double* Bm, const double* M0,
                                                const double* xq, const double* jitter2, double* Mk, double* Bk, double* W,
                                                int nq, int N, int n, int m, void* stream) {
    return bcbf::launch_posterior_shared_reg<double>(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, n, m, stream, 2);
}
#endif
