// Regime S in fp64: many queries against ONE GP on v_mfma_f64_16x16x4_f64 (the reference's unicycle module is fp64,
// unicycle_move_to_pose.py:50; custom_predict with b test points, control_affine_model.py:536, 1051-1091).
//
// Same blocked forward substitution as posterior_shared.hip (fp32), rebuilt around what the fp64 MFMA gives:
//   * one wave = 4 queries = 16 right-hand-side columns (4 per query, unused ones zero); one wave per SIMD;
//   * the fp64 accumulator holds row g + 4r (lane group g = lane >> 4, register r) of a 16-row tile in column
//     j = lane & 15 -- and the B operand of the next MFMA wants B[k = g][n = j]: with k-step s = (tile u, register r)
//     covering block rows 16u + 4r + g, an accumulator register IS a B operand, with no interleaving of rows at all;
//   * so W = L^-1 Phi never leaves the registers: wreg[K][s] (8 doubles per 32-row block, pinned to the accumulation
//     registers, of which a wave alone on its SIMD has 256) is written by the diagonal step of block K and read as the
//     B operand of every later tile (I, K); blocks 12-14 (6 tiles read them) sit in a per-lane LDS slab instead so that
//     the accumulators themselves fit.  The loops over blocks are compile-time loops: every wreg index is a constant;
//   * A operands: every 32x32 tile of the packed operator is fetched ONCE per workgroup (16-byte buffer loads, two per
//     thread) into a ring of three LDS tiles and read by all four waves with explicit ds_read_b64, issued a whole tile
//     ahead of the MFMAs that consume them (bank-conflict-free image: see gload);  diagonal step: A = the stored
//     inverse (full-tile copy), B = Phi - acc.  One barrier per tile.
//   History at N = 512, 4096 queries: streaming VALU kernel 0.27 ms -> per-wave 8-byte operand loads from L2 0.145 ->
//   tiles shared through LDS 0.137 -> W pinned to AGPRs (no spills) + next tile's operands prefetched 0.095 ms
//   (a bare loop of this MFMA sustains 31.4 ns per instruction with one wave per SIMD: 66 us for the 2112 of a wave).
// The explicit prefetch (asm issue, asm wait) is only sound while the register allocator does not spill an operand
// between the two: build.py compiles this file with -Rpass-analysis=kernel-resource-usage and refuses a build whose
// kernels report a non-zero scratch size.  State dimensions n <= 4 (wider ones stream); N <= 32 * PS64_MAXBLK; beyond
// that, and for few queries, the streaming kernel (posterior_step.hip, two queries per workgroup) answers.
#include "bcbf_common.h"
#include "diag_tile64.h"      // LdsDouble

namespace bcbf {

using f64x4s = __attribute__((__vector_size__(4 * sizeof(double)))) double;
using u32x4q = __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned;

constexpr int PS64_MAXBLK = 16;          // N <= 512
#ifndef BCBF_PS64_REGBLK
#define BCBF_PS64_REGBLK 12
#endif
constexpr int PS64_REGBLK = BCBF_PS64_REGBLK;   // W blocks 0 .. REGBLK-1 live in registers (16 each: the accumulation registers
                                         // hold 256), blocks REGBLK .. 14 (read by 6 tiles at most) in a per-lane LDS slab;
                                         // block 15 is never an operand
constexpr int PS64_SLABBLK = PS64_MAXBLK - 1 - PS64_REGBLK;

// compile-time loop: the body sees a constant index (every wreg[][] subscript must be one, or the array leaves the
// register file for scratch memory -- "#pragma unroll" alone is a request the optimizer declines for 2000-MFMA bodies)
template <int I> struct Ic { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(Ic<B>{}); static_for<B + 1, E>(f); }
}

// block row of tile t in the order (0,0), (1,0), (1,1), (2,0), ...
constexpr int tile_row(int t) { int I = 0; while ((I + 1) * (I + 2) / 2 <= t) ++I; return I; }

// A operands of one tile: 16 explicit ds_read_b64 (a[u'][s] at byte offset OFF + 1024 s from the lane's two bases), and the
// wait that makes their results usable
template <int OFF> __device__ __forceinline__ void lds_get16(double (&a)[2][8], unsigned a0, unsigned a1) {
#define BCBF_RD(u_, s_, addr) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[u_][s_]) : "v"(addr), "n"(OFF + 1024 * s_))
    BCBF_RD(0, 0, a0); BCBF_RD(1, 0, a1); BCBF_RD(0, 1, a0); BCBF_RD(1, 1, a1);
    BCBF_RD(0, 2, a0); BCBF_RD(1, 2, a1); BCBF_RD(0, 3, a0); BCBF_RD(1, 3, a1);
    BCBF_RD(0, 4, a0); BCBF_RD(1, 4, a1); BCBF_RD(0, 5, a0); BCBF_RD(1, 5, a1);
    BCBF_RD(0, 6, a0); BCBF_RD(1, 6, a1); BCBF_RD(0, 7, a0); BCBF_RD(1, 7, a1);
#undef BCBF_RD
}
__device__ __forceinline__ void lds_wait16(double (&a)[2][8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[0][6]), "+v"(a[0][7]),
                   "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[1][4]), "+v"(a[1][5]), "+v"(a[1][6]), "+v"(a[1][7]));
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
    const int wave = threadIdx.x >> 6;                 // blockDim.x == 256: all four waves stage tiles
    double* Xs = smem64;                               // [Np][NS]  (state dim padded to NS with zeros)
    double* Us = Xs + (size_t)Np * NS;                 // [Np][C]   (rows >= N are zero: padded rows contribute nothing)
    double* Vs = Us + (size_t)Np * C;                  // [Np][NS]
    double* Ts = Vs + (size_t)Np * NS;                 // three 32x32 operator tiles (ring), 16-byte aligned
    double* Wl = Ts + 3072 + (size_t)wave * (PS64_SLABBLK * 8 * 64) + (threadIdx.x & 63);   // this lane's slots of W blocks >= REGBLK
    for (int i = threadIdx.x; i < Np * NS; i += blockDim.x) {
        const int row = i / NS, d = i - row * NS;
        const bool ok = row < N && d < n;
        Xs[i] = ok ? X[(size_t)row * n + d] : 0.0;
        Vs[i] = ok ? Vw[(size_t)row * n + d] : 0.0;
    }
    for (int i = threadIdx.x; i < Np * C; i += blockDim.x) Us[i] = i < N * C ? UHB[i] : 0.0;

    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int ql = j >> 2, c = j & 3;                  // query slot in the wave, component
    const int q = (blockIdx.x * 4 + wave) * QW + ql;
    const bool qok = q < nq, cok = c < C;              // (a wave past the end still stages tiles and meets the barriers)
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
    // Tile (I, K) of the operator, L[32I + r][32K + col] (K < I: the packed off-diagonal part, element (row, col) at
    // lop_base(col) + row; K == I: the full-tile copy of inv(L_II), column stride 32), is fetched ONCE per workgroup:
    // 512 16-byte pieces (column = piece / 16, rows 2 (piece % 16), +1), two per thread, 16 lanes per 256-byte column.
    // LDS image: column-major, 256 bytes per column, the two 128-byte halves of ODD columns swapped -- an A-operand read
    // (ds_read_b64: lanes 0-31 = lane groups g, g+1 = two adjacent columns, 16 rows each) then covers all 64 banks once.
    const int pc0 = threadIdx.x >> 4, pp = threadIdx.x & 15;      // piece h: column pc0 + 16 h, piece pp
    auto gload = [&](u32x4q (&r)[2], int I, int K) {
        const bool isdiag = K == I;
        const int stride = isdiag ? NB : Np - NB * (K + 1);                   // column stride inside block column K
        const int base = isdiag ? lop_dfull_block(I, Np) : lop_base<V>(K * NB, Np) + I * NB;   // element (row 0 of the tile, col 0)
        const int voff = (pc0 * stride + 2 * pp) * (int)sizeof(double);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            r[h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (base + 16 * h * stride) * (int)sizeof(double), 0);
    };
    const int wr_off = pc0 * NB + ((pp ^ ((pc0 & 1) << 3)) << 1);             // (16 columns further: + 512 doubles)
    auto lds_put = [&](double* tb, const u32x4q (&r)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4q*>(tb + wr_off + 512 * h) = r[h];
    };
    // a[u'][s] = tile[row 16u' + j][col 4s + g]
    const int rd_off0 = g * NB + ((((j >> 1)) ^ ((g & 1) << 3)) << 1) + (j & 1);
    const int rd_off1 = g * NB + (((8 | (j >> 1)) ^ ((g & 1) << 3)) << 1) + (j & 1);

    double wreg[PS64_REGBLK][8];                       // W_K: rows 16u + 4r + g of block K at index 4u + r, column j
    auto wget = [&](auto Kc, int s_) -> double {
        constexpr int K_ = decltype(Kc)::value;
        if constexpr (K_ < PS64_REGBLK) return wreg[K_][s_];
        else return Wl[((K_ - PS64_REGBLK) * 8 + s_) * 64];
    };
    // Pipeline, step t = tile t (tiles in the order (0,0), (1,0), (1,1), (2,0), ...; three LDS buffers):
    //   write tile t+2 (fetched during step t-1) into buffer (t+2) % 3  |  fetch tile t+3 into registers  |
    //   read the A operands of tile t+1 from buffer (t+1) % 3  |  multiply tile t (operands read a step ago)  |  barrier
    auto fetch = [&](u32x4q (&r)[2], auto tc) {
        constexpr int t_ = decltype(tc)::value;
        constexpr int I_ = tile_row(t_);
        gload(r, I_, t_ - I_ * (I_ + 1) / 2);
    };
    // The A operands of the NEXT tile are read with explicit ds_read_b64 (issued here, at the head of the step, and waited
    // for by lds_wait16() at its tail): written as plain loads the scheduler sinks each one to its use, a single MFMA ahead,
    // and every MFMA pair then waits out an LDS round trip (measured: 92 cycles per MFMA instead of 64).
    const unsigned ts_a0 = (unsigned)(size_t)(LdsDouble*)(Ts + rd_off0), ts_a1 = (unsigned)(size_t)(LdsDouble*)(Ts + rd_off1);
    u32x4q stage[2];
    double acur[2][8], anxt[2][8], pend[8];
    fetch(stage, Ic<0>{});
    lds_put(Ts, stage);
    fetch(stage, Ic<1>{});
    lds_put(Ts + 1024, stage);
    fetch(stage, Ic<2>{});
    __syncthreads();                                   // staging of X / UH B / Vw and tiles 0, 1 visible
    lds_get16<0>(acur, ts_a0, ts_a1);
    lds_wait16(acur);
    // Gram row / mean column of this lane's query from the W tile of block Ib held in pend[]
    auto epilogue = [&](int Ib) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double v = pend[e];
            const int row = Ib * NB + 16 * (e >> 2) + 4 * (e & 3) + g;
            const double vb[4] = {dpp_qd<0x00>(v), dpp_qd<0x55>(v), dpp_qd<0xAA>(v), dpp_qd<0xFF>(v)};
#pragma unroll
            for (int a_ = 0; a_ < C; ++a_) gram[a_] += v * vb[a_];             // lane c: G[c][a]
#pragma unroll
            for (int d = 0; d < NS; ++d) mk[d] += Vs[row * NS + d] * v;        // lane c: (Vw'W)[d][c]
        }
    };
    static_for<0, PS64_MAXBLK>([&](auto Ict) {
        constexpr int I = decltype(Ict)::value;
        if (I < nblk) {
            f64x4s acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            // ---- off-diagonal tiles (I, K), K < I: acc += L_IK W_K
            static_for<0, I>([&](auto Kct) {
                constexpr int K = decltype(Kct)::value;
                constexpr int t = I * (I + 1) / 2 + K;
                lds_put(Ts + 1024 * ((t + 2) % 3), stage);
                fetch(stage, Ic<t + 3>{});
                lds_get16<((t + 1) % 3) * 8192>(anxt, ts_a0, ts_a1);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const double wk = wget(Kct, s);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[0][s], wk, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[1][s], wk, acc1, 0, 0, 0);
                }
                if constexpr (K == 0 && I > 0) epilogue(I - 1);                // VALU work beside this row's first MFMAs
                lds_wait16(anxt);
#pragma unroll
                for (int s = 0; s < 8; ++s) { acur[0][s] = anxt[0][s]; acur[1][s] = anxt[1][s]; }
                __syncthreads();
            });
            constexpr int t = I * (I + 1) / 2 + I;
            lds_put(Ts + 1024 * ((t + 2) % 3), stage);
            fetch(stage, Ic<t + 3>{});
            lds_get16<((t + 1) % 3) * 8192>(anxt, ts_a0, ts_a1);
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
                pend[e] = v;
                if constexpr (I < PS64_REGBLK) { wreg[I][e] = v; asm volatile("" : "+a"(wreg[I][e])); }   // lives in the accumulation registers
                else if constexpr (I < PS64_MAXBLK - 1) Wl[((I - PS64_REGBLK) * 8 + e) * 64] = v;
            }
            if (Wout != nullptr && qok && cok) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    Wout[((size_t)q * Np + I * NB + 16 * (e >> 2) + 4 * (e & 3) + g) * C + c] = pend[e];
            }
            lds_wait16(anxt);
#pragma unroll
            for (int s = 0; s < 8; ++s) { acur[0][s] = anxt[0][s]; acur[1][s] = anxt[1][s]; }
            if (I == nblk - 1) epilogue(I);
            __syncthreads();
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

static int padded_state_dim64(int n) { return n <= 2 ? 2 : n; }   // instantiated widths: 2, 3, 4

// N <= 512, n <= 4 (wider states need more registers than the explicit operand prefetch leaves: they stream) and the
// staged copies fit in LDS
bool posterior_shared64_fits(int N, int n, int m) {
    const size_t Np = round_up(N, NB);
    return n <= 4 && Np <= (size_t)NB * PS64_MAXBLK && (Np * (2 * padded_state_dim64(n) + m + 1) + 3072 + 4 * PS64_SLABBLK * 512) * sizeof(double) <= 160 * 1024;
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
#ifndef BCBF_PS64_DEV          // (development: compile two instantiations only)
        case 2: BCBF_PSH64(2); break;
        case 4: BCBF_PSH64(4); break;
#endif
        case 3: BCBF_PSH64(3); break;
        default: break;
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
    const size_t lds = ((size_t)Np * (2 * NSp + C) + 3072 + (Np > NB * PS64_REGBLK ? 4 * PS64_SLABBLK * 512 : 0)) * sizeof(double);
    const int waves = (nq + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((waves + 3) / 4), block(256);            // one wave per SIMD (the W tiles of a wave take 256 VGPRs)
    switch (m) {
#ifdef BCBF_PS64_DEV
        case 2: launch_shared64_c<3>(NSp, grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n); break;
        default: break;
#else
        case 1: launch_shared64_c<2>(NSp, grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n); break;
        case 2: launch_shared64_c<3>(NSp, grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n); break;
        default: launch_shared64_c<4>(NSp, grid, block, lds, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, Mk, Bk, W, nq, N, Np, n); break;
#endif
    }
    return check_launch("posterior_shared64");
}
