// K1 + K2, fp64, ONE WAVE PER INSTANCE (batches): the blocked left-looking Cholesky of refit_mfma64.hip re-parallelised.
//
// Why: in the workgroup-per-instance kernel the 32x32 diagonal tile of every block column is factored and inverted by
// ONE wave while the other three wait (at N = 256 that serial chain was 60 % of an instance's lifetime; MFMA pipe busy
// 15 %).  The chain cannot be shortened by more waves -- but a batch has thousands of independent chains.  Here an
// instance is owned by a single wave (no workgroup barriers, per-wave LDS), so every SIMD of the chip advances its own
// chain, and the chain itself is rebuilt around 4-column blocks:
//
//   diagonal tile S (32x32, symmetric): round 2 held it as lane = (row, column half) with the four raw columns of a step
//   published to LDS, the 4x4 diagonal block factored AND inverted redundantly in every lane (4 rsqrt chains: the only
//   serial part), a rank-4 update of 16 elements per lane and a separate 8-step inverse pass (diag_tile64.h:
//   diag_factor_invert, still used by the few-instances fp32 form): ~13 k cycles per tile instead of ~65 k.  Round 3
//   (diag_factor_invert_acc): the tile never leaves the MFMA accumulator layout the update stream produces, the rank-4
//   updates are four 16x16x4 MFMAs whose operands ARE the panel entries a lane has just formed, and the inverse rides
//   along as a second accumulator set: 14.5 -> 8.4 us per tile.
//
// The rank-32 updates (v_mfma_f64_16x16x4_f64 on transposed tiles, operands straight from the packed operator: the
// two 16-row halves of a tile are interleaved, MFMA index mu of half h = tile row 2 mu + h, so that a lane's two operand
// values are adjacent rows and arrive in one 16-byte load) and the panel solve (accumulator registers as B operands)
// are those of refit_mfma64.hip, run tile after tile by the owning wave (fp32: two tiles of a block column per pass,
// sharing the A operand).  Same outputs, same packed layout (bcbf_common.h), same info convention.  The second kernel of
// this file gives an instance TWO waves (small systems: a serial chain wave and a bulk wave with look-ahead).
//
// A wave alone on its SIMD hides nothing: every wait is paid in full.  What that meant here (cycle counters,
// -DBCBF_RW64_PROF, N = 256): the K_b value pass was 44 % of the kernel -- not its arithmetic, but `if (d < n)` around
// each row load and each LDS read: every access became a basic block with a complete wait behind it (ten serialized
// memory round trips and 120 LDS round trips per tile).  Zero-filled fixed-width rows + out-of-range buffer offsets
// (straight-line code, all accesses in flight together) took that pass from 18 k to 4 k cycles per tile:
// 0.63 -> 0.45 ms at 1024 x 256, 10.7 -> 8.2 ms at 4096 x 512 (same box).  (The same rewrite LOSES in the
// workgroup-form kernels, refit_mfma*.hip: they run four waves per SIMD against a 128 / 256-register cap, waits are
// hidden by the other waves and the extra values in flight spill.)
#include <type_traits>
#include "bcbf_common.h"
#ifdef BCBF_RP_TRACE
// time stamps inside the diagonal tile's factor + inverse (first call of workgroup 0 only), after the hand-off stamps
namespace bcbf { __device__ long long rp_trace_buf[2][4096]; __device__ int dt_trace_n; }
#define BCBF_DT_STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { const int i_ = bcbf::dt_trace_n; \
    if (i_ < 40) { bcbf::rp_trace_buf[0][4000 + i_] = ((long long)(k) << 56) | (long long)wall_clock64(); bcbf::dt_trace_n = i_ + 1; } } } while (0)
#endif
#include "diag_tile64.h"

namespace bcbf {

using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
using f32x4w = __attribute__((__vector_size__(4 * sizeof(float)))) float;

#ifndef BCBF_RP_CHAIN_PRIO
#define BCBF_RP_CHAIN_PRIO 0     // two-wave / team forms: 1 = the chain wave (diagonal tiles) issues with priority over co-resident bulk waves
                                 // (s_setprio 3).  Measured, round 6, fp64 N = 256: 1024 instances 0.295 -> 0.295 / 0.320 ms (two runs), 512: 0.205 -> 0.195,
                                 // 768: 0.254 -> 0.281 -- no gain: the chain waits on LDS round trips and MFMA latencies, not on issue slots
#endif
#ifndef BCBF_RW64_WPB
#define BCBF_RW64_WPB 1          // waves (= instances) per workgroup (measured: 1 beats 2 by 5 %, 3 loses 80 %: LDS)
#endif
#ifndef BCBF_RW64_OCC
#define BCBF_RW64_OCC 1          // waves per SIMD the register allocation aims at (2 = 256 VGPRs spills ~100 registers
                                 // and is 1.6x slower at N = 256, batch 1024: 0.94 vs 0.57 ms)
#endif
// fp32: both allocations are compiled; two waves per SIMD (256 registers each, ~40 spilled) win once the batch gives every
// SIMD more than one instance (4096 x 512: 5.4 -> 4.8 ms, 4096 x 256: 1.33 -> 1.12), one wave per SIMD below that
// (1024 x 256: 0.35 vs 0.44 ms)
#ifndef BCBF_RW32_KS
#define BCBF_RW32_KS 8
#endif
#ifndef BCBF_RW32_PAIRS
#define BCBF_RW32_PAIRS 1        // two tiles of a block column per pass of the update stream (shared A operand)
#endif
#ifndef BCBF_RW32_SUPER
#define BCBF_RW32_SUPER 1        // fp32, N >= 512: TWO block columns per pass (64-column super-panels, 2 x 2 tiles per stream pass): the
                                 // panels left of a super-panel are streamed once per 64 columns instead of once per 32.  Measured
                                 // (ms off / on): 4096 x 1024 27.4 / 21.8, 1024 x 512 1.03 / 0.98, 4096 x 512 3.79 / 3.67 (FETCH_SIZE 17.3 ->
                                 // 14.2 GB at two waves per SIMD, 11.0 GB at one); N = 256: 4096 x 256 0.70 / 0.82 -- too few columns
                                 // left of a super-panel to pay for the wider working set: off below N = 512
#endif
#ifndef BCBF_RW32_SUPER_KS
#define BCBF_RW32_SUPER_KS 4     // k-steps per pipeline stage of the four-stream pass at two waves per SIMD (2: +3 %, 8: +12 % -- registers); at one
                                 // wave per SIMD the stage is 8 deep (1024 x 512 0.90 -> 0.88 ms, 4096 x 1024 20.0 -> 19.6)
#endif
#ifndef BCBF_RW32_SUPER_DIAG3
#define BCBF_RW32_SUPER_DIAG3 1  // 1: the three tiles of the diagonal 2 x 2 block in one stream pass; 0: the diagonal tile by itself first (measured:
                                 // more scratch, not less -- the extra stream instantiation costs more than the two tiles it parks)
#endif
#ifndef BCBF_RW_SUPER_AINV_LDS
#define BCBF_RW_SUPER_AINV_LDS 1 // super-panels: the two inverted diagonal tiles stay in LDS and the panel solves read their operands from
                                 // there right before use (16 ds_reads per solve pair) instead of holding 2 x 16 (fp64: 2 x 32) registers
#endif
#ifndef BCBF_RW_VALUES_FAST
#define BCBF_RW_VALUES_FAST 1    // K_b values of tiles off the diagonal and clear of the padding without the per-value compares / selects
#endif
#ifndef BCBF_RW_FAST44
#define BCBF_RW_FAST44 0
#endif
#ifndef BCBF_RW_VALUES_FAST64
#define BCBF_RW_VALUES_FAST64 1
#endif
#ifndef BCBF_RW_VALUES_FAST_DIAG
#define BCBF_RW_VALUES_FAST_DIAG 1
#endif
#ifndef BCBF_RW_SUPER_INREG
#define BCBF_RW_SUPER_INREG 1    // super-panels: the 32-deep update of column J + 1 by column J takes the just-solved tiles L_IJ' straight from the
                                 // registers they were solved into (accumulator layout = B operand with a permuted contraction order; the A
                                 // operand L_{J+1,J} is loaded once per super-panel in that order) -- no store -> fence -> load round trip per
                                 // row pair, no B streams
#endif
#ifndef BCBF_RW_SUPER_INREG_OCC2
#define BCBF_RW_SUPER_INREG_OCC2 1    // ... at two waves per SIMD too, with the A operand re-read per row pair (after the leaner value pass: 4096 x 512
                                      // 3.14 -> 3.09 ms; before it the extra live registers cost 3 %)
#endif
#ifndef BCBF_RW_SUPER_A2_RELOAD
#define BCBF_RW_SUPER_A2_RELOAD 0   // 1: that A operand is re-read (L2 hits) at every row pair instead of living in 16 registers across the stream loops
                                    // (parking it in LDS instead costs the 4 KB that take the kernel's LDS past 160 KB / 8 waves per CU)
#endif
#ifndef BCBF_RW64_SUPER
#define BCBF_RW64_SUPER 1        // fp64 batches from N = 1024: the same super-panels (the plain fp64 path reads TWO panels per tile)
#endif
#ifndef BCBF_RW64_SUPER_KS
#define BCBF_RW64_SUPER_KS 4     // (1 / 2 / 4 measured at 1024 x 1024: 14.3 / 11.7 / 11.1 ms)
#endif
#ifndef BCBF_RW32_SUPER_AHEAD
#define BCBF_RW32_SUPER_AHEAD 0  // 1: row inputs of the next tile pair loaded before the stream pass -- 36 registers live across it; at two waves
                                 // per SIMD that is 80 B more scratch and 4096 x 512 runs 3.95 instead of 3.67 ms
#endif
#ifndef BCBF_RW64_PAIRS
#define BCBF_RW64_PAIRS 0        // fp64: measured slower below N = 1024 (4096 x 512: 8.15 against 7.4 ms; 1024 x 1024: 14.1 against 14.7)
#endif
#ifndef BCBF_RW64_KS
#define BCBF_RW64_KS 8           // k-steps (of 4 columns) per software-pipeline stage of the update stream (8 beats 4 by 4 %)
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

template <typename T, bool SUP = false> struct RWShared {
    DiagTile<T> d;                            // the diagonal tile's working set (diag_tile64.h)
    T colX[NB][BCBF_MAX_STATE_DIM];
    T colUH[NB][BCBF_MAX_CTRL_DIM + 1];
    unsigned pack_rc[LOP_DB / 2];             // (row, column) of the entries of a packed inverted diagonal block, two per word
    T colX2[SUP ? NB : 1][BCBF_MAX_STATE_DIM];          // second block column of a super-panel (fp32 one-wave form)
    T colUH2[SUP ? NB : 1][BCBF_MAX_CTRL_DIM + 1];
    T xinvJ[SUP ? NB : 1][DT_LS];                       // inv(L_JJ) of the super-panel's FIRST column (d.xinv holds the second's)
};
// packed lower triangle, column-major: entry k <-> (r, c), r >= c; (r | c << 8) per entry, 0xffff = the block's padding
__device__ inline void rw_pack_table(__attribute__((address_space(3))) unsigned* tab, int tid, int nthreads) {
    for (int w = tid; w < LOP_DB / 2; w += nthreads) {
        unsigned word = 0;
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * w + h;
            // column c starts at entry c (65 - c) / 2: the root of that quadratic, then one step either way
            int c = (int)((65.0f - __builtin_sqrtf(4225.0f - 8.0f * (float)(k < 528 ? k : 527))) * 0.5f);
            c = c < 0 ? 0 : c > NB - 1 ? NB - 1 : c;
            if (c < NB - 1 && lop_dinv_col(c + 1) + c + 1 <= k) ++c;
            if (lop_dinv_col(c) + c > k) --c;
            const int r = k - lop_dinv_col(c);
            const unsigned rc = k < 528 ? (unsigned)(r | (c << 8)) : 0xffffu;
            word |= rc << (16 * h);
        }
        tab[w] = word;
    }
}

// what differs between the two precisions: the MFMA, the row its accumulator register r holds in lane group g (the
// M index: g + 4r in fp64, 4g + r in fp32 -- the same value is the contraction index of k-step r when an accumulator is
// fed back as a B operand), the exponential and the load width
template <typename T> struct RW;
template <> struct RW<double> {
    using acc_t = f64x4;
    using vec2 = double2;
    __device__ static int midx(int r, int g) { return 4 * r + g; }
    static constexpr int PANEL_R0 = 2;                // k-steps that meet the upper 16 rows of inv(L_JJ): midx < 8 <=> r < 2
    __device__ static acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static double exp_neg(double x) { return exp_neg64(x); }
    __device__ static double bload(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0)); }
    __device__ static double2 bload2(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
};
template <> struct RW<float> {
    using acc_t = f32x4w;
    using vec2 = float2;
    __device__ static int midx(int r, int g) { return 4 * g + r; }
    static constexpr int PANEL_R0 = 4;                // midx < 8 <=> g < 2: a property of the lane, every k-step runs
    __device__ static acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static float exp_neg(float x) { return __expf(-x); }
    __device__ static float bload(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
    __device__ static float2 bload2(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
};

// SUP (fp32): the super-panel form (two block columns per pass, N >= 512) -- its own instantiation, so that the plain form
// keeps the registers and the LDS it had (compiled into one kernel, 4096 x 256 went from 0.70 to 0.79 ms for code it never runs)
template <typename T, bool FROM_DENSE, int OCC, bool SUP = false>
__global__ void __launch_bounds__(64 * RW_WPB, OCC)
refit_wave_kernel(const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ Bm,
                    const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ jitter,
                    const T* __restrict__ Kdense, T* __restrict__ Lop, T* __restrict__ UHBout,
                    T* __restrict__ Ldense, int* __restrict__ info, int Bt, int N, int Np, int n, int C, const int* only_bad) {
    constexpr int V = Vec<T>::V, ES = (int)sizeof(T);
    using P = RW<T>;
    using acc_t = typename P::acc_t;
    using T2 = typename P::vec2;
    static_assert(!SUP || sizeof(T) == 8 || BCBF_RW32_PAIRS, "the super-panel form lives in the pair path");
    __shared__ RWShared<T, SUP> shm[RW_WPB];
    // wave-uniform instance index: the per-instance pointers and hyper-parameters then live in SGPRs (scalar loads)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x * RW_WPB + wave;
    if (b >= Bt) return;                                      // whole wave: no workgroup barrier anywhere below
    if (only_bad != nullptr && only_bad[b] == 0) { if (lane == 0) info[b] = 0; return; }     // bcbf_refit_retry: factored already
    __attribute__((address_space(3))) RWShared<T, SUP>& sh = *(__attribute__((address_space(3))) RWShared<T, SUP>*)&shm[wave];   // keep ds_* ops
    const int j16 = lane & 15, g = lane >> 4;                 // MFMA roles

    T* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    const T* Xb = FROM_DENSE ? nullptr : X + (size_t)b * N * n;
    const T* UHb = FROM_DENSE ? nullptr : UH + (size_t)b * N * C;
    const T* Kb = FROM_DENSE ? Kdense + (size_t)b * N * N : nullptr;
    T* Ld = Ldense ? Ldense + (size_t)b * N * N : nullptr;
    // (registers hold the first RXD state components of a row -- no reference system has more than 4; wider states read
    //  the rest from memory in the value pass)
    constexpr int RXD = 4;
    T iell[RXD], Bmr[(BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1)];
    T s2 = T(0.0);
    if (!FROM_DENSE) {
        s2 = s2p[b];
#pragma unroll
        for (int d = 0; d < RXD; ++d) iell[d] = d < n ? T(1.0) / ell[(size_t)b * n + d] : T(0.0);
#pragma unroll
        for (int a = 0; a < (BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1); ++a)
            Bmr[a] = a < C * C ? Bm[(size_t)b * C * C + a] : T(0.0);
        for (int i = lane; i < N; i += 64)
            for (int c = 0; c < C; ++c) {
                T s = T(0.0);
                for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                UHBout[((size_t)b * N + i) * C + c] = s;
            }
    }
    if (Ld)
        for (int e = lane; e < N * N; e += 64) { const int i = e / N, j = e - i * N; if (j > i) Ld[e] = T(0.0); }
    const T* UHBb = FROM_DENSE ? nullptr : UHBout + (size_t)b * N * C;      // rows of UH B, read back per tile
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();

    int fail = 0;
    const int nblk = Np / NB;
    // row inputs through buffer descriptors (zero-filled out-of-range reads; jitter may be absent: an empty buffer)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Xb), 0, FROM_DENSE ? 0 : N * n * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(UHBb), 0, FROM_DENSE ? 0 : N * C * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsJ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>((!FROM_DENSE && jitter) ? jitter + (size_t)b * N : X), 0, (!FROM_DENSE && jitter) ? N * ES : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(lop, 0, (unsigned)min((size_t)0xfffffff0u, lop_elems<V>(Np) * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsUH = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(UHb), 0, FROM_DENSE ? 0 : N * C * ES, 0x00020000);
    rw_pack_table(sh.pack_rc, lane, 64);
    __builtin_amdgcn_wave_barrier();
#ifdef BCBF_RW64_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    RW_T0();
    for (int J = 0; J < nblk && fail == 0; ++J) {
        const int col0 = J * NB;
        if (!FROM_DENSE) {
            // (zero-filled to the full widths: the value pass below reads fixed-width rows without a branch; the loads
            //  themselves are branch-free too -- an absent component is an out-of-range buffer offset and reads as zero --
            //  and all in flight together)
            constexpr int SX = NB * BCBF_MAX_STATE_DIM / 64, SU = NB * (BCBF_MAX_CTRL_DIM + 1) / 64;
            T sx[SX], su[SU];
#pragma unroll
            for (int t = 0; t < SX; ++t) {
                const int e = lane + 64 * t, c = e / BCBF_MAX_STATE_DIM, d = e % BCBF_MAX_STATE_DIM;
                sx[t] = P::bload(rsX, (col0 + c < N && d < n) ? ((col0 + c) * n + d) * ES : -ES);
            }
#pragma unroll
            for (int t = 0; t < SU; ++t) {
                const int e = lane + 64 * t, c = e / (BCBF_MAX_CTRL_DIM + 1), a = e % (BCBF_MAX_CTRL_DIM + 1);
                su[t] = P::bload(rsUH, (col0 + c < N && a < C) ? ((col0 + c) * C + a) * ES : -ES);
            }
#pragma unroll
            for (int t = 0; t < SX; ++t) { const int e = lane + 64 * t; sh.colX[e / BCBF_MAX_STATE_DIM][e % BCBF_MAX_STATE_DIM] = sx[t]; }
#pragma unroll
            for (int t = 0; t < SU; ++t) { const int e = lane + 64 * t; sh.colUH[e / (BCBF_MAX_CTRL_DIM + 1)][e % (BCBF_MAX_CTRL_DIM + 1)] = su[t]; }
            if (SUP && J + 1 < nblk) {      // super-panel: the inputs of block column J + 1 too
#pragma unroll
                for (int t = 0; t < SX; ++t) {
                    const int e = lane + 64 * t, c = e / BCBF_MAX_STATE_DIM, d = e % BCBF_MAX_STATE_DIM;
                    sx[t] = P::bload(rsX, (col0 + NB + c < N && d < n) ? ((col0 + NB + c) * n + d) * ES : -ES);
                }
#pragma unroll
                for (int t = 0; t < SU; ++t) {
                    const int e = lane + 64 * t, c = e / (BCBF_MAX_CTRL_DIM + 1), a = e % (BCBF_MAX_CTRL_DIM + 1);
                    su[t] = P::bload(rsUH, (col0 + NB + c < N && a < C) ? ((col0 + NB + c) * C + a) * ES : -ES);
                }
#pragma unroll
                for (int t = 0; t < SX; ++t) { const int e = lane + 64 * t; sh.colX2[e / BCBF_MAX_STATE_DIM][e % BCBF_MAX_STATE_DIM] = sx[t]; }
#pragma unroll
                for (int t = 0; t < SU; ++t) { const int e = lane + 64 * t; sh.colUH2[e / (BCBF_MAX_CTRL_DIM + 1)][e % (BCBF_MAX_CTRL_DIM + 1)] = su[t]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        RW_ACC(0);                                             // column staging
        T ainv[2][2][4];                                  // inv(L_JJ) as panel-solve A operands, loaded after the diagonal tile

        // fp32: two tiles of a block column per pass of the update stream (they share the A operand: three panel reads where
        // two single tiles make four -- 4096 x 512 fp32 fetches 21 GB past L2 per launch for a 2 GB result, PMC).  fp64: one
        // tile per pass -- the pair path needs more than the 512 registers there and measured slower below N = 1024 (4096 x
        // 512: 8.15 against 7.4 ms).  Two bodies: the single-tile stream written over the pair path's helpers came out
        // with twice the register moves in its loop (177 instructions per 32 MFMAs against 111) and lost 5 %.
        constexpr bool PAIRS = SUP || (sizeof(T) == 4 ? BCBF_RW32_PAIRS : BCBF_RW64_PAIRS);
        if constexpr (PAIRS) {
            // inputs of a tile's two rows per lane (x_i, (UH B)_i, jitter_i): loaded one step ahead, so that the loads are in
            // flight during the previous step's update stream instead of queueing behind its panel stores.
            // Branch-free: every load is issued, a component the model does not have (d >= n, c >= C) or a row past the end
            // is an out-of-range buffer offset and reads as zero.  (Written with `if (d < n)` around plain loads each one
            // became its own basic block with a full wait behind it: ten serialized memory round trips per tile, 44 % of
            // the kernel at N = 256.)
            struct Rows { T rx[2][RXD], ru[2][BCBF_MAX_CTRL_DIM + 1], rj[2]; };
            Rows rw0, rw1;
            auto load_rows = [&](Rows& q, int I_) {
                if (FROM_DENSE) return;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = I_ * NB + 2 * j16 + ib;
                    const bool in = I_ < nblk && i < N;
#pragma unroll
                    for (int d = 0; d < RXD; ++d)
                        q.rx[ib][d] = P::bload(rsX, (in && d < n) ? (i * n + d) * ES : -ES);
#pragma unroll
                    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c)
                        q.ru[ib][c] = P::bload(rsU, (in && c < C) ? (i * C + c) * ES : -ES);
                    q.rj[ib] = P::bload(rsJ, in ? i * ES : -ES);
                }
            };
            // ---- initial value of a tile: acc[cb][ib][r] = -K_b'(c = 2 (4r + g) + cb, i = 32 I + 2 j16 + ib)  (the accumulators
            //      carry -S', see the update stream; halves interleaved: the two halves of an operand are adjacent rows, one
            //      16-byte load).  Straight-line: the column's inputs come from LDS as fixed-width rows (4 state components,
            //      4 control components; zeros beyond n / C, where iell and the row inputs are zero too), ONE read per column for
            //      both rows of the lane.  (With `if (d < n)` around each LDS read every read sat in a basic block of its own
            //      with a full wait behind it: 120 serialized LDS round trips per tile.)
            auto values_s = [&](acc_t (&acc)[2][2], const Rows& q, int I, auto sel) {
                constexpr int SEL = decltype(sel)::value;          // 0: block column J (colX / colUH), 1: block column J + 1
                const int col0 = (J + SEL) * NB;                   // (shadows the loop's col0)
                const int irow = I * NB + 2 * j16;
                if (FROM_DENSE) {
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        const int i = irow + ib;
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 2 * P::midx(r, g) + cb, j = col0 + c;
                                T val;
                                if (i >= N || j >= N) val = (i == j) ? T(1.0) : T(0.0);      // padding: identity
                                else val = (j <= i) ? Kb[(size_t)i * N + j] : Kb[(size_t)j * N + i];
                                acc[cb][ib][r] = -val;
                            }
                    }
                } else if (BCBF_RW_VALUES_FAST && (sizeof(T) == 4 || BCBF_RW_VALUES_FAST64) && (BCBF_RW_VALUES_FAST_DIAG || I > J + 1) && I != J + SEL && (I + 1) * NB <= N && col0 + NB <= N && n <= 4) {
                    // a tile off the diagonal and clear of the padding (120 of the 136 tiles at N = 512): no diagonal jitter, no
                    // identity padding -- 16 instructions per value instead of 30 (the common shapes n <= 3, C <= 3 skip the zero
                    // fourth component as well)
                    const T ms2 = -s2;
                    auto fast = [&](auto ndc, auto ncc) {
                        constexpr int ND = decltype(ndc)::value, NC = decltype(ncc)::value;
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 2 * P::midx(r, g) + cb;
                                T cx[4], cu[4];
#pragma unroll
                                for (int d = 0; d < 4; ++d) {
                                    cx[d] = SEL ? sh.colX2[c][d] : sh.colX[c][d];
                                    cu[d] = SEL ? sh.colUH2[c][d] : sh.colUH[c][d];
                                }
#pragma unroll
                                for (int ib = 0; ib < 2; ++ib) {
                                    T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                                    for (int d = 0; d < ND; ++d) { const T z = (q.rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
#pragma unroll
                                    for (int a_ = 0; a_ < NC; ++a_) uu += q.ru[ib][a_] * cu[a_];
                                    acc[cb][ib][r] = ms2 * P::exp_neg(T(T(0.5)) * d2) * uu;
                                }
                            }
                    };
                    // Every row input is pinned as "used" here (empty asm statements with the value as input: no instruction).
                    // Observed without the pins: the fp64 super-panel instantiation (256 + 256 registers, scratch) came out WRONG with
                    // this branch, deterministically -- the last diagonal tile of every system factored as if it had no jitter
                    // (three-component body), zeros from the third block row on (four-component body) -- and correct again as soon
                    // as the branch merely NAMED q.rj (a select that is never taken was enough).  The inputs this branch does not
                    // read (rj; rx[.][3], ru[.][3]) are live only into the other branch; whatever the optimizer does with their
                    // loads then (sinking them is what the symptom suggests), the values that arrive there were not the ones
                    // load_rows fetched.  fp32 passed every parity test without the pins; it gets them all the same.
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        asm volatile("" :: "v"(q.rj[ib]));
#pragma unroll
                        for (int d = 0; d < RXD; ++d) asm volatile("" :: "v"(q.rx[ib][d]));
#pragma unroll
                        for (int a_ = 0; a_ < BCBF_MAX_CTRL_DIM + 1; ++a_) asm volatile("" :: "v"(q.ru[ib][a_]));
                    }
                    if (!BCBF_RW_FAST44 && OCC == 1 && n <= 3 && C <= 3) fast(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});   // (two waves per SIMD: one body -- the second costs registers there, 4096 x 256 0.72 against 0.77 ms)
                    else fast(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
                } else {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 2 * P::midx(r, g) + cb, j = col0 + c;
                            T cx[4], cu[4];
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                cx[d] = SEL ? sh.colX2[c][d] : sh.colX[c][d];
                                cu[d] = SEL ? sh.colUH2[c][d] : sh.colUH[c][d];
                            }
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) {
                                const int i = irow + ib;
                                T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                                for (int d = 0; d < 4; ++d) { const T z = (q.rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
                                if (n > 4) {                                          // (wave-uniform; no reference system has n > 4)
                                    for (int d = 4; d < n; ++d) {
                                        const T xi = i < N ? Xb[(size_t)i * n + d] : T(0.0);
                                        const T z = (xi - (SEL ? sh.colX2[c][d] : sh.colX[c][d])) / ell[(size_t)b * n + d];
                                        d2 += z * z;
                                    }
                                }
#pragma unroll
                                for (int a_ = 0; a_ < 4; ++a_) uu += q.ru[ib][a_] * cu[a_];
                                T val = s2 * P::exp_neg(T(T(0.5)) * d2) * uu + (i == j ? q.rj[ib] : T(0.0));
                                val = (i >= N || j >= N) ? ((i == j) ? T(1.0) : T(0.0)) : val;   // padding: identity
                                acc[cb][ib][r] = -val;
                            }
                        }
                }
            };
            auto values = [&](acc_t (&acc)[2][2], const Rows& q, int I) { values_s(acc, q, I, std::integral_constant<int, 0>{}); };
            // ---- -S' += L_J L_I'  over all previous columns (software pipelined: next stage's operands in flight), for NT = 1 or
            //      2 tiles of the block column at once: the two tiles share the A operand (block row J), so a pair reads three
            //      panels where two single tiles read four -- the batches this form serves are bound by exactly that traffic
            //      (4096 x 512 fp32: 21 GB fetched past L2 per launch for a 2 GB result, PMC).  The accumulators hold -S': the MFMA
            //      has no negate modifier, and flipping an operand costs four VALU instructions per k-step against once per tile
            //      at the consumer.  Addresses: inside block column K the packed columns are a fixed stride apart, so a load is
            //      buffer base + SCALAR offset (column block, k-step) + per-lane offset (lane group's column, row): two
            //      multiply-adds per fetch instead of the column-offset polynomial per load (a wave issues in order: address
            //      arithmetic does not hide behind its own MFMAs)
            // (two separate bodies: written as one body over NT the single-tile stream came out with twice the register moves and
            //  six times the waits in its loop -- 177 instructions per 32 MFMAs against 111 -- and fp64 lost 5 %)
            auto update = [&](acc_t (&acc)[2][2], int I) {
                constexpr int KS = sizeof(T) == 4 ? BCBF_RW32_KS : BCBF_RW64_KS;
                const int irow = I * NB + 2 * j16;
                T2 a_nxt[KS], b_nxt[KS];                         // (.x, .y) = the two halves cb / ib: adjacent rows, one 16-byte load
                auto fetch = [&](int kk) {
                    const int K = kk / NB, stride = Np - NB * (K + 1);   // (wave-uniform: scalar registers)
                    // (the row bias -32 (K + 1) of a packed column goes into the per-lane part: a scalar offset is unsigned)
                    const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
                    const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        const int so = (base + 4 * s_ * stride) * ES;
                        a_nxt[s_] = P::bload2(rsL, va, so);
                        b_nxt[s_] = P::bload2(rsL, vb, so);
                    }
                };
                if (col0 > 0) fetch(0);
                for (int kk = 0; kk < col0; kk += 4 * KS) {
                    T a_cur[KS][2], b_cur[KS][2];
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y;
                        b_cur[s_][0] = b_nxt[s_].x; b_cur[s_][1] = b_nxt[s_].y;
                    }
                    if (kk + 4 * KS < col0) fetch(kk + 4 * KS);
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib)
                                acc[cb][ib] = P::mfma(a_cur[s_][cb], b_cur[s_][ib], acc[cb][ib]);
                }
            };
            auto update2 = [&](acc_t (&acc0)[2][2], acc_t (&acc1)[2][2], int I) {       // tiles I and I + 1
                constexpr int KS = (sizeof(T) == 4 ? BCBF_RW32_KS : BCBF_RW64_KS) / 2;
                const int irow = I * NB + 2 * j16;
                T2 a_nxt[KS], b0_nxt[KS], b1_nxt[KS];
                auto fetch = [&](int kk) {
                    const int K = kk / NB, stride = Np - NB * (K + 1);
                    const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
                    const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        const int so = (base + 4 * s_ * stride) * ES;
                        a_nxt[s_] = P::bload2(rsL, va, so);
                        b0_nxt[s_] = P::bload2(rsL, vb, so);
                        b1_nxt[s_] = P::bload2(rsL, vb + NB * ES, so);
                    }
                };
                if (col0 > 0) fetch(0);
                for (int kk = 0; kk < col0; kk += 4 * KS) {
                    T a_cur[KS][2], b0_cur[KS][2], b1_cur[KS][2];
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y;
                        b0_cur[s_][0] = b0_nxt[s_].x; b0_cur[s_][1] = b0_nxt[s_].y;
                        b1_cur[s_][0] = b1_nxt[s_].x; b1_cur[s_][1] = b1_nxt[s_].y;
                    }
                    if (kk + 4 * KS < col0) fetch(kk + 4 * KS);
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) {
                                acc0[cb][ib] = P::mfma(a_cur[s_][cb], b0_cur[s_][ib], acc0[cb][ib]);
                                acc1[cb][ib] = P::mfma(a_cur[s_][cb], b1_cur[s_][ib], acc1[cb][ib]);
                            }
                }
            };
            // ---- panel:  L_IJ' = inv(L_JJ) S'   (accumulator registers of -S' are the B operands, ainv = -inv(L_JJ))
            auto solve_store_s = [&](acc_t (&acc)[2][2], int I, int Jc, const T (&ainv)[2][2][4]) {
                const int col0 = Jc * NB;                               // (shadows the loop's col0 / ainv)
                const int irow = I * NB + 2 * j16;
#pragma unroll
                for (int cbp = 0; cbp < 2; ++cbp) {                     // (one half of the output columns at a time: 2 results live)
                    acc_t y[2];                                         // [ib]
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        acc_t yy = {0, 0, 0, 0};
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < (cbp == 0 ? P::PANEL_R0 : 4); ++r)
                                yy = P::mfma(ainv[cbp][cb][r], acc[cb][ib][r], yy);
                        y[ib] = yy;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * cbp + P::midx(r, g);
                        T2 v; v.x = y[0][r]; v.y = y[1][r];             // rows irow, irow + 1 adjacent: one 16-byte store
                        *reinterpret_cast<T2*>(lop + lop_base<V>(col0 + c, Np) + irow) = v;
                        if (Ld && col0 + c < N) {
                            if (irow < N) Ld[(size_t)irow * N + col0 + c] = v.x;
                            if (irow + 1 < N) Ld[(size_t)(irow + 1) * N + col0 + c] = v.y;
                        }
                    }
                }
            };

            auto solve_store = [&](acc_t (&acc)[2][2], int I) { solve_store_s(acc, I, J, ainv); };

            // =================== the diagonal tile: factor L_JJ, invert it ===================
            auto diag_tile_s = [&](acc_t (&acc)[2][2], int Jc, T (&ainv)[2][2][4]) {
                const int J = Jc, col0 = Jc * NB;                       // (shadow the loop's J / col0 / ainv)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = -acc[cb][ib];
                // diag_tile64.h: on the matrix cores, out of the accumulators; xinv -> inv(L), tile -> L (dense output only)
                const int bad = diag_factor_invert_acc<T>(BCBF_LDS_TILE(T, sh.d), acc, lane, Ld != nullptr);
                RW_ACC(3);                                        // factor + inverse
                if (bad != 0 && col0 + bad <= N) fail = col0 + bad;
                if (Ld && lane < NB && col0 + lane < N) {
                    for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = sh.d.tile[lane][c];
                }
                {
                    // global copies from LDS, contiguous across the wave, 16-byte stores: the full column-major tile
                    // (shared-model kernel) and the packed lower triangle (streaming kernels)
                    const int bfull = lop_dfull_block(J, Np), bpack = lop_dinv_block(J, Np);
#pragma unroll
                    for (int t = 0; t < NB * NB / 128; ++t) {
                        const int e = 2 * lane + 128 * t, c = e >> 5, r = e & 31;
                        T2 v; v.x = sh.d.xinv[r][c]; v.y = sh.d.xinv[r + 1][c];
                        *reinterpret_cast<T2*>(lop + bfull + e) = v;
                    }
#pragma unroll
                    for (int t = 0; t < (LOP_DB + 127) / 128; ++t) {
                        const int k = 2 * lane + 128 * t;              // two consecutive entries of the packed triangle
                        if (k < LOP_DB) {
                            const unsigned rc = sh.pack_rc[k >> 1];    // (r0 | c0 << 8 | r1 << 16 | c1 << 24), 0xff.. = padding
                            const int r0 = rc & 0xff, c0 = (rc >> 8) & 0xff, r1 = (rc >> 16) & 0xff, c1 = rc >> 24;
                            T2 v;
                            v.x = r0 < NB ? sh.d.xinv[r0][c0] : T(0.0);
                            v.y = r1 < NB ? sh.d.xinv[r1][c1] : T(0.0);
                            *reinterpret_cast<T2*>(lop + bpack + k) = v;
                        }
                    }
                }
                if (fail != 0) return;
                // A operands of the panel solve: output row c' = 16 cbp + j16, contraction index c = 2 (4r + g) + cb (the
                // column an accumulator register of S' holds).  inv(L_JJ) is lower triangular: c' < 16 meets c < 16 only,
                // that is r < 2
#pragma unroll
                for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ainv[cbp][cb][r] = -sh.d.xinv[16 * cbp + j16][2 * P::midx(r, g) + cb];   // (acc = -S')
            };
            auto diag_tile = [&](acc_t (&acc)[2][2]) { diag_tile_s(acc, J, ainv); };
            if constexpr (SUP) {
            if (J + 1 < nblk) {
                // ============ SUPER-PANEL: block columns J and J + 1 in one pass over everything to their left ============
                // Left-looking with 32-column panels reads, for every tile (I, J), the block rows I and J of all columns
                // left of J: 2 J panels per tile, 5.4 MB per instance at N = 512 for a 0.5 MB factor -- and with ~2000
                // instances in flight (1 GB) none of it hits a cache: C3 fp32 fetched 17.3 GB per launch and ran at the
                // speed of those re-reads (profiles/r03_pmc_traffic_refit.json).  A wave that holds the 2 x 2 tiles
                // (I, I+1) x (J, J+1) at once reads FOUR block rows (J, J+1, I, I+1) for FOUR tiles: one panel per tile
                // instead of 1.5 (pairs) or 2 (single tiles).  What a column still owes its left neighbour INSIDE the
                // super-panel -- tile (I, J+1) needs L_IJ L_{J+1,J}' -- is a 32-deep update from panels this wave has just
                // stored (L2 / L1 hits), applied after the tiles of column J are solved.
                const int col1 = col0 + NB;
                T ainv1[2][2][4];
                auto load_ainv = [&](T (&av)[2][2][4], auto second) {
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int rr_ = 16 * cbp + j16, cc_ = 2 * P::midx(r, g) + cb;
                                av[cbp][cb][r] = -(decltype(second)::value ? sh.d.xinv[rr_][cc_] : sh.xinvJ[rr_][cc_]);
                            }
                };
                // panel solve that also hands the solved tile back: yk[cbp][ib][r] = L_IJ'(c = 16 cbp + midx(r, g), i = irow + ib)
                auto solve_store_keep = [&](acc_t (&acc)[2][2], int I, int Jc, const T (&av)[2][2][4], acc_t (&yk)[2][2]) {
                    const int cbase = Jc * NB, irow = I * NB + 2 * j16;
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp) {
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) {
                            acc_t yy = {0, 0, 0, 0};
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                                for (int r = 0; r < (cbp == 0 ? P::PANEL_R0 : 4); ++r)
                                    yy = P::mfma(av[cbp][cb][r], acc[cb][ib][r], yy);
                            yk[cbp][ib] = yy;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * cbp + P::midx(r, g);
                            T2 v; v.x = yk[cbp][0][r]; v.y = yk[cbp][1][r];
                            *reinterpret_cast<T2*>(lop + lop_base<V>(cbase + c, Np) + irow) = v;
                        }
                    }
                };
                // A operand of the in-register update: L_{J+1,J}[rows 2 j16, 2 j16 + 1 of block J + 1][column col0 + 16 cbp + midx(r, g)]
                // (fp32 only: in fp64, where the extra live registers are 32 + 32, 1024 x 1024 takes 13.8 ms with it, 12.8 without.
                //  One wave per SIMD: 1024 x 512 0.99 -> 0.95 ms, 4096 x 1024 21.3 -> 20.3 ms.  Two waves per SIMD: the operand is
                //  re-read at every row pair (L2 hits) instead of held -- 4096 x 512 3.14 -> 3.09 ms; with the round's first value
                //  pass, which left fewer registers, the same switch cost 3 %)
                constexpr bool INREG = BCBF_RW_SUPER_INREG && (OCC == 1 || BCBF_RW_SUPER_INREG_OCC2) && sizeof(T) == 4;
                constexpr bool A2_RELOAD = BCBF_RW_SUPER_A2_RELOAD || OCC == 2;
                T2 a2[2][4];
                auto load_a2 = [&]() {
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            a2[cbp][r] = *reinterpret_cast<const T2*>(lop + lop_base<V>(col0 + 16 * cbp + P::midx(r, g), Np) + col1 + 2 * j16);
                };
                // -S'(I, J+1) += L_{J+1,J} L_IJ'  with L_IJ' out of the registers of its solve
                auto inner_update = [&](acc_t (&t)[2][2], const acc_t (&yk)[2][2]) {
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) {
                                t[0][ib] = P::mfma(a2[cbp][r].x, yk[cbp][ib][r], t[0][ib]);
                                t[1][ib] = P::mfma(a2[cbp][r].y, yk[cbp][ib][r], t[1][ib]);
                            }
                };
                auto keep_first_inverse = [&]() {              // d.xinv -> xinvJ (the second factorisation overwrites d.xinv)
                    if (BCBF_RW_SUPER_AINV_LDS) {
                        for (int e = lane; e < NB * NB; e += 64) sh.xinvJ[e >> 5][e & 31] = sh.d.xinv[e >> 5][e & 31];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                };
                constexpr int KS4 = sizeof(T) == 4 ? (OCC == 1 ? 8 : BCBF_RW32_SUPER_KS) : BCBF_RW64_SUPER_KS;
                auto negate = [&](acc_t (&acc)[2][2]) { (void)acc; };
                (void)negate;
                // -S' += L_a L_b' for k in [k0, k1): one tile, A operand = block row at arow0, B operand = rows of tile I
                auto update_r = [&](acc_t (&acc)[2][2], int arow0, int I, int k0, int k1) {
                    constexpr int KS = sizeof(T) == 4 ? 4 : 2;
                    const int irow = I * NB + 2 * j16;
                    T2 a_nxt[KS], b_nxt[KS];
                    auto fetch = [&](int kk) {
                        const int K = kk / NB, stride = Np - NB * (K + 1);
                        const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
                        const int va = (g * stride + arow0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_) {
                            const int so = (base + 4 * s_ * stride) * ES;
                            a_nxt[s_] = P::bload2(rsL, va, so);
                            b_nxt[s_] = P::bload2(rsL, vb, so);
                        }
                    };
                    if (k1 > k0) fetch(k0);
                    for (int kk = k0; kk < k1; kk += 4 * KS) {
                        T a_cur[KS][2], b_cur[KS][2];
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_) {
                            a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y;
                            b_cur[s_][0] = b_nxt[s_].x; b_cur[s_][1] = b_nxt[s_].y;
                        }
                        if (kk + 4 * KS < k1) fetch(kk + 4 * KS);
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                                for (int ib = 0; ib < 2; ++ib)
                                    acc[cb][ib] = P::mfma(a_cur[s_][cb], b_cur[s_][ib], acc[cb][ib]);
                    }
                };
                // ... two tiles I, I + 1 of one block column (shared A operand)
                auto update2_r = [&](acc_t (&acc0)[2][2], acc_t (&acc1)[2][2], int arow0, int I, int k0, int k1) {
                    constexpr int KS = sizeof(T) == 4 ? 4 : 2;
                    const int irow = I * NB + 2 * j16;
                    T2 a_nxt[KS], b0_nxt[KS], b1_nxt[KS];
                    auto fetch = [&](int kk) {
                        const int K = kk / NB, stride = Np - NB * (K + 1);
                        const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
                        const int va = (g * stride + arow0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_) {
                            const int so = (base + 4 * s_ * stride) * ES;
                            a_nxt[s_] = P::bload2(rsL, va, so);
                            b0_nxt[s_] = P::bload2(rsL, vb, so);
                            b1_nxt[s_] = P::bload2(rsL, vb + NB * ES, so);
                        }
                    };
                    if (k1 > k0) fetch(k0);
                    for (int kk = k0; kk < k1; kk += 4 * KS) {
                        T a_cur[KS][2], b0_cur[KS][2], b1_cur[KS][2];
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_) {
                            a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y;
                            b0_cur[s_][0] = b0_nxt[s_].x; b0_cur[s_][1] = b0_nxt[s_].y;
                            b1_cur[s_][0] = b1_nxt[s_].x; b1_cur[s_][1] = b1_nxt[s_].y;
                        }
                        if (kk + 4 * KS < k1) fetch(kk + 4 * KS);
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                                for (int ib = 0; ib < 2; ++ib) {
                                    acc0[cb][ib] = P::mfma(a_cur[s_][cb], b0_cur[s_][ib], acc0[cb][ib]);
                                    acc1[cb][ib] = P::mfma(a_cur[s_][cb], b1_cur[s_][ib], acc1[cb][ib]);
                                }
                    }
                };
                // ... the 2 x 2 tiles (I, I+1) x (J, J+1) over k in [0, col0): four operand streams for four tiles.  NR = 1: only
                // row block I (the odd tile at the bottom), three streams for two tiles.  I == J + 1 - 1 ... the diagonal 2 x 2
                // block is the same pass with the B streams BEING the A streams (DIAG: tiles (J,J), (J+1,J), (J+1,J+1); the
                // fourth accumulator is not used)
                auto update4 = [&](acc_t (&a00)[2][2], acc_t (&a01)[2][2], acc_t (&a10)[2][2], acc_t (&a11)[2][2], int I, auto nr,
                                   auto diag) {
                    constexpr int NR = decltype(nr)::value;
                    constexpr int DIAG = (int)decltype(diag)::value;      // 1: tiles (J,J), (J+1,J), (J+1,J+1) from the two A streams; 2: the last two
                    constexpr int KS = KS4;
                    const int irow = I * NB + 2 * j16;
                    T2 aj_nxt[KS], ak_nxt[KS], b0_nxt[DIAG ? 1 : KS], b1_nxt[(DIAG || NR < 2) ? 1 : KS];
                    auto fetch = [&](int kk) {
                        const int K = kk / NB, stride = Np - NB * (K + 1);
                        const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
                        const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_) {
                            const int so = (base + 4 * s_ * stride) * ES;
                            aj_nxt[s_] = P::bload2(rsL, va, so);
                            ak_nxt[s_] = P::bload2(rsL, va + NB * ES, so);
                            if constexpr (!DIAG) {
                                b0_nxt[s_] = P::bload2(rsL, vb, so);
                                if constexpr (NR == 2) b1_nxt[s_] = P::bload2(rsL, vb + NB * ES, so);
                            }
                        }
                    };
                    if (col0 > 0) fetch(0);
                    for (int kk = 0; kk < col0; kk += 4 * KS) {
                        T aj[KS][2], ak[KS][2], b0[KS][2], b1[KS][2];
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_) {
                            aj[s_][0] = aj_nxt[s_].x; aj[s_][1] = aj_nxt[s_].y;
                            ak[s_][0] = ak_nxt[s_].x; ak[s_][1] = ak_nxt[s_].y;
                            if constexpr (!DIAG) {
                                b0[s_][0] = b0_nxt[s_].x; b0[s_][1] = b0_nxt[s_].y;
                                if constexpr (NR == 2) { b1[s_][0] = b1_nxt[s_].x; b1[s_][1] = b1_nxt[s_].y; }
                            }
                        }
                        if (kk + 4 * KS < col0) fetch(kk + 4 * KS);
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                                for (int ib = 0; ib < 2; ++ib) {
                                    if constexpr (DIAG != 0) {
                                        if constexpr (DIAG == 1) a00[cb][ib] = P::mfma(aj[s_][cb], aj[s_][ib], a00[cb][ib]);      // (J,   J)
                                        a10[cb][ib] = P::mfma(aj[s_][cb], ak[s_][ib], a10[cb][ib]);      // (J+1, J)
                                        a11[cb][ib] = P::mfma(ak[s_][cb], ak[s_][ib], a11[cb][ib]);      // (J+1, J+1)
                                    } else {
                                        a00[cb][ib] = P::mfma(aj[s_][cb], b0[s_][ib], a00[cb][ib]);      // (I,   J)
                                        a01[cb][ib] = P::mfma(ak[s_][cb], b0[s_][ib], a01[cb][ib]);      // (I,   J+1)
                                        if constexpr (NR == 2) {
                                            a10[cb][ib] = P::mfma(aj[s_][cb], b1[s_][ib], a10[cb][ib]);  // (I+1, J)
                                            a11[cb][ib] = P::mfma(ak[s_][cb], b1[s_][ib], a11[cb][ib]);  // (I+1, J+1)
                                        }
                                    }
                                }
                    }
                };
                using IC0 = std::integral_constant<int, 0>;
                using IC1 = std::integral_constant<int, 1>;
                using IC2 = std::integral_constant<int, 2>;
                // stores of this wave feed loads of other lanes of this wave below: wave scope, program order -- the fence keeps
                // the compiler from moving the loads up and drains the stores first
                auto wave_fence = [&]() {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                };
                // ---- the diagonal 2 x 2 block
                load_rows(rw0, J);
                load_rows(rw1, J + 1);
                {
                    acc_t t10[2][2], t11[2][2];
#if BCBF_RW32_SUPER_DIAG3
                    acc_t t00[2][2];
                    values_s(t00, rw0, J, IC0{});
                    values_s(t10, rw1, J + 1, IC0{});
                    values_s(t11, rw1, J + 1, IC1{});
                    load_rows(rw0, J + 2);
                    load_rows(rw1, J + 3);
                    RW_ACC(1);
                    update4(t00, t00, t10, t11, J, IC2{}, std::true_type{});
                    RW_ACC(2);
                    diag_tile_s(t00, J, ainv);
                    keep_first_inverse();
                    RW_ACC(4);
                    if (fail != 0) break;
#else
                    {   // (the diagonal tile by itself first: its factorisation wants the registers the other two tiles would hold)
                        acc_t t00[2][2];
                        values_s(t00, rw0, J, IC0{});
                        update_r(t00, col0, J, 0, col0);
                        diag_tile_s(t00, J, ainv);
                        keep_first_inverse();
                        if (fail != 0) break;
                    }
                    values_s(t10, rw1, J + 1, IC0{});
                    values_s(t11, rw1, J + 1, IC1{});
                    load_rows(rw0, J + 2);
                    load_rows(rw1, J + 3);
                    update4(t10, t10, t10, t11, J, IC2{}, std::integral_constant<int, 2>{});
#endif
                    solve_store_s(t10, J + 1, J, ainv);                     // L_{J+1,J}
                    wave_fence();
                    RW_ACC(5);
                    if (INREG && !A2_RELOAD) load_a2();
                    update_r(t11, col1, J + 1, col0, col1);                 // -= L_{J+1,J} L_{J+1,J}'
                    RW_ACC(2);
                    diag_tile_s(t11, J + 1, ainv1);
                    RW_ACC(4);
                    if (fail != 0) break;
                }
                // ---- the tiles below it: two block rows x two block columns per pass
                for (int I = J + 2; I < nblk; I += 2) {
                    acc_t t00[2][2], t01[2][2];
                    if (I + 1 < nblk) {
                        acc_t t10[2][2], t11[2][2];
                        values_s(t00, rw0, I, IC0{});
                        values_s(t01, rw0, I, IC1{});
                        values_s(t10, rw1, I + 1, IC0{});
                        values_s(t11, rw1, I + 1, IC1{});
                        if (BCBF_RW32_SUPER_AHEAD) { load_rows(rw0, I + 2); load_rows(rw1, I + 3); }
                        RW_ACC(1);
                        update4(t00, t01, t10, t11, I, IC2{}, std::false_type{});
                        RW_ACC(2);
                        if (INREG && BCBF_RW_SUPER_AINV_LDS) {
                            T av[2][2][4];
                            acc_t y0[2][2], y1[2][2];
                            if (A2_RELOAD) load_a2();
                            load_ainv(av, std::false_type{});
                            solve_store_keep(t00, I, J, av, y0);
                            inner_update(t01, y0);
                            solve_store_keep(t10, I + 1, J, av, y1);
                            inner_update(t11, y1);
                            RW_ACC(5);
                        } else {
                        if (BCBF_RW_SUPER_AINV_LDS) {
                            T av[2][2][4];
                            load_ainv(av, std::false_type{});
                            solve_store_s(t00, I, J, av);
                            solve_store_s(t10, I + 1, J, av);
                        } else {
                            solve_store_s(t00, I, J, ainv);
                            solve_store_s(t10, I + 1, J, ainv);
                        }
                        wave_fence();
                        RW_ACC(5);
                        update2_r(t01, t11, col1, I, col0, col1);           // -= L_{I,J} L_{J+1,J}',  L_{I+1,J} L_{J+1,J}'
                        }
                        RW_ACC(2);
                        if (BCBF_RW_SUPER_AINV_LDS) {
                            T av[2][2][4];
                            load_ainv(av, std::true_type{});
                            solve_store_s(t01, I, J + 1, av);
                            solve_store_s(t11, I + 1, J + 1, av);
                        } else {
                            solve_store_s(t01, I, J + 1, ainv1);
                            solve_store_s(t11, I + 1, J + 1, ainv1);
                        }
                        if (!BCBF_RW32_SUPER_AHEAD) { load_rows(rw0, I + 2); load_rows(rw1, I + 3); }
                        RW_ACC(5);
                    } else {
                        values_s(t00, rw0, I, IC0{});
                        values_s(t01, rw0, I, IC1{});
                        update4(t00, t01, t00, t01, I, IC1{}, std::false_type{});
                        if (INREG && BCBF_RW_SUPER_AINV_LDS) {
                            T av[2][2][4];
                            acc_t y0[2][2];
                            if (A2_RELOAD) load_a2();
                            load_ainv(av, std::false_type{});
                            solve_store_keep(t00, I, J, av, y0);
                            inner_update(t01, y0);
                            load_ainv(av, std::true_type{});
                            solve_store_s(t01, I, J + 1, av);
                        } else if (BCBF_RW_SUPER_AINV_LDS) {
                            T av[2][2][4];
                            load_ainv(av, std::false_type{});
                            solve_store_s(t00, I, J, av);
                            wave_fence();
                            update_r(t01, col1, I, col0, col1);
                            load_ainv(av, std::true_type{});
                            solve_store_s(t01, I, J + 1, av);
                        } else {
                            solve_store_s(t00, I, J, ainv);
                            wave_fence();
                            update_r(t01, col1, I, col0, col1);
                            solve_store_s(t01, I, J + 1, ainv1);
                        }
                    }
                }
                ++J;                                                        // (the loop's own ++J makes it two)
                continue;
            }
            }
            load_rows(rw0, J);
            {
                {
                    acc_t acc[2][2];
                    values(acc, rw0, J);
                    load_rows(rw0, J + 1);
                    load_rows(rw1, J + 2);
                    RW_ACC(1);
                    update(acc, J);
                    RW_ACC(2);
                    diag_tile(acc);
                    RW_ACC(4);
                    if (fail != 0) break;
                }
                // the tiles below it, two at a time
                for (int I = J + 1; I < nblk; I += 2) {
                    acc_t acc0[2][2], acc1[2][2];
                    if (I + 1 < nblk) {
                        values(acc0, rw0, I);
                        values(acc1, rw1, I + 1);
                        load_rows(rw0, I + 2);
                        load_rows(rw1, I + 3);
                        RW_ACC(1);
                        update2(acc0, acc1, I);
                        RW_ACC(2);
                        solve_store(acc0, I);
                        solve_store(acc1, I + 1);
                    } else {
                        values(acc0, rw0, I);
                        RW_ACC(1);
                        update(acc0, I);
                        RW_ACC(2);
                        solve_store(acc0, I);
                    }
                    RW_ACC(5);
                }
            }
        } else {
            // inputs of a tile's two rows per lane (x_i, (UH B)_i, jitter_i): loaded one tile ahead, so that the loads are in
            // flight during the previous tile's update stream instead of queueing behind its panel stores
            T rx[2][RXD], ru[2][BCBF_MAX_CTRL_DIM + 1], rj[2];
            // Branch-free: every load is issued, a component the model does not have (d >= n, c >= C) or a row past the end
            // is an out-of-range buffer offset and reads as zero.  (Written with `if (d < n)` around plain loads each one
            // became its own basic block with a full wait behind it: ten serialized memory round trips per tile, 44 % of
            // the kernel at N = 256.)
            auto load_rows = [&](int I_) {
                if (FROM_DENSE) return;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = I_ * NB + 2 * j16 + ib;
                    const bool in = I_ < nblk && i < N;
#pragma unroll
                    for (int d = 0; d < RXD; ++d)
                        rx[ib][d] = P::bload(rsX, (in && d < n) ? (i * n + d) * ES : -ES);
#pragma unroll
                    for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c)
                        ru[ib][c] = P::bload(rsU, (in && c < C) ? (i * C + c) * ES : -ES);
                    rj[ib] = P::bload(rsJ, in ? i * ES : -ES);
                }
            };
            load_rows(J);
            for (int I = J; I < nblk; ++I) {
                const int irow = I * NB + 2 * j16;                    // + ib
                acc_t acc[2][2];                                      // [cb][ib]: -S'[c = 2 (4r + g) + cb][i = 2 j16 + ib]  (halves interleaved:
                                                                      // the two halves of an operand are adjacent rows, one 16-byte load)
                // ---- initial value K_b'(c, i).  Straight-line: the column's inputs come from LDS as fixed-width rows (4 state
                //      components, 4 control components; zeros beyond n / C, where iell and the row inputs are zero too), ONE
                //      read per column for both rows of the lane.  (With `if (d < n)` around each LDS read every read sat in a
                //      basic block of its own with a full wait behind it: 120 serialized LDS round trips per tile.)
                if (FROM_DENSE) {
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        const int i = irow + ib;
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 2 * P::midx(r, g) + cb, j = col0 + c;
                                T val;
                                if (i >= N || j >= N) val = (i == j) ? T(1.0) : T(0.0);      // padding: identity
                                else val = (j <= i) ? Kb[(size_t)i * N + j] : Kb[(size_t)j * N + i];
                                acc[cb][ib][r] = -val;
                            }
                    }
                } else if (BCBF_RW_VALUES_FAST && I != J && (I + 1) * NB <= N && col0 + NB <= N && n <= 4) {
                    // (the lean value pass of the tile-pair path above, for the single-tile stream: interior tiles; inputs pinned)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        asm volatile("" :: "v"(rj[ib]));
#pragma unroll
                        for (int d = 0; d < RXD; ++d) asm volatile("" :: "v"(rx[ib][d]));
#pragma unroll
                        for (int a = 0; a < BCBF_MAX_CTRL_DIM + 1; ++a) asm volatile("" :: "v"(ru[ib][a]));
                    }
                    const T ms2 = -s2;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 2 * P::midx(r, g) + cb;
                            T cx[4], cu[4];
#pragma unroll
                            for (int d = 0; d < 4; ++d) { cx[d] = sh.colX[c][d]; cu[d] = sh.colUH[c][d]; }
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) {
                                T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                                for (int d = 0; d < 4; ++d) { const T z = (rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
#pragma unroll
                                for (int a = 0; a < 4; ++a) uu += ru[ib][a] * cu[a];
                                acc[cb][ib][r] = ms2 * P::exp_neg(T(T(0.5)) * d2) * uu;
                            }
                        }
                } else {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 2 * P::midx(r, g) + cb, j = col0 + c;
                            T cx[4], cu[4];
#pragma unroll
                            for (int d = 0; d < 4; ++d) { cx[d] = sh.colX[c][d]; cu[d] = sh.colUH[c][d]; }
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) {
                                const int i = irow + ib;
                                T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                                for (int d = 0; d < 4; ++d) { const T z = (rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
                                if (n > 4) {                                          // (wave-uniform; no reference system has n > 4)
                                    for (int d = 4; d < n; ++d) {
                                        const T xi = i < N ? Xb[(size_t)i * n + d] : T(0.0);
                                        const T z = (xi - sh.colX[c][d]) / ell[(size_t)b * n + d];
                                        d2 += z * z;
                                    }
                                }
#pragma unroll
                                for (int a = 0; a < 4; ++a) uu += ru[ib][a] * cu[a];
                                T val = s2 * P::exp_neg(T(T(0.5)) * d2) * uu + (i == j ? rj[ib] : T(0.0));
                                val = (i >= N || j >= N) ? ((i == j) ? T(1.0) : T(0.0)) : val;   // padding: identity
                                acc[cb][ib][r] = -val;                // the accumulators carry -S' (see the update stream)
                            }
                        }
                }
                load_rows(I + 1);
                RW_ACC(1);                                            // K_b values
                // ---- -S' += L_J L_I'  over all previous columns (software pipelined: next stage's operands in flight).  The
                //      accumulators hold -S': the MFMA has no negate modifier, and flipping an operand costs four VALU instructions
                //      per k-step against once per tile at the consumer.  Addresses: inside block column K the packed columns are
                //      a fixed stride apart, so a load is  buffer base + SCALAR offset (column block, k-step) + per-lane offset
                //      (lane group's column, row): two multiply-adds per fetch instead of the column-offset polynomial per load
                //      (a wave issues in order: address arithmetic does not hide behind its own MFMAs)
                constexpr int KS = sizeof(T) == 4 ? BCBF_RW32_KS : BCBF_RW64_KS;
                T2 a_nxt[KS], b_nxt[KS];                         // (.x, .y) = the two halves cb / ib: adjacent rows, one 16-byte load
                auto fetch = [&](int kk) {
                    const int K = kk / NB, stride = Np - NB * (K + 1);   // (wave-uniform: scalar registers)
                    // (the row bias -32 (K + 1) of a packed column goes into the per-lane part: a scalar offset is unsigned)
                    const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
                    const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        const int so = (base + 4 * s_ * stride) * ES;
                        a_nxt[s_] = P::bload2(rsL, va, so);
                        b_nxt[s_] = P::bload2(rsL, vb, so);
                    }
                };
                if (col0 > 0) fetch(0);
                for (int kk = 0; kk < col0; kk += 4 * KS) {
                    T a_cur[KS][2], b_cur[KS][2];
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y;
                        b_cur[s_][0] = b_nxt[s_].x; b_cur[s_][1] = b_nxt[s_].y;
                    }
                    if (kk + 4 * KS < col0) fetch(kk + 4 * KS);
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib)
                                acc[cb][ib] = P::mfma(a_cur[s_][cb], b_cur[s_][ib], acc[cb][ib]);
                }

                RW_ACC(2);                                            // update stream
                if (I == J) {
                    // =================== the diagonal tile: factor L_JJ, invert it ===================
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = -acc[cb][ib];
                    // diag_tile64.h: on the matrix cores, out of the accumulators; xinv -> inv(L), tile -> L (dense output only)
                    const int bad = diag_factor_invert_acc<T>(BCBF_LDS_TILE(T, sh.d), acc, lane, Ld != nullptr);
                    RW_ACC(3);                                        // factor + inverse
                    if (bad != 0 && col0 + bad <= N) fail = col0 + bad;
                    if (Ld && lane < NB && col0 + lane < N) {
                        for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = sh.d.tile[lane][c];
                    }
                    {
                        // global copies from LDS, contiguous across the wave, 16-byte stores: the full column-major tile
                        // (shared-model kernel) and the packed lower triangle (streaming kernels)
                        const int bfull = lop_dfull_block(J, Np), bpack = lop_dinv_block(J, Np);
#pragma unroll
                        for (int t = 0; t < NB * NB / 128; ++t) {
                            const int e = 2 * lane + 128 * t, c = e >> 5, r = e & 31;
                            T2 v; v.x = sh.d.xinv[r][c]; v.y = sh.d.xinv[r + 1][c];
                            *reinterpret_cast<T2*>(lop + bfull + e) = v;
                        }
#pragma unroll
                        for (int t = 0; t < (LOP_DB + 127) / 128; ++t) {
                            const int k = 2 * lane + 128 * t;              // two consecutive entries of the packed triangle
                            if (k < LOP_DB) {
                                const unsigned rc = sh.pack_rc[k >> 1];    // (r0 | c0 << 8 | r1 << 16 | c1 << 24), 0xff.. = padding
                                const int r0 = rc & 0xff, c0 = (rc >> 8) & 0xff, r1 = (rc >> 16) & 0xff, c1 = rc >> 24;
                                T2 v;
                                v.x = r0 < NB ? sh.d.xinv[r0][c0] : T(0.0);
                                v.y = r1 < NB ? sh.d.xinv[r1][c1] : T(0.0);
                                *reinterpret_cast<T2*>(lop + bpack + k) = v;
                            }
                        }
                    }
                    if (fail != 0) break;
                    // A operands of the panel solve: output row c' = 16 cbp + j16, contraction index c = 2 (4r + g) + cb (the
                    // column an accumulator register of S' holds).  inv(L_JJ) is lower triangular: c' < 16 meets c < 16 only,
                    // that is r < 2
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) ainv[cbp][cb][r] = -sh.d.xinv[16 * cbp + j16][2 * P::midx(r, g) + cb];   // (acc = -S')
                    RW_ACC(4);                                        // stores of the inverse
                } else {
                    // ---- panel:  L_IJ' = inv(L_JJ) S'   (accumulator registers of -S' are the B operands, ainv = -inv(L_JJ))
                    acc_t y[2][2];                                      // [ib][cbp]
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int cbp = 0; cbp < 2; ++cbp) {
                            acc_t yy = {0, 0, 0, 0};
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                                for (int r = 0; r < (cbp == 0 ? P::PANEL_R0 : 4); ++r)
                                    yy = P::mfma(ainv[cbp][cb][r], acc[cb][ib][r], yy);
                            y[ib][cbp] = yy;
                        }
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * cbp + P::midx(r, g);
                            T2 v; v.x = y[0][cbp][r]; v.y = y[1][cbp][r];   // rows irow, irow + 1 adjacent: one 16-byte store
                            *reinterpret_cast<T2*>(lop + lop_base<V>(col0 + c, Np) + irow) = v;
                            if (Ld && col0 + c < N) {
                                if (irow < N) Ld[(size_t)irow * N + col0 + c] = v.x;
                                if (irow + 1 < N) Ld[(size_t)(irow + 1) * N + col0 + c] = v.y;
                            }
                        }
                    RW_ACC(5);                                        // panel solve + stores
                }
            }
        }
        __threadfence_block();                 // this block column's panels are read back (by other lanes) from here on
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) info[b] = fail;
#ifdef BCBF_RW64_PROF
    if (Ld && lane == 0 && N > 8)
        for (int k = 0; k < 6; ++k) Ld[1 + k] = (T)prof[k];
    // (no dense output: the counters go above the diagonal of the first full inverse tile -- column 31, rows 0..5, zeros
    //  otherwise; development builds only)
    if (!Ld && lane == 0)
        for (int k = 0; k < 6; ++k) lop[lop_dfull_block(0, Np) + NB * 31 + k] = (T)prof[k];
#endif
}

#ifndef BCBF_RP_KS64
#define BCBF_RP_KS64 2           // k-steps (of 4 columns) per pipeline stage of the update stream, fp64 (4: no faster, spills more)
#endif
#ifndef BCBF_RP_KS32
#define BCBF_RP_KS32 4           // ... fp32
#endif
#ifndef BCBF_RP_ACACHE
#define BCBF_RP_ACACHE 0         // fp64: the bulk wave keeps the two left-most tiles of block row J (the A operand every tile of column J
                                 // shares) in LDS for the column: 25 of the 168 panel-tile reads of a C2 instance gone.  OFF -- measured
                                 // (round 5, 1024 x 256 fp64, same box): 0.324 ms with the cache against 0.289 without; the fill at the head
                                 // of a column and the LDS reads inside the stream cost the bulk wave more than the L2 / fabric reads they
                                 // replace -- the re-reads are not what the launch waits for, whatever the byte count says
#endif
// -DBCBF_RP_TRACE (development, tools/dev/trace_refit_pair.py): 100 MHz time stamps of workgroup 0's two waves at every hand-off
#ifdef BCBF_RP_TRACE
extern "C" __attribute__((visibility("default"))) int bcbf_debug_rp_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rp_trace_buf), sizeof(rp_trace_buf));
}
#define RP_T(w) do { if (b == 0 && lane == 0 && tix < 4000) rp_trace_buf[w][tix] = wall_clock64(); ++tix; } while (0)
#else
#define RP_T(w) do {} while (0)
#endif
constexpr int RA_MAXBLK = 16;          // block columns the look-ahead form handles (Np <= 512)

// ----------------------------------------------------------------------------------------------------------------
// TWO WAVES PER INSTANCE (round 3): the same left-looking factorisation with its serial chain -- factor the diagonal tile
// of block column J, solve the one panel tile below it, bring the next diagonal tile up to date, factor it -- on wave 0,
// which does NOTHING ELSE; every other tile of the matrix is wave 1's, which works one block column behind:
//   wave 0, the CHAIN:  the diagonal tile's values and its updates over the columns that are complete are formed while it
//           waits; S'_JJ -= L_{J,J-1} L_{J,J-1}' as soon as wave 1 delivers that panel tile; factor + invert on the matrix
//           cores straight out of the accumulators (diag_tile64.h: diag_factor_invert_acc); column count -> LDS word
//           (wave 1 takes the inverse out of LDS); inverse -> global memory.
//   wave 1, the BULK:  S' of the column's first panel tile is formed while it waits; on the column count: inv(L_JJ) ->
//           registers, L_{J+1,J}' = inv(L_JJ) S' -> stored, published (the chain continues on it); then every other tile
//           of column J: values, update stream, solve straight out of the accumulators (no S' parked in memory), store.
// Hand-offs are four counters in LDS (diagonal tiles inverted; first / second panel tiles delivered; columns complete: release
// fence -> word -> acquire fence, workgroup scope: same CU, same L1); wave 1 fences once per column for everything but
// the chain's tile.  Both waves fit the 256-register budget of two waves per SIMD, so a SIMD hosts waves of two different
// instances.  Same outputs, same packed layout, same info convention; no dense output.
// (A first split -- wave 1 streaming every tile's S' through memory, wave 0 factoring the diagonal tiles and solving the
// parked S' -- left the streamer as the critical path, moved a quarter more bytes and was 15 % slower: 0.377 against
// 0.320 ms at 1024 x 256 fp64, tools/bench_refit_forms.py.)
#ifdef BCBF_RP_TRACE
#define RA_T(w) RP_T(w)
#else
#define RA_T(w) do {} while (0)
#endif
constexpr int RP_ACACHE_GROUPS = 16;          // cached k-groups of 4 columns: two 32-column tiles
template <typename T> struct RAShared {
    DiagTile<T> d;                            // wave 0's (d.tile is NOT used by this kernel's diagonal-tile routine: with the A-operand
                                              //  cache on it holds the cache's first tile)
    T arow1[BCBF_RP_ACACHE && sizeof(T) == 8 ? 8 : 1][64][2] __attribute__((aligned(16)));    // ... its second tile
    T colX[2][NB][BCBF_MAX_STATE_DIM];        // [wave]: each wave stages the column block it is forming values for
    T colUH[2][NB][BCBF_MAX_CTRL_DIM + 1];
    unsigned pack_rc[LOP_DB / 2];             // (row, column) of the entries of a packed inverted diagonal block, two per word
    int inv_ready;                            // wave 0 -> 1: diagonal tiles factored, inverse in global memory (a large number after a failed pivot)
    int first_done;                           // wave 1 -> 0: columns J whose first panel tile L_{J+1,J} is complete
    int second_done;                          // wave 1 -> 0: ... whose second panel tile L_{J+2,J} is complete too
    int cols_done;                            // wave 1 -> 0: columns complete
    int fail;
};

template <typename T>
__global__ void __launch_bounds__(128, 2)
refit_pair_kernel(const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ Bm,
                   const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ jitter,
                   T* __restrict__ Lop, T* __restrict__ UHBout, int* __restrict__ info, int Bt, int N, int Np, int n, int C, const int* only_bad) {
    constexpr int V = Vec<T>::V, ES = (int)sizeof(T);
    using P = RW<T>;
    using acc_t = typename P::acc_t;
    using T2 = typename P::vec2;
    __shared__ __attribute__((aligned(16))) RAShared<T> shm;
    __attribute__((address_space(3))) RAShared<T>& sp = *(__attribute__((address_space(3))) RAShared<T>*)&shm;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    if (only_bad != nullptr && only_bad[b] == 0) { if (threadIdx.x == 0) info[b] = 0; return; }     // bcbf_refit_retry: factored already
    const int j16 = lane & 15, g = lane >> 4;
    constexpr bool ACACHE = BCBF_RP_ACACHE && sizeof(T) == 8;
    static_assert(!ACACHE || sizeof(T) * NB * DT_LS >= 8 * 64 * sizeof(T2), "the cache's first tile lives in d.tile");
    // k-group grp (4 columns) of the cached A operand: this lane's two rows (the layout `update` loads them in)
    auto acache = [&](int grp) -> __attribute__((address_space(3))) T* {
        return grp < 8 ? &sp.d.tile[0][0] + (grp * 64 + lane) * 2 : &sp.arow1[ACACHE ? grp - 8 : 0][lane][0];
    };
    T* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    const T* Xb = X + (size_t)b * N * n;
    const T* UHb = UH + (size_t)b * N * C;
    const int nblk = Np / NB;
    {   // UH B rows (both waves), the hand-off words
        T Bmr[(BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1)];
#pragma unroll
        for (int a = 0; a < (BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1); ++a) Bmr[a] = a < C * C ? Bm[(size_t)b * C * C + a] : T(0.0);
        for (int i = threadIdx.x; i < N; i += 128)
            for (int c = 0; c < C; ++c) {
                T s = T(0.0);
                for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                UHBout[((size_t)b * N + i) * C + c] = s;
            }
        if (threadIdx.x == 0) { sp.inv_ready = 0; sp.first_done = 0; sp.second_done = 0; sp.cols_done = 0; sp.fail = 0; }
        rw_pack_table(sp.pack_rc, threadIdx.x, 128);
    }
    __threadfence_block();
    __syncthreads();
    auto wait_for = [&](__attribute__((address_space(3))) int* word, int want) {
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto publish = [&](__attribute__((address_space(3))) int* word, int value) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto failed = [&]() { return __hip_atomic_load(&sp.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };

    // ---- what both waves do to a tile: kernel values, the update stream (each on its own registers and its own column
    //      block in LDS)
    const T* UHBb = UHBout + (size_t)b * N * C;
    // (registers: the first four state components only -- no reference system has more; wider states take the slow
    //  path below, reading the rest from memory)
    T iell[4];
    const T s2 = s2p[b];
#pragma unroll
    for (int d = 0; d < 4; ++d) iell[d] = d < n ? T(1.0) / ell[(size_t)b * n + d] : T(0.0);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Xb), 0, N * n * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(UHBb), 0, N * C * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsJ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(jitter ? jitter + (size_t)b * N : X), 0, jitter ? N * ES : 0, 0x00020000);
    auto& cX = sp.colX[wave];
    auto& cU = sp.colUH[wave];
    // a column block's inputs, zero-filled to fixed widths: loads ISSUED early (into registers, out-of-range offsets read
    // as zero), written to LDS when the previous block's values are done -- the round trip hides under other work
    const __amdgpu_buffer_rsrc_t rsUH = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(UHb), 0, N * C * ES, 0x00020000);
    constexpr int SX = NB * BCBF_MAX_STATE_DIM / 64, SU = NB * (BCBF_MAX_CTRL_DIM + 1) / 64;
    T sx[SX], su[SU];
    auto stage_issue = [&](int J) {
        const int col0 = J * NB;
#pragma unroll
        for (int t = 0; t < SX; ++t) {
            const int e = lane + 64 * t, c = e / BCBF_MAX_STATE_DIM, d = e % BCBF_MAX_STATE_DIM;
            sx[t] = P::bload(rsX, (col0 + c < N && d < n) ? ((col0 + c) * n + d) * ES : -ES);
        }
#pragma unroll
        for (int t = 0; t < SU; ++t) {
            const int e = lane + 64 * t, c = e / (BCBF_MAX_CTRL_DIM + 1), a = e % (BCBF_MAX_CTRL_DIM + 1);
            su[t] = P::bload(rsUH, (col0 + c < N && a < C) ? ((col0 + c) * C + a) * ES : -ES);
        }
    };
    // (fp32 only: in fp64 the 48 registers that stay live across the inverse copy / the column fence spill -- 320 B of
    //  scratch against 104 -- and the kernel is slower for it; there a column's inputs are fetched where they are used)
    constexpr bool EARLY = sizeof(T) == 4;
    auto stage_commit = [&]() {
        __builtin_amdgcn_wave_barrier();                           // every lane is done with the previous column block
#pragma unroll
        for (int t = 0; t < SX; ++t) { const int e = lane + 64 * t; cX[e / BCBF_MAX_STATE_DIM][e % BCBF_MAX_STATE_DIM] = sx[t]; }
#pragma unroll
        for (int t = 0; t < SU; ++t) { const int e = lane + 64 * t; cU[e / (BCBF_MAX_CTRL_DIM + 1)][e % (BCBF_MAX_CTRL_DIM + 1)] = su[t]; }
        __builtin_amdgcn_wave_barrier();
    };
    T rx[2][4], ru[2][BCBF_MAX_CTRL_DIM + 1], rj[2];
    auto load_rows = [&](int I_) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int i = I_ * NB + 2 * j16 + ib;
            const bool in = I_ < nblk && i < N;
#pragma unroll
            for (int d = 0; d < 4; ++d) rx[ib][d] = P::bload(rsX, (in && d < n) ? (i * n + d) * ES : -ES);
#pragma unroll
            for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) ru[ib][c] = P::bload(rsU, (in && c < C) ? (i * C + c) * ES : -ES);
            rj[ib] = P::bload(rsJ, in ? i * ES : -ES);
        }
    };
    // acc[cb][ib][r] = K_b'(column col0 + 2 midx(r, g) + cb, row 32 I + 2 j16 + ib)   (rows of block row I in rx / ru / rj)
    auto values = [&](acc_t (&acc)[2][2], int I, int J) {
        const int col0 = J * NB, irow = I * NB + 2 * j16;
        if (BCBF_RW_VALUES_FAST && I != J && (I + 1) * NB <= N && col0 + NB <= N && n <= 4) {
            // interior tile: the lean value pass of refit_wave_kernel (no jitter / padding selects; row inputs pinned, see there)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                asm volatile("" :: "v"(rj[ib]));
#pragma unroll
                for (int d = 0; d < 4; ++d) asm volatile("" :: "v"(rx[ib][d]));
#pragma unroll
                for (int a = 0; a < 4; ++a) asm volatile("" :: "v"(ru[ib][a]));
            }
            const T ms2 = -s2;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 2 * P::midx(r, g) + cb;
                    T cx[4], cu[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) { cx[d] = cX[c][d]; cu[d] = cU[c][d]; }
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                        for (int d = 0; d < 4; ++d) { const T z = (rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
#pragma unroll
                        for (int a = 0; a < 4; ++a) uu += ru[ib][a] * cu[a];
                        acc[cb][ib][r] = ms2 * P::exp_neg(T(T(0.5)) * d2) * uu;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            return;
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 2 * P::midx(r, g) + cb, j = col0 + c;
                T cx[4], cu[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) { cx[d] = cX[c][d]; cu[d] = cU[c][d]; }
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = irow + ib;
                    T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                    for (int d = 0; d < 4; ++d) { const T z = (rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
                    if (n > 4) {                                  // (wave-uniform, rare)
                        for (int d = 4; d < n; ++d) {
                            const T xi = i < N ? Xb[(size_t)i * n + d] : T(0.0);
                            const T z = (xi - cX[c][d]) / ell[(size_t)b * n + d];
                            d2 += z * z;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) uu += ru[ib][a] * cu[a];
                    T val = s2 * P::exp_neg(T(T(0.5)) * d2) * uu + (i == j ? rj[ib] : T(0.0));
                    val = (i >= N || j >= N) ? ((i == j) ? T(1.0) : T(0.0)) : val;
                    acc[cb][ib][r] = -val;                         // the accumulators carry -S' (see update)
                }
                // one column's inputs at a time: left alone the scheduler hoists the LDS reads of all eight columns
                // (64 values) to the top of the tile
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    // acc += L_{J, k0..k1} L_{I, k0..k1}'   (columns k0 <= k < k1 of the packed operator; software pipelined).  acc holds
    // -S': the MFMA has no negate modifier, and flipping an operand costs four VALU instructions per k-step against once
    // per tile at the consumer.  Addresses: inside block column K the packed columns are a fixed stride apart, so a load
    // is  buffer base + SCALAR offset (column block, k-step) + per-lane offset (lane group's column, row)  -- two
    // multiply-adds per fetch instead of the full column-offset polynomial per load (the VALU work of this loop was
    // a third of its time: the wave issues in order, address arithmetic does not hide behind its own MFMAs)
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(lop, 0, (int)(lop_elems<V>(Np) * ES), 0x00020000);
    int kc = 0;                       // bulk wave, A-operand cache: columns [0, kc) of block row J sit in LDS for this column
    auto update = [&](acc_t (&acc)[2][2], int I, int J, int k0, int k1) {
        constexpr int KS = sizeof(T) == 8 ? BCBF_RP_KS64 : BCBF_RP_KS32;
        const int col0 = J * NB, irow = I * NB + 2 * j16;
        if (k0 >= k1) return;
        T2 a_nxt[KS], b_nxt[KS];
        auto fetch = [&](int kk) {
            const int K = kk / NB, stride = Np - NB * (K + 1);     // (wave-uniform: scalar registers)
            // (lop_base of a block column's first column is NEGATIVE for K = 0 -- rows count from 32 (K + 1) -- and a
            //  scalar offset is unsigned: the row bias goes into the per-lane part, which it leaves non-negative)
            const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
            const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                const int so = (base + 4 * s_ * stride) * ES;
                if (ACACHE && kk + 4 * s_ < kc) {                      // (wave-uniform)
                    const __attribute__((address_space(3))) T* cp = acache((kk >> 2) + s_);
                    a_nxt[s_].x = cp[0]; a_nxt[s_].y = cp[1];
                } else a_nxt[s_] = P::bload2(rsL, va, so);
                b_nxt[s_] = P::bload2(rsL, vb, so);
            }
        };
        fetch(k0);
        for (int kk = k0; kk < k1; kk += 4 * KS) {
            T a_cur[KS][2], b_cur[KS][2];
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y;
                b_cur[s_][0] = b_nxt[s_].x; b_cur[s_][1] = b_nxt[s_].y;
            }
            if (kk + 4 * KS < k1) fetch(kk + 4 * KS);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = P::mfma(a_cur[s_][cb], b_cur[s_][ib], acc[cb][ib]);
        }
    };
    // the diagonal tile's stream: both operands are block row J (one load per k-step instead of two)
    auto update_diag = [&](acc_t (&acc)[2][2], int J, int k0, int k1) {
        constexpr int KS = sizeof(T) == 8 ? BCBF_RP_KS64 : BCBF_RP_KS32;
        const int col0 = J * NB;
        if (k0 >= k1) return;
        T2 a_nxt[KS];
        auto fetch = [&](int kk) {
            const int K = kk / NB, stride = Np - NB * (K + 1);
            const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
            const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) a_nxt[s_] = P::bload2(rsL, va, (base + 4 * s_ * stride) * ES);
        };
        fetch(k0);
        for (int kk = k0; kk < k1; kk += 4 * KS) {
            T a_cur[KS][2];
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) { a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y; }
            if (kk + 4 * KS < k1) fetch(kk + 4 * KS);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = P::mfma(a_cur[s_][cb], a_cur[s_][ib], acc[cb][ib]);
        }
    };
    int tix = 0; (void)tix;

    if (wave == 0) {
        // =============================== the CHAIN ===============================
#if BCBF_RP_CHAIN_PRIO
        __builtin_amdgcn_s_setprio(3);        // the chain is the launch's critical path: issue ahead of the bulk waves sharing the SIMD
#endif
        int fail = 0;
        if (EARLY) { load_rows(0); stage_issue(0); stage_commit(); }
        for (int J = 0; J < nblk; ++J) {
            const int col0 = J * NB;
            acc_t acc[2][2];
            RA_T(0);                                               // 0: column starts
            // the diagonal tile: values (fp32: its inputs were fetched under the previous column's inverse copy), the
            // updates over the columns whose panels of block row J exist ...
            if (!EARLY) { load_rows(J); stage_issue(J); stage_commit(); }
            values(acc, J, J);
            if (J > 1) {
                wait_for(&sp.second_done, J - 1);                  // L_{J,J-2}: the second tile of column J - 2 (and every tile left of it)
                update_diag(acc, J, 0, col0 - NB);
            }
            RA_T(0);                                               // 1: prepared
            if (J > 0) {                                           // ... and, as soon as wave 1 delivers it, over L_{J,J-1}
                wait_for(&sp.first_done, J);
                update_diag(acc, J, col0 - NB, col0);
            }
            RA_T(0);                                               // 2: diagonal tile up to date
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = -acc[cb][ib];
            const int bad = diag_factor_invert_acc<T>(BCBF_LDS_TILE(T, sp.d), acc, lane, false);
            RA_T(0);                                               // 3: factored + inverted
            if (bad != 0 && col0 + bad <= N) fail = col0 + bad;
            if (fail != 0) {
                if (lane == 0) __hip_atomic_store(&sp.fail, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                publish(&sp.inv_ready, 1 << 20);                   // lets wave 1 run out
                break;
            }
            // wave 1 takes inv(L_JJ) out of LDS (nothing writes xinv again before wave 1 has delivered this column's
            // first panel tile): the copy to global memory is off the chain
            publish(&sp.inv_ready, J + 1);
            if (EARLY && J + 1 < nblk) { load_rows(J + 1); stage_issue(J + 1); }
            {
                // 16-byte stores (13 instructions for both copies instead of 33 of 8 bytes: under load it is the number
                // of store instructions a wave pays for, not their bytes)
                const int bfull = lop_dfull_block(J, Np), bpack = lop_dinv_block(J, Np);
#pragma unroll
                for (int t = 0; t < NB * NB / 128; ++t) {
                    const int e = 2 * lane + 128 * t, c = e >> 5, r = e & 31;
                    T2 v; v.x = sp.d.xinv[r][c]; v.y = sp.d.xinv[r + 1][c];
                    *reinterpret_cast<T2*>(lop + bfull + e) = v;
                }
#pragma unroll
                for (int t = 0; t < (LOP_DB + 127) / 128; ++t) {
                    const int k = 2 * lane + 128 * t;                  // two consecutive entries of the packed triangle
                    if (k < LOP_DB) {
                        const unsigned rc = sp.pack_rc[k >> 1];        // (r0 | c0 << 8 | r1 << 16 | c1 << 24), 0xff.. = padding
                        const int r0 = rc & 0xff, c0 = (rc >> 8) & 0xff, r1 = (rc >> 16) & 0xff, c1 = rc >> 24;
                        T2 v;
                        v.x = r0 < NB ? sp.d.xinv[r0][c0] : T(0.0);
                        v.y = r1 < NB ? sp.d.xinv[r1][c1] : T(0.0);
                        *reinterpret_cast<T2*>(lop + bpack + k) = v;
                    }
                }
            }
            if (EARLY && J + 1 < nblk) stage_commit();
            RA_T(0);                                               // 4: inverse stored, published
        }
        if (lane == 0) info[b] = fail;
        return;
    }
    // =============================== the BULK ===============================
    if (EARLY && nblk > 1) { load_rows(1); stage_issue(0); stage_commit(); }
    for (int J = 0; J + 1 < nblk; ++J) {
        const int col0 = J * NB;
        T ainv[2][2][4];
        RA_T(1);                                                   // 0: column starts
        if (!EARLY) { load_rows(J + 1); stage_issue(J); stage_commit(); }
        if constexpr (ACACHE) {
            // block row J, columns [0, kc): the tiles L_{J,0}, L_{J,1} this wave stored in earlier columns (behind its own column
            // fences), in the per-lane layout `update` loads its A operand in -- one read for the nblk - J - 1 tiles of the column
            // (the first tile below, formed before the chain's inverse arrives, included).  Nothing else touches d.tile / arow1.
            kc = (J + 2 < nblk) ? min(col0, 4 * RP_ACACHE_GROUPS) : 0;            // (a column with one tile has nothing to share)
            for (int kk = 0; kk < kc; kk += 16) {
                T2 v[4];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const int k_ = kk + 4 * s_, K = k_ / NB, stride = Np - NB * (K + 1);
                    const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (k_ - K * NB) * stride;
                    v[s_] = P::bload2(rsL, (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES, base * ES);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) { __attribute__((address_space(3))) T* cp = acache((kk >> 2) + s_); cp[0] = v[s_].x; cp[1] = v[s_].y; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        for (int I = J + 1; I < nblk; ++I) {
            const int irow = I * NB + 2 * j16;
            acc_t acc[2][2];
            values(acc, I, J);
            load_rows(I + 1);
            update(acc, I, J, 0, col0);
            if (I == J + 1) {
                // (the first tile's S' was formed before this wait: it is the tile the chain continues on)
                RA_T(1);                                           // 1: first S' formed
                wait_for(&sp.inv_ready, J + 1);
                if (failed()) return;
                RA_T(1);                                           // 2: inv(L_JJ) arrived
                // A operands of the panel solve: output row c' = 16 cbp + j16, contraction index c = 2 midx(r, g) + cb
                // (the column an accumulator register of S' holds); inv(L_JJ) is lower triangular: c' < 16 meets c < 16
                // only (PANEL_R0)
#pragma unroll
                for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ainv[cbp][cb][r] = -sp.d.xinv[16 * cbp + j16][2 * P::midx(r, g) + cb];   // (acc = -S')
            }
            // L_IJ' = inv(L_JJ) S': the accumulator registers of S' are the B operands
            acc_t y[2][2];                                          // [ib][cbp]
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int cbp = 0; cbp < 2; ++cbp) {
                    acc_t yy = {0, 0, 0, 0};
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < (cbp == 0 ? P::PANEL_R0 : 4); ++r)
                            yy = P::mfma(ainv[cbp][cb][r], acc[cb][ib][r], yy);
                    y[ib][cbp] = yy;
                }
#pragma unroll
            for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T2 v; v.x = y[0][cbp][r]; v.y = y[1][cbp][r];       // rows irow, irow + 1 adjacent: one 16-byte store
                    *reinterpret_cast<T2*>(lop + lop_base<V>(col0 + 16 * cbp + P::midx(r, g), Np) + irow) = v;
                }
            if (I == J + 1) {
                publish(&sp.first_done, J + 1);
                RA_T(1);                                           // 3: the chain's panel tile delivered
            } else if (I == J + 2) {
                publish(&sp.second_done, J + 1);                   // (the chain's next diagonal tile but one starts on it)
            }
        }
        // one fence for the rest of the column: nobody reads these tiles before the next column (this wave: as update
        // operands from the next column on; the chain: for the diagonal tile after next)
        if (EARLY && J + 2 < nblk) { load_rows(J + 2); stage_issue(J + 1); }    // (the next column's inputs: in flight under the fence)
        publish(&sp.cols_done, J + 1);
        if (EARLY && J + 2 < nblk) stage_commit();
        RA_T(1);                                                   // 4: column complete
    }
}

template <typename T>
static int launch_refit_pair(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2, const T* jitter, T* Lop, T* UHB,
                              int* info, int Bt, int N, int Np, int n, int C, hipStream_t st) {
    if (Np / NB > RA_MAXBLK) return -1;
    if ((unsigned long long)lop_elems<16 / (int)sizeof(T)>(Np) * sizeof(T) >= (1ull << 31)) return -1;    // 32-bit byte offsets
    hipLaunchKernelGGL((refit_pair_kernel<T>), dim3(Bt), dim3(128), 0, st, X, UH, Bm, ell, s2, jitter, Lop, UHB, info, Bt, N, Np, n, C, g_refit_only_bad);
    return 0;
}
int launch_refit_pair64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                         const double* jitter, double* Lop, double* UHB, int* info, int Bt, int N, int Np, int n, int C,
                         hipStream_t st) {
    return launch_refit_pair<double>(X, UH, Bm, ell, s2, jitter, Lop, UHB, info, Bt, N, Np, n, C, st);
}
int launch_refit_pair32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                         const float* jitter, float* Lop, float* UHB, int* info, int Bt, int N, int Np, int n, int C,
                         hipStream_t st) {
    return launch_refit_pair<float>(X, UH, Bm, ell, s2, jitter, Lop, UHB, info, Bt, N, Np, n, C, st);
}

// ----------------------------------------------------------------------------------------------------------------
// FEW LARGE INSTANCES: A TEAM OF WAVES PER INSTANCE (round 3).  The two-wave kernel above with the bulk role on NW - 1
// waves: wave 0 is still the serial chain (diagonal tiles on the matrix cores, nothing else), bulk wave w takes every
// (NW - 1)-th tile of a block column.  A tile's inputs come from whichever waves made the tiles to its left, so the
// hand-off is per BLOCK ROW: pdone[I] counts the columns whose L_IJ is complete (it only ever grows: L_{I,J+1} is started
// by a wave that has seen L_IJ).  Bulk wave 0 (the tile the chain continues on) takes inv(L_JJ) out of LDS as in the
// two-wave kernel; the others may be columns behind and read the copy in global memory, behind a second counter.
// One workgroup of NW = 8 waves per instance (NW = 4: two instances per CU): two waves per SIMD, 256 registers each -- the form
// for a handful of systems of N >= 512 (the facade's fit / clear_cache refits of ONE model), where the workgroup form
// leaves three of its four waves waiting on the diagonal tile.
constexpr int RT_MAXBLK = 256;         // block columns the team form handles (N <= 8192)
template <typename T, int NW> struct RTShared {
    DiagTile<T> d;                            // wave 0's
    T colX[NW][NB][BCBF_MAX_STATE_DIM];       // [wave]: each wave stages the column block it is forming values for
    T colUH[NW][NB][BCBF_MAX_CTRL_DIM + 1];
    unsigned pack_rc[LOP_DB / 2];
    int pdone[RT_MAXBLK];                     // block row I: columns J whose L_IJ is complete
    int inv_ready;                            // diagonal tiles factored, inverse in LDS
    int inv_global;                           // ... and copied to global memory
    int fail;
};

// KIND: data kernel of the on-the-fly K_b values -- 0 = RBF (the reference's), 1 = Matern-5/2 (opt-in, bcbf_refit_matern52:
// only this form is compiled for it, the team of eight serves every batch size there)
template <typename T, int NW, bool FROM_DENSE, int KIND = 0>
__global__ void __launch_bounds__(64 * NW, 8 / NW)
refit_team_kernel(const T* __restrict__ X, const T* __restrict__ UH, const T* __restrict__ Bm,
                   const T* __restrict__ ell, const T* __restrict__ s2p, const T* __restrict__ jitter,
                   const T* __restrict__ Kdense, T* __restrict__ Lop, T* __restrict__ UHBout, T* __restrict__ Ldense,
                   int* __restrict__ info, int Bt, int N, int Np, int n, int C, const int* only_bad) {
    constexpr int V = Vec<T>::V, ES = (int)sizeof(T);
    using P = RW<T>;
    using acc_t = typename P::acc_t;
    using T2 = typename P::vec2;
    __shared__ __attribute__((aligned(16))) RTShared<T, NW> shm;
    __attribute__((address_space(3))) RTShared<T, NW>& sp = *(__attribute__((address_space(3))) RTShared<T, NW>*)&shm;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    if (only_bad != nullptr && only_bad[b] == 0) { if (threadIdx.x == 0) info[b] = 0; return; }     // bcbf_refit_retry: factored already
    const int j16 = lane & 15, g = lane >> 4;
    T* __restrict__ lop = Lop + (size_t)b * lop_elems<V>(Np);
    // (FROM_DENSE: K_b is given -- bcbf_potrf -- and the kernel values are read, not formed; no X / UH / UH B)
    const T* Xb = FROM_DENSE ? nullptr : X + (size_t)b * N * n;
    const T* UHb = FROM_DENSE ? nullptr : UH + (size_t)b * N * C;
    const T* Kb = FROM_DENSE ? Kdense + (size_t)b * N * N : nullptr;
    T* Ld = Ldense ? Ldense + (size_t)b * N * N : nullptr;          // dense L on request (zeros above the diagonal)
    const int nblk = Np / NB;
    {   // UH B rows (all waves), the hand-off words
        if constexpr (!FROM_DENSE) {
            T Bmr[(BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1)];
#pragma unroll
            for (int a = 0; a < (BCBF_MAX_CTRL_DIM + 1) * (BCBF_MAX_CTRL_DIM + 1); ++a) Bmr[a] = a < C * C ? Bm[(size_t)b * C * C + a] : T(0.0);
            for (int i = threadIdx.x; i < N; i += 64 * NW)
                for (int c = 0; c < C; ++c) {
                    T s = T(0.0);
                    for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
                    UHBout[((size_t)b * N + i) * C + c] = s;
                }
        }
        if (Ld)
            for (int e = threadIdx.x; e < N * N; e += 64 * NW) { const int i = e / N, j = e - i * N; if (j > i) Ld[e] = T(0.0); }
        if (threadIdx.x == 0) { sp.inv_ready = 0; sp.inv_global = 0; sp.fail = 0; }
        for (int i = threadIdx.x; i < RT_MAXBLK; i += 64 * NW) sp.pdone[i] = 0;
        rw_pack_table(sp.pack_rc, threadIdx.x, 64 * NW);
    }
    __threadfence_block();
    __syncthreads();
    // (a waiter also leaves when a pivot has failed: the wave it waits for may have left already)
    auto wait_for = [&](__attribute__((address_space(3))) int* word, int want) {
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want &&
               __hip_atomic_load(&sp.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto publish = [&](__attribute__((address_space(3))) int* word, int value) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto failed = [&]() { return __hip_atomic_load(&sp.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };

    // ---- what every wave does to a tile: kernel values, the update stream (each on its own registers and its own column
    //      block in LDS)
    const T* UHBb = FROM_DENSE ? nullptr : UHBout + (size_t)b * N * C;
    // (registers: the first four state components only -- no reference system has more; wider states take the slow
    //  path below, reading the rest from memory)
    T iell[4];
    const T s2 = FROM_DENSE ? T(0.0) : s2p[b];
#pragma unroll
    for (int d = 0; d < 4; ++d) iell[d] = (!FROM_DENSE && d < n) ? T(1.0) / ell[(size_t)b * n + d] : T(0.0);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(FROM_DENSE ? Lop : Xb), 0, FROM_DENSE ? 0 : N * n * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(FROM_DENSE ? Lop : UHBb), 0, FROM_DENSE ? 0 : N * C * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsJ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>((!FROM_DENSE && jitter) ? jitter + (size_t)b * N : Lop), 0, (!FROM_DENSE && jitter) ? N * ES : 0, 0x00020000);
    auto& cX = sp.colX[wave];
    auto& cU = sp.colUH[wave];
    // a column block's inputs, zero-filled to fixed widths (out-of-range offsets read as zero), through registers into LDS
    const __amdgpu_buffer_rsrc_t rsUH = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(FROM_DENSE ? Lop : UHb), 0, FROM_DENSE ? 0 : N * C * ES, 0x00020000);
    constexpr int SX = NB * BCBF_MAX_STATE_DIM / 64, SU = NB * (BCBF_MAX_CTRL_DIM + 1) / 64;
    T sx[SX], su[SU];
    auto stage_issue = [&](int J) {
        if (FROM_DENSE) return;
        const int col0 = J * NB;
#pragma unroll
        for (int t = 0; t < SX; ++t) {
            const int e = lane + 64 * t, c = e / BCBF_MAX_STATE_DIM, d = e % BCBF_MAX_STATE_DIM;
            sx[t] = P::bload(rsX, (col0 + c < N && d < n) ? ((col0 + c) * n + d) * ES : -ES);
        }
#pragma unroll
        for (int t = 0; t < SU; ++t) {
            const int e = lane + 64 * t, c = e / (BCBF_MAX_CTRL_DIM + 1), a = e % (BCBF_MAX_CTRL_DIM + 1);
            su[t] = P::bload(rsUH, (col0 + c < N && a < C) ? ((col0 + c) * C + a) * ES : -ES);
        }
    };
    auto stage_commit = [&]() {
        if (FROM_DENSE) return;
        __builtin_amdgcn_wave_barrier();                           // every lane is done with the previous column block
#pragma unroll
        for (int t = 0; t < SX; ++t) { const int e = lane + 64 * t; cX[e / BCBF_MAX_STATE_DIM][e % BCBF_MAX_STATE_DIM] = sx[t]; }
#pragma unroll
        for (int t = 0; t < SU; ++t) { const int e = lane + 64 * t; cU[e / (BCBF_MAX_CTRL_DIM + 1)][e % (BCBF_MAX_CTRL_DIM + 1)] = su[t]; }
        __builtin_amdgcn_wave_barrier();
    };
    T rx[2][4], ru[2][BCBF_MAX_CTRL_DIM + 1], rj[2];
    auto load_rows = [&](int I_) {
        if (FROM_DENSE) return;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int i = I_ * NB + 2 * j16 + ib;
            const bool in = I_ < nblk && i < N;
#pragma unroll
            for (int d = 0; d < 4; ++d) rx[ib][d] = P::bload(rsX, (in && d < n) ? (i * n + d) * ES : -ES);
#pragma unroll
            for (int c = 0; c < BCBF_MAX_CTRL_DIM + 1; ++c) ru[ib][c] = P::bload(rsU, (in && c < C) ? (i * C + c) * ES : -ES);
            rj[ib] = P::bload(rsJ, in ? i * ES : -ES);
        }
    };
    // acc[cb][ib][r] = K_b'(column col0 + 2 midx(r, g) + cb, row 32 I + 2 j16 + ib)   (rows of block row I in rx / ru / rj)
    auto values = [&](acc_t (&acc)[2][2], int I, int J) {
        const int col0 = J * NB, irow = I * NB + 2 * j16;
        if constexpr (FROM_DENSE) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const int i = irow + ib;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 2 * P::midx(r, g) + cb, j = col0 + c;
                        T val;
                        if (i >= N || j >= N) val = (i == j) ? T(1.0) : T(0.0);          // padding: identity
                        else val = (j <= i) ? Kb[(size_t)i * N + j] : Kb[(size_t)j * N + i];
                        acc[cb][ib][r] = -val;
                    }
            }
            return;
        }
        if (BCBF_RW_VALUES_FAST && KIND == 0 && I != J && (I + 1) * NB <= N && col0 + NB <= N && n <= 4) {
            // interior tile: the lean value pass of refit_wave_kernel (no jitter / padding selects; row inputs pinned, see there)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                asm volatile("" :: "v"(rj[ib]));
#pragma unroll
                for (int d = 0; d < 4; ++d) asm volatile("" :: "v"(rx[ib][d]));
#pragma unroll
                for (int a = 0; a < 4; ++a) asm volatile("" :: "v"(ru[ib][a]));
            }
            const T ms2 = -s2;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 2 * P::midx(r, g) + cb;
                    T cx[4], cu[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) { cx[d] = cX[c][d]; cu[d] = cU[c][d]; }
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                        for (int d = 0; d < 4; ++d) { const T z = (rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
#pragma unroll
                        for (int a = 0; a < 4; ++a) uu += ru[ib][a] * cu[a];
                        acc[cb][ib][r] = ms2 * P::exp_neg(T(T(0.5)) * d2) * uu;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            return;
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 2 * P::midx(r, g) + cb, j = col0 + c;
                T cx[4], cu[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) { cx[d] = cX[c][d]; cu[d] = cU[c][d]; }
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = irow + ib;
                    T d2 = T(0.0), uu = T(0.0);
#pragma unroll
                    for (int d = 0; d < 4; ++d) { const T z = (rx[ib][d] - cx[d]) * iell[d]; d2 += z * z; }
                    if (n > 4) {                                  // (wave-uniform, rare)
                        for (int d = 4; d < n; ++d) {
                            const T xi = i < N ? Xb[(size_t)i * n + d] : T(0.0);
                            const T z = (xi - cX[c][d]) / ell[(size_t)b * n + d];
                            d2 += z * z;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) uu += ru[ib][a] * cu[a];
                    T shape;
                    if constexpr (KIND != 0) {                     // Matern-5/2 (1), RBF x Matern-5/2 (2): bcbf_common.h
                        T dshape_;
                        kernel_shape(KIND, d2, [](T q_) { return P::exp_neg(-q_); }, shape, dshape_);
                    } else shape = P::exp_neg(T(T(0.5)) * d2);
                    T val = s2 * shape * uu + (i == j ? rj[ib] : T(0.0));
                    val = (i >= N || j >= N) ? ((i == j) ? T(1.0) : T(0.0)) : val;
                    acc[cb][ib][r] = -val;                         // the accumulators carry -S' (see update)
                }
                // one column's inputs at a time: left alone the scheduler hoists the LDS reads of all eight columns
                // (64 values) to the top of the tile
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    // acc += L_{J, k0..k1} L_{I, k0..k1}'   (columns k0 <= k < k1 of the packed operator; software pipelined).  acc holds
    // -S': the MFMA has no negate modifier, and flipping an operand costs four VALU instructions per k-step against once
    // per tile at the consumer.  Addresses: inside block column K the packed columns are a fixed stride apart, so a load
    // is  buffer base + SCALAR offset (column block, k-step) + per-lane offset (lane group's column, row)  -- two
    // multiply-adds per fetch instead of the full column-offset polynomial per load (the VALU work of this loop was
    // a third of its time: the wave issues in order, address arithmetic does not hide behind its own MFMAs)
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(lop, 0, (int)(lop_elems<V>(Np) * ES), 0x00020000);
    auto update = [&](acc_t (&acc)[2][2], int I, int J, int k0, int k1) {
        constexpr int KS = sizeof(T) == 8 ? BCBF_RP_KS64 : BCBF_RP_KS32;
        const int col0 = J * NB, irow = I * NB + 2 * j16;
        if (k0 >= k1) return;
        T2 a_nxt[KS], b_nxt[KS];
        auto fetch = [&](int kk) {
            const int K = kk / NB, stride = Np - NB * (K + 1);     // (wave-uniform: scalar registers)
            // (lop_base of a block column's first column is NEGATIVE for K = 0 -- rows count from 32 (K + 1) -- and a
            //  scalar offset is unsigned: the row bias goes into the per-lane part, which it leaves non-negative)
            const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
            const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES, vb = (g * stride + irow - NB * (K + 1)) * ES;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                const int so = (base + 4 * s_ * stride) * ES;
                a_nxt[s_] = P::bload2(rsL, va, so);
                b_nxt[s_] = P::bload2(rsL, vb, so);
            }
        };
        fetch(k0);
        for (int kk = k0; kk < k1; kk += 4 * KS) {
            T a_cur[KS][2], b_cur[KS][2];
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y;
                b_cur[s_][0] = b_nxt[s_].x; b_cur[s_][1] = b_nxt[s_].y;
            }
            if (kk + 4 * KS < k1) fetch(kk + 4 * KS);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = P::mfma(a_cur[s_][cb], b_cur[s_][ib], acc[cb][ib]);
        }
    };
    // the diagonal tile's stream: both operands are block row J (one load per k-step instead of two)
    auto update_diag = [&](acc_t (&acc)[2][2], int J, int k0, int k1) {
        constexpr int KS = sizeof(T) == 8 ? BCBF_RP_KS64 : BCBF_RP_KS32;
        const int col0 = J * NB;
        if (k0 >= k1) return;
        T2 a_nxt[KS];
        auto fetch = [&](int kk) {
            const int K = kk / NB, stride = Np - NB * (K + 1);
            const int base = lop_base<V>(K * NB, Np) + NB * (K + 1) + (kk - K * NB) * stride;
            const int va = (g * stride + col0 + 2 * j16 - NB * (K + 1)) * ES;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) a_nxt[s_] = P::bload2(rsL, va, (base + 4 * s_ * stride) * ES);
        };
        fetch(k0);
        for (int kk = k0; kk < k1; kk += 4 * KS) {
            T a_cur[KS][2];
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) { a_cur[s_][0] = a_nxt[s_].x; a_cur[s_][1] = a_nxt[s_].y; }
            if (kk + 4 * KS < k1) fetch(kk + 4 * KS);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = P::mfma(a_cur[s_][cb], a_cur[s_][ib], acc[cb][ib]);
        }
    };
    int tix = 0; (void)tix;

    if (wave == 0) {
        // =============================== the CHAIN ===============================
#if BCBF_RP_CHAIN_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        int fail = 0;
        for (int J = 0; J < nblk; ++J) {
            const int col0 = J * NB;
            acc_t acc[2][2];
            load_rows(J); stage_issue(J); stage_commit();
            values(acc, J, J);
            if (J > 1) {
                wait_for(&sp.pdone[J], J - 1);                     // L_{J,J-2} and every tile left of it
                if (failed()) break;
                update_diag(acc, J, 0, col0 - NB);
            }
            if (J > 0) {                                           // ... and, as soon as it is delivered, over L_{J,J-1}
                wait_for(&sp.pdone[J], J);
                if (failed()) break;
                update_diag(acc, J, col0 - NB, col0);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc[cb][ib] = -acc[cb][ib];
            const int bad = diag_factor_invert_acc<T>(BCBF_LDS_TILE(T, sp.d), acc, lane, Ld != nullptr);
            if (bad != 0 && col0 + bad <= N) fail = col0 + bad;
            if (Ld && lane < NB && col0 + lane < N) {
                for (int c = 0; c <= lane; ++c) if (col0 + c < N) Ld[(size_t)(col0 + lane) * N + col0 + c] = sp.d.tile[lane][c];
            }
            if (fail != 0) {
                if (lane == 0) __hip_atomic_store(&sp.fail, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;                                             // (every waiter polls the flag)
            }
            // bulk wave 0 takes inv(L_JJ) out of LDS (nothing writes xinv again before it has delivered this column's first
            // panel tile); the others read the copy in global memory once the second counter says it is complete
            publish(&sp.inv_ready, J + 1);
            {
                const int bfull = lop_dfull_block(J, Np), bpack = lop_dinv_block(J, Np);
#pragma unroll
                for (int t = 0; t < NB * NB / 128; ++t) {
                    const int e = 2 * lane + 128 * t, c = e >> 5, r = e & 31;
                    T2 v; v.x = sp.d.xinv[r][c]; v.y = sp.d.xinv[r + 1][c];
                    *reinterpret_cast<T2*>(lop + bfull + e) = v;
                }
#pragma unroll
                for (int t = 0; t < (LOP_DB + 127) / 128; ++t) {
                    const int k_ = 2 * lane + 128 * t;
                    if (k_ < LOP_DB) {
                        const unsigned rc = sp.pack_rc[k_ >> 1];
                        const int r0 = rc & 0xff, c0 = (rc >> 8) & 0xff, r1 = (rc >> 16) & 0xff, c1 = rc >> 24;
                        T2 v;
                        v.x = r0 < NB ? sp.d.xinv[r0][c0] : T(0.0);
                        v.y = r1 < NB ? sp.d.xinv[r1][c1] : T(0.0);
                        *reinterpret_cast<T2*>(lop + bpack + k_) = v;
                    }
                }
            }
            publish(&sp.inv_global, J + 1);
        }
        if (lane == 0) info[b] = fail;
        return;
    }
    // =============================== the BULK waves ===============================
    // bulk wave w takes the tiles I = J + 1 + w, J + 1 + w + (NW - 1), ... of block column J; per block row I a word in LDS
    // counts the columns whose L_IJ is complete (whoever made them)
    constexpr int NBW = NW - 1;
    const int w = wave - 1;
    for (int J = 0; J + 1 < nblk; ++J) {
        const int col0 = J * NB;
        if (J + 1 + w >= nblk) break;                              // (no tile of this or any later column for this wave)
        T ainv[2][2][4];
        load_rows(J + 1 + w); stage_issue(J); stage_commit();
        for (int I = J + 1 + w; I < nblk; I += NBW) {
            const int irow = I * NB + 2 * j16;
            acc_t acc[2][2];
            values(acc, I, J);
            load_rows(I + NBW);
            if (J > 0) {
                wait_for(&sp.pdone[I], J);                         // block rows I and J complete through column J - 1
                wait_for(&sp.pdone[J], J);
                if (failed()) return;
            }
            update(acc, I, J, 0, col0);
            if (I == J + 1 + w) {
                // (this wave's first tile of the column: its S' was formed before this wait)
                if (w == 0) {
                    wait_for(&sp.inv_ready, J + 1);
                    if (failed()) return;
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) ainv[cbp][cb][r] = -sp.d.xinv[16 * cbp + j16][2 * P::midx(r, g) + cb];   // (acc = -S')
                } else {
                    wait_for(&sp.inv_global, J + 1);
                    if (failed()) return;
                    const T* xf = lop + lop_dfull_block(J, Np);
#pragma unroll
                    for (int cbp = 0; cbp < 2; ++cbp)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) ainv[cbp][cb][r] = -xf[NB * (2 * P::midx(r, g) + cb) + 16 * cbp + j16];
                }
            }
            // L_IJ' = inv(L_JJ) S': the accumulator registers of -S' are the B operands, ainv = -inv(L_JJ)
#pragma unroll
            for (int cbp = 0; cbp < 2; ++cbp) {
                acc_t y[2];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    acc_t yy = {0, 0, 0, 0};
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < (cbp == 0 ? P::PANEL_R0 : 4); ++r)
                            yy = P::mfma(ainv[cbp][cb][r], acc[cb][ib][r], yy);
                    y[ib] = yy;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * cbp + P::midx(r, g);
                    T2 v; v.x = y[0][r]; v.y = y[1][r];
                    *reinterpret_cast<T2*>(lop + lop_base<V>(col0 + c, Np) + irow) = v;
                    if (Ld && col0 + c < N) {
                        if (irow < N) Ld[(size_t)irow * N + col0 + c] = v.x;
                        if (irow + 1 < N) Ld[(size_t)(irow + 1) * N + col0 + c] = v.y;
                    }
                }
            }
            publish(&sp.pdone[I], J + 1);
        }
    }
}

template <typename T>
static int launch_refit_team(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2, const T* jitter, const T* Kdense,
                             T* Lop, T* UHB, T* Ldense, int* info, int Bt, int N, int Np, int n, int C, int nw, hipStream_t st,
                             int kind = 0) {
    if (Np / NB > RT_MAXBLK) return -1;
    if (kind != 0) {
        if (Kdense || (unsigned long long)lop_elems<16 / (int)sizeof(T)>(Np) * sizeof(T) >= (1ull << 31)) return -1;
        if (kind == 1)
            hipLaunchKernelGGL((refit_team_kernel<T, 8, false, 1>), dim3(Bt), dim3(512), 0, st, X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, g_refit_only_bad);
        else if (kind == 2)
            hipLaunchKernelGGL((refit_team_kernel<T, 8, false, 2>), dim3(Bt), dim3(512), 0, st, X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, g_refit_only_bad);
        else return -1;
        return 0;
    }
    // the kernel addresses one instance's operator with 32-bit byte offsets (buffer resource size, scalar + lane offsets)
    if ((unsigned long long)lop_elems<16 / (int)sizeof(T)>(Np) * sizeof(T) >= (1ull << 31)) return -1;
    // four waves per instance (two workgroups per CU) when the batch needs more than one workgroup per CU but not more than
    // two: 512 x 256 fp64 0.259 (two waves per instance) / 0.330 (team of eight, two rounds) / 0.210 ms
    if (nw == 4 && !Kdense) {
        hipLaunchKernelGGL((refit_team_kernel<T, 4, false>), dim3(Bt), dim3(256), 0, st, X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, g_refit_only_bad);
        return 0;
    }
    if (Kdense)
        hipLaunchKernelGGL((refit_team_kernel<T, 8, true>), dim3(Bt), dim3(512), 0, st, X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, g_refit_only_bad);
    else
        hipLaunchKernelGGL((refit_team_kernel<T, 8, false>), dim3(Bt), dim3(512), 0, st, X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, g_refit_only_bad);
    return 0;
}
int launch_refit_team64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                        const double* jitter, const double* Kdense, double* Lop, double* UHB, double* Ldense, int* info, int Bt, int N,
                        int Np, int n, int C, int nw, hipStream_t st) {
    return launch_refit_team<double>(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, nw, st);
}
}  // namespace bcbf

// Fused K_b build + jittered Cholesky + packing with the OPT-IN Matern-5/2 data kernel (the reference has no Matern kernel;
// BASELINE.json's north_star names one): arguments of bcbf_refit.  One form -- a team of eight waves per instance.
#define BCBF_REFIT_MATERN(SUF, T, NAME, KINDV)                                                                              \
    extern "C" int NAME##SUF(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2, const T* jitter,                   \
                                             T* Lop, T* UHB, T* Ldense, int* info, int Bt, int N, int n, int m, void* stream) { \
        using namespace bcbf;                                                                                                \
        if (Bt <= 0) return BCBF_OK;                                                                                         \
        if (!X || !UH || !Bm || !ell || !s2 || !Lop || !UHB || !info || N < 1) return BCBF_EINVAL;                            \
        if (n < 1 || n > BCBF_MAX_STATE_DIM || m < 1 || m > BCBF_MAX_CTRL_DIM) return BCBF_EINVAL;                            \
        if (launch_refit_team<T>(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, Ldense, info, Bt, N, round_up(N, NB), n,      \
                                 m + 1, 8, (hipStream_t)stream, KINDV) != 0)                                                 \
            return BCBF_EINVAL;                                                                                              \
        return check_launch("refit_matern52");                                                                               \
    }
BCBF_REFIT_MATERN(f32, float, bcbf_refit_matern52_, 1)
BCBF_REFIT_MATERN(f64, double, bcbf_refit_matern52_, 1)
BCBF_REFIT_MATERN(f32, float, bcbf_refit_rbfm52_, 2)          // the product kernel RBF x Matern-5/2
BCBF_REFIT_MATERN(f64, double, bcbf_refit_rbfm52_, 2)
#undef BCBF_REFIT_MATERN
// bcbf_refit_retry (refit.hip) for a state of any data kernel: kernel_kind 0 = RBF (bcbf_refit_retry itself), 1 = Matern-5/2,
// 2 = RBF x Matern-5/2 -- only the instances with prev_info[b] != 0 are factored again, the others report 0
extern "C" int bcbf_refit_mfma_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                                   float*, float*, float*, int*, int, int, int, int, void*);
extern "C" int bcbf_refit_mfma_f64(const double*, const double*, const double*, const double*, const double*, const double*,
                                   const double*, double*, double*, double*, int*, int, int, int, int, void*);
#define BCBF_REFIT_RETRY_KIND(SUF, T)                                                                                        \
    extern "C" int bcbf_refit_retry_kind_##SUF(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2,               \
                                               const T* jitter, T* Lop, T* UHB, const int* prev_info, int* info, int Bt,      \
                                               int N, int n, int m, int kernel_kind, void* stream) {                         \
        if (Bt <= 0) return BCBF_OK;                                                                                         \
        if (!X || !UH || !Bm || !ell || !s2 || !UHB || !prev_info || prev_info == info) return BCBF_EINVAL;                  \
        if (kernel_kind < 0 || kernel_kind >= bcbf::BCBF_KINDS) return BCBF_EINVAL;                                          \
        bcbf::g_refit_only_bad = prev_info;                                                                                  \
        const int rc = kernel_kind == 0 ? bcbf_refit_mfma_##SUF(X, UH, Bm, ell, s2, jitter, nullptr, Lop, UHB, nullptr, info, \
                                                                Bt, N, n, m, stream)                                         \
                     : kernel_kind == 1 ? bcbf_refit_matern52_##SUF(X, UH, Bm, ell, s2, jitter, Lop, UHB, nullptr, info, Bt,  \
                                                                    N, n, m, stream)                                         \
                                        : bcbf_refit_rbfm52_##SUF(X, UH, Bm, ell, s2, jitter, Lop, UHB, nullptr, info, Bt, N, \
                                                                  n, m, stream);                                             \
        bcbf::g_refit_only_bad = nullptr;                                                                                    \
        return rc;                                                                                                           \
    }
BCBF_REFIT_RETRY_KIND(f32, float)
BCBF_REFIT_RETRY_KIND(f64, double)
#undef BCBF_REFIT_RETRY_KIND
namespace bcbf {
int launch_refit_team32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                        const float* jitter, const float* Kdense, float* Lop, float* UHB, float* Ldense, int* info, int Bt, int N,
                        int Np, int n, int C, int nw, hipStream_t st) {
    return launch_refit_team<float>(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, nw, st);
}

// Called by bcbf_refit_mfma_f64 / _f32 for batches.
template <typename T>
static int launch_refit_wave(const T* X, const T* UH, const T* Bm, const T* ell, const T* s2, const T* jitter,
                             const T* Kdense, T* Lop, T* UHB, T* Ldense, int* info, int Bt, int N, int Np, int n, int C,
                             hipStream_t st) {
    const dim3 grid((Bt + RW_WPB - 1) / RW_WPB), block(64 * RW_WPB);
    int dev_ = 0, cus = 256;
    (void)hipGetDevice(&dev_);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_);
    bool two = sizeof(T) == 4 && Bt >= 8 * cus && Np < 1024;   // more than one instance per SIMD (N >= 1024: one wave with all 512
                                                               // registers is faster even then, 4096 x 1024: 20.7 against 21.6 ms)
    if (const char* e = getenv("BCBF_RW32_OCC")) two = e[0] == '2';      // (development: force the allocation)
#define BCBF_RW_LAUNCH(OCC_, ...)                                                                                         \
    do {                                                                                                                  \
        if (Kdense)                                                                                                       \
            hipLaunchKernelGGL((refit_wave_kernel<T, true, OCC_>), grid, block, 0, st, nullptr, nullptr, nullptr, nullptr,  \
                               nullptr, nullptr, Kdense, Lop, nullptr, Ldense, info, Bt, N, Np, 0, 0, g_refit_only_bad);                    \
        else                                                                                                              \
            hipLaunchKernelGGL((refit_wave_kernel<T, false, OCC_, ##__VA_ARGS__>), grid, block, 0, st, X, UH, Bm, ell, s2, jitter,        \
                               nullptr, Lop, UHB, Ldense, info, Bt, N, Np, n, C, g_refit_only_bad);                                         \
    } while (0)
    if constexpr (sizeof(T) == 4) {
        // super-panels from N = 512 on (BCBF_RW32_SUPER; never for a given dense K_b or a dense output: those are the
        // few-system paths)
        bool sup = BCBF_RW32_SUPER && BCBF_RW32_PAIRS && Np >= 512 && !Kdense && !Ldense;
        if (const char* e = getenv("BCBF_RW32_SUPER_FORCE")) sup = e[0] == '1' && !Kdense;      // (development)
        if (sup) { if (two) BCBF_RW_LAUNCH(2, true); else BCBF_RW_LAUNCH(1, true); }
        else { if (two) BCBF_RW_LAUNCH(2); else BCBF_RW_LAUNCH(1); }
    } else {
        // fp64: from N = 1024 (1024 x 1024 14.5 -> 12.7 ms; at N = 512 the four fp64 tiles -- 128 accumulator registers -- cost
        // more than the halved panel reads save: 4096 x 512 7.35 -> 9.1 ms)
        bool sup = BCBF_RW64_SUPER && Np >= 1024 && !Kdense && !Ldense;
        if (const char* e = getenv("BCBF_RW64_SUPER_FORCE")) sup = e[0] == '1' && !Kdense;      // (development)
        if (sup) BCBF_RW_LAUNCH(BCBF_RW64_OCC, true); else BCBF_RW_LAUNCH(BCBF_RW64_OCC);
    }
#undef BCBF_RW_LAUNCH
    return 0;
}
int launch_refit_wave64(const double* X, const double* UH, const double* Bm, const double* ell, const double* s2,
                        const double* jitter, const double* Kdense, double* Lop, double* UHB, double* Ldense, int* info,
                        int Bt, int N, int Np, int n, int C, hipStream_t st) {
    return launch_refit_wave<double>(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, st);
}
int launch_refit_wave32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                        const float* jitter, const float* Kdense, float* Lop, float* UHB, float* Ldense, int* info,
                        int Bt, int N, int Np, int n, int C, hipStream_t st) {
    return launch_refit_wave<float>(X, UH, Bm, ell, s2, jitter, Kdense, Lop, UHB, Ldense, info, Bt, N, Np, n, C, st);
}

}  // namespace bcbf
