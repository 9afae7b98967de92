// K1 + K2, fp32, LARGE BATCHES of N = 384 .. 512: ONE WORKGROUP OF FOUR WAVES PER INSTANCE, the factor's left part staged
// through LDS in column slabs.
//
// Why another form.  The one-wave-per-instance kernel (refit_wave64.hip) feeds every MFMA of the left-looking update from
// global memory: a 2 x 2 block of tiles reads four 32-row operand tiles per 32 columns, 2.7 MB per instance for a 0.5 MB
// factor, in 128-byte pieces (a tile's column), with ~2000 instances (1 GB) in flight: no cache holds that, the launch
// runs at the speed of those re-reads (C3: 11 GB in 3.07 ms, MFMA pipe 38 % busy).  Here the four waves of a
// workgroup own the row tiles of ONE instance; for a 64-column super-panel (block columns J0, J0 + 1) the columns to its
// left are copied into LDS once, 16 at a time, as whole column runs (rows 32 J0 .. Np - 1: up to 1.75 KB contiguous), and
// EVERY operand of the update comes from there: the two panel rows (A operands, shared by all waves) are simply the first
// 64 rows of the slab, the B operands are a wave's own rows.  Global reads per instance: each tile once per super-panel
// that needs it -- half the bytes of the 2 x 2 form -- and two instances in flight per CU (512 x 0.5 MB: the Infinity
// Cache holds them) instead of eight.
//
// Row tile t of the super-panel (block row J0 + t) belongs to wave t mod 4, slot t / 4; a slot holds the tile's two
// accumulators S'(J0), S'(J0 + 1) transposed as in refit_mfma.hip (lane = row, registers = columns), so that the panel
// solve takes them as B operands, and column J0 + 1 takes what it owes column J0 from the registers of that solve (the
// A operand L_{J0+1,J0}, solved by wave 1, crosses through LDS).  The two diagonal tiles are factored and inverted by
// their owners (diag_tile64.h, on the matrix cores) while the other waves wait at a barrier -- that chain is what the
// second workgroup of the CU hides.  Outputs, layout and info convention: bcbf_common.h, as every other form.

#include <type_traits>
#include "bcbf_common.h"
#include "diag_tile64.h"

namespace bcbf {

using f32x16s = __attribute__((__vector_size__(16 * sizeof(float)))) float;
using f32x4s = __attribute__((__vector_size__(4 * sizeof(float)))) float;
#define RS_AS3 __attribute__((address_space(3)))

constexpr int RS_NW = 4;                 // waves per workgroup
constexpr int RS_R = 4;                  // row tiles a wave holds (N <= 32 * RS_NW * RS_R = 512)
constexpr int RS_MAXNT = RS_NW * RS_R - 2;   // row tiles under the first super-panel that has anything to its left
constexpr int RS_NBUF = 4;               // slabs in LDS: one being read, three in flight
constexpr int RS_NPW = 4;                // 16-byte pieces of a slab per lane, at most (14 wave-instructions over four waves)
constexpr int RS_SLAB = 8 * NB * RS_MAXNT;   // floats per slab buffer: 8 columns x 448 rows (16 x <= 192, 32 x 64 for the later super-panels)

struct RSShared {
    DiagTile<float> d;                   // the diagonal tile's working set; d.xinv = inv(L_JJ) of the tile factored last
    float l10[16][64];                   // L_{J0+1,J0}' in the register layout of its solve ([register][lane])
    float colX[2 * NB][4];               // inputs of the super-panel's 64 columns (zero-filled to four components)
    float colUH[2 * NB][4];
    int fail;
    int pad_[3];
    float slab[RS_NBUF][RS_SLAB];
};

// -DBCBF_RS_PROF (development): wall-clock ticks (100 MHz) per phase, thread 0 of workgroup 0, summed over the super-panels
#ifdef BCBF_RS_PROF
__device__ long long rs_prof[16];
#define RS_T(k) do { if (b == 0 && tid == 0) { const long long n_ = wall_clock64(); rs_prof[k] += n_ - t_; t_ = n_; } } while (0)
#else
#define RS_T(k) do {} while (0)
#endif
__device__ inline int rs_acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ inline float rs_bload(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

__global__ void __launch_bounds__(64 * RS_NW, 2)
refit_slab_kernel_f32(const float* __restrict__ X, const float* __restrict__ UH, const float* __restrict__ Bm,
                      const float* __restrict__ ell, const float* __restrict__ s2p, const float* __restrict__ jitter,
                      float* __restrict__ Lop, float* __restrict__ UHBout, int* __restrict__ info, int N, int Np, int n, int C,
                      const int* only_bad) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    RS_AS3 RSShared& sh = *(RS_AS3 RSShared*)rs_smem;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (only_bad != nullptr && only_bad[b] == 0) { if (tid == 0) info[b] = 0; return; }     // bcbf_refit_retry: factored already
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
    float* __restrict__ lop = Lop + (size_t)b * lop_elems<4>(Np);
    const float* Xb = X + (size_t)b * N * n;
    const float* UHb = UH + (size_t)b * N * C;
    float iell[4], Bmr[16];
    const float s2 = s2p[b];
#pragma unroll
    for (int d = 0; d < 4; ++d) iell[d] = d < n ? 1.f / ell[(size_t)b * n + d] : 0.f;
#pragma unroll
    for (int a = 0; a < 16; ++a) Bmr[a] = a < C * C ? Bm[(size_t)b * C * C + a] : 0.f;
    for (int i = tid; i < N; i += 64 * RS_NW)
        for (int c = 0; c < C; ++c) {
            float s = 0.f;
            for (int a = 0; a < C; ++a) s += UHb[(size_t)i * C + a] * Bmr[a * C + c];
            UHBout[((size_t)b * N + i) * C + c] = s;
        }
    if (tid == 0) sh.fail = 0;
    __threadfence_block();
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xb), 0, N * n * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsUH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(UHb), 0, N * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(UHBout + (size_t)b * N * C, 0, N * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsJ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(jitter ? jitter + (size_t)b * N : X), 0, jitter ? N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(lop, 0, (unsigned)(lop_elems<4>(Np) * 4), 0x00020000);

    const int nblk = Np / NB;
#ifdef BCBF_RS_PROF
    long long t_ = wall_clock64();
#endif
    for (int J0 = 0; J0 < nblk; J0 += 2) {
        const int nt = nblk - J0, col0 = J0 * NB, pitch = nt * NB;
        const int nlive = (nt - wave + 3) >> 2;               // this wave's slots q < nlive hold row tiles t = wave + 4 q < nt
        // ---- inputs of the 64 columns: one value per thread and array
        {
            const int c = tid >> 2, d = tid & 3;
            sh.colX[c][d] = rs_bload(rsX, (col0 + c < N && d < n) ? ((col0 + c) * n + d) * 4 : -4);
            sh.colUH[c][d] = rs_bload(rsUH, (col0 + c < N && d < C) ? ((col0 + c) * C + d) * 4 : -4);
        }
        // ---- slab copy, global -> LDS without registers (buffer_load ... lds, 16 bytes per lane, a wave-instruction fills 1 KB of LDS).
        //      A slab is KCP = 8 nu columns x rows 32 J0 .. Np - 1 (nu = 1, 2, 4 as the rows get fewer: 8 - 14 KB); its LDS image is
        //      column after column, so piece p (16 bytes) is rows 4 rq .. of column p / (8 nt): per-lane SOURCE offsets, formed once per
        //      super-panel; the nu nt wave-instructions of a slab are dealt to the four waves in turn
        int lhv = lh;
        asm volatile("" : "+v"(lhv));                          // (addresses derived from the lane's half are formed per super-panel, not hoisted and spilled)
        const int nu = nt >= 8 ? 1 : nt >= 4 ? 2 : 4, kcp = 8 * nu;
        const int q8 = 8 * nt, ninst = nu * nt, mine = (ninst - wave + RS_NW - 1) / RS_NW;   // pieces per column; this wave's instructions per slab
        const float invq8 = 1.0f / (float)q8;
        int pcol[RS_NPW], prq[RS_NPW];
#pragma unroll
        for (int j = 0; j < RS_NPW; ++j) {
            const int pp = (j * RS_NW + wave) * 64 + lane;
            pcol[j] = (int)(((float)pp + 0.5f) * invq8);
            prq[j] = 16 * (pp - pcol[j] * q8);
        }
        auto issue = [&](int ch) {
            const int kk0 = ch * kcp, K = kk0 >> 5, cb0 = kk0 & 31;
            const int tb = lop_base<4>(K * NB, Np), cs = Np - NB * (K + 1);
            const int so = (tb + cb0 * cs + col0) * 4, cs4 = 4 * cs;       // wave-uniform
            RS_AS3 float* dst = &sh.slab[ch % RS_NBUF][wave * 256];
#pragma unroll
            for (int j = 0; j < RS_NPW; ++j)
                if (j < mine)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (RS_AS3 void*)(dst + j * RS_NW * 256), 16, pcol[j] * cs4 + prq[j], so, 0, 0);
        };
        // all but this wave's youngest `keep` loads have landed
        auto wait_but = [&](int keep) {
            switch (keep) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            }
        };
        const int nch = col0 / kcp;                            // slabs left of the super-panel (0, or at least 2)
#pragma unroll
        for (int a = 0; a < RS_NBUF - 1; ++a) if (a < nch) issue(a);
        __syncthreads();                                       // (A) column inputs visible
        RS_T(0);

        // ---- a diagonal tile: factor, invert (d.xinv), write both copies of the inverse
        auto factor = [&](const f32x16s& s, int J) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sh.d.tile[4 * lhv + (r & 3) + 8 * (r >> 2)][li] = s[r];
            __builtin_amdgcn_wave_barrier();
            const int j16 = lane & 15, g = lane >> 4;
            f32x4s S[2][2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[cb][ib][r] = sh.d.tile[2 * (4 * g + r) + cb][2 * j16 + ib];
            __builtin_amdgcn_wave_barrier();
            const int bad = diag_factor_invert_acc<float>(BCBF_LDS_TILE(float, sh.d), S, lane, false);
            if (lane == 0 && bad != 0 && J * NB + bad <= N) sh.fail = J * NB + bad;
            const int bfull = lop_dfull_block(J, Np), bpack = lop_dinv_block(J, Np);
#pragma unroll
            for (int t = 0; t < NB * NB / 64; ++t) {
                const int e = lane + 64 * t, c = e >> 5, r = e & 31;
                const float xv = sh.d.xinv[r][c];
                lop[bfull + e] = xv;
                if (r >= c) lop[bpack + lop_dinv_col(c) + r] = xv;
            }
            if (lane < LOP_DB - 528) lop[bpack + 528 + lane] = 0.f;      // the block's padding
        };
        auto store_tile = [&](const f32x16s& y, int I, int J) {
            const int tb = lop_base<4>(J * NB, Np), cs = Np - NB * (J + 1);            // column c of block column J: tb + c cs (+ row)
            // (lop_base is negative for the first columns: the scalar offset takes the column's first stored row along, the lane's offset
            //  counts rows from there -- a negative scalar offset would wrap)
            const int vo = (4 * lhv * cs + (I - J - 1) * NB + li) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[r]), rsL, vo, (tb + NB * (J + 1) + ((r & 3) + 8 * (r >> 2)) * cs) * 4, 0);
        };

        // ---- the super-panel with this wave's slot count as a compile-time constant (the accumulators of the slots it does not have
        //      do not exist: written with run-time slot tests the register allocator shuffled and spilled whole accumulators between the
        //      cases).  Returns false after a failed pivot.
        auto panel = [&](auto nlc) -> bool {
            constexpr int NL = decltype(nlc)::value;
            f32x16s acc[NL > 0 ? NL : 1][2];
            // ---- initial values: acc[q][cj][r] = K_b(i = 32 I + li, j = col0 + 32 cj + rho(r, lh))
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int t = wave + 4 * q, I = J0 + t, i = I * NB + li;
                // (a row's components beyond n / C are whatever follows it in memory: they meet the zero-filled column inputs and zero
                //  inverse length scales; a row of the padding reads out of range = zeros)
                const int ox = i < N ? i * n * 4 : 0x7fffff00, ou = i < N ? i * C * 4 : 0x7fffff00;
                float rx[4], ru[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) rx[d] = rs_bload(rsX, ox + 4 * d);
#pragma unroll
                for (int c = 0; c < 4; ++c) ru[c] = rs_bload(rsU, ou + 4 * c);
                const float rj = rs_bload(rsJ, i < N ? i * 4 : 0x7fffff00);
                const RS_AS3 float* cxp = &sh.colX[4 * lhv][0];
                const RS_AS3 float* cup = &sh.colUH[4 * lhv][0];
                const int dj = i - col0 - 4 * lhv, jl = col0 + 4 * lhv;       // i == j  <=>  dj == the register's column;  j = jl + that column
#pragma unroll
                for (int cj = 0; cj < 2; ++cj) {
                    if (q == 0 && cj == 1 && t == 0) {             // above the diagonal: never read
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][cj][r] = 0.f;
                        continue;
                    }
                    const bool fast = I > J0 + cj && (I + 1) * NB <= N;        // off the diagonal and clear of the padding (wave-uniform)
                    auto tile_values = [&](auto fastc) {
                        constexpr bool FAST = decltype(fastc)::value;
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int cc = 32 * cj + (r & 3) + 8 * (r >> 2);       // column of the super-panel, less 4 * (lane's half)
                            const f32x4s cx = *(const RS_AS3 f32x4s*)(cxp + 4 * cc), cu = *(const RS_AS3 f32x4s*)(cup + 4 * cc);
                            float d2 = 0.f, uu = 0.f;
#pragma unroll
                            for (int d = 0; d < 4; ++d) { const float z = (rx[d] - cx[d]) * iell[d]; d2 += z * z; }
#pragma unroll
                            for (int a = 0; a < 4; ++a) uu += ru[a] * cu[a];
                            float val = s2 * __expf(-0.5f * d2) * uu;
                            if (!FAST) {                                           // (selects, no branches)
                                const bool dg = dj == cc;
                                val = dg ? val + rj : val;
                                val = (i >= N || jl + cc >= N) ? (dg ? 1.f : 0.f) : val;      // padding: identity
                            }
                            v[r] = val;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][cj][r] = v[r];
                    };
                    if (fast) tile_values(std::true_type{}); else tile_values(std::false_type{});
                }
            }
            RS_T(1);
            // ---- S' -= L_J L_I'  over the columns left of the super-panel, slab by slab out of LDS
            for (int ch = 0; ch < nch; ++ch) {
                // slab ch has landed (this wave's pieces: the counted wait; everybody's: the barrier, which also says that slab ch - 1's
                // buffer has been read by all) -- the raw barrier: __syncthreads() would drain the loads in flight
                const int ahead = nch - 1 - ch;
                wait_but((ahead < RS_NBUF - 2 ? ahead : RS_NBUF - 2) * mine);
                __builtin_amdgcn_s_barrier();
                if (ch + RS_NBUF - 1 < nch) issue(ch + RS_NBUF - 1);
                if (NL > 0) {
                    const RS_AS3 float* sl = &sh.slab[ch % RS_NBUF][lh * pitch + li];
                    for (int u = 0; u < nu; ++u, sl += 8 * pitch) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const float a0 = -sl[2 * s * pitch], a1 = -sl[2 * s * pitch + NB];
#pragma unroll
                            for (int q = 0; q < NL; ++q) {
                                const float bq = sl[2 * s * pitch + NB * (wave + 4 * q)];
                                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq, acc[q][0], 0, 0, 0);
                                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq, acc[q][1], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            // ---- panel solve  L_IJ' = inv(L_JJ) S'  (accumulator registers of S' are the B operands)
            float av[16];
            auto load_ainv = [&]() {
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = sh.d.xinv[li][4 * lhv + (r & 3) + 8 * (r >> 2)];
            };
            auto solve = [&](const f32x16s& s) {
                f32x16s y = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) y = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], s[r], y, 0, 0, 0);
                return y;
            };
            RS_T(2);
            // ================= block column J0
            if (NL > 0 && wave == 0) factor(acc[0][0], J0);
            RS_T(3);
            __syncthreads();                                       // (B)
            if (sh.fail != 0) return false;
            if (NL > 0) load_ainv();
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int t = wave + 4 * q;
                if (q == 0 && t == 0) continue;
                const f32x16s y = solve(acc[q][0]);
                store_tile(y, J0 + t, J0);
                acc[q][0] = y;
                if (q == 0 && t == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sh.l10[r][lane] = y[r];
                }
            }
            RS_T(4);
            __syncthreads();                                       // (C) L_{J0+1,J0} visible
            // ================= what block column J0 + 1 owes J0, from the registers of the solve
            if (NL > 0) {
                float a2[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) a2[r] = -sh.l10[r][lane];
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    if (q == 0 && wave == 0) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], acc[q][0][r], acc[q][1], 0, 0, 0);
                }
            }
            // ================= block column J0 + 1
            RS_T(5);
            if (NL > 0 && wave == 1) factor(acc[0][1], J0 + 1);
            __syncthreads();                                       // (D)
            RS_T(6);
            if (sh.fail != 0) return false;
            if (NL > 0) load_ainv();
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int t = wave + 4 * q;
                if (q == 0 && t < 2) continue;
                const f32x16s y = solve(acc[q][1]);
                store_tile(y, J0 + t, J0 + 1);
            }
            RS_T(7);
            __threadfence_block();
            __syncthreads();                                       // (E) the super-panel's columns are readable
            RS_T(8);
            return true;
        };
        bool good;
        switch (nlive) {
            case 0: good = panel(std::integral_constant<int, 0>{}); break;
            case 1: good = panel(std::integral_constant<int, 1>{}); break;
            case 2: good = panel(std::integral_constant<int, 2>{}); break;
            case 3: good = panel(std::integral_constant<int, 3>{}); break;
            default: good = panel(std::integral_constant<int, 4>{}); break;
        }
        if (!good) break;
    }
    if (tid == 0) info[b] = sh.fail;
}

// 0 = launched; 1 = a size this form does not address
int launch_refit_slab32(const float* X, const float* UH, const float* Bm, const float* ell, const float* s2,
                        const float* jitter, float* Lop, float* UHB, int* info, int Bt, int N, int Np, int n, int C,
                        hipStream_t st) {
    if (Np % (2 * NB) != 0 || Np < 4 * NB || Np > NB * RS_NW * RS_R || n > 4 || C > 4) return 1;
    static const bool attr_ = [] {
        (void)hipFuncSetAttribute((const void*)refit_slab_kernel_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RSShared));
        return true;
    }();
    (void)attr_;
    hipLaunchKernelGGL(refit_slab_kernel_f32, dim3(Bt), dim3(64 * RS_NW), sizeof(RSShared), st, X, UH, Bm, ell, s2, jitter, Lop,
                       UHB, info, N, Np, n, C, g_refit_only_bad);
    return 0;
}

}  // namespace bcbf
#ifdef BCBF_RS_PROF
extern "C" __attribute__((visibility("default"))) int bcbf_debug_rs_prof(long long* out, int reset) {
    long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bcbf::rs_prof), sizeof(z)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(bcbf::rs_prof), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif
