// K_b^-1 = Linv' Linv from the dense triangular inverse Linv = L^-1 (bcbf_trtri) -- the fit path's K_b^-1
// (ControlAffineRegressor.fit, control_affine_model.py:268-335: gpytorch forms inv-quad / logdet terms by its own solves).
// One wave per 32 x 32 output tile (I, J), J <= I:  C[i][j] = sum_{k >= 32 I} Linv[k][32 I + i] Linv[k][32 J + j]
// (rows above tile row I are zero in column block I: Linv is lower triangular).  Both MFMA operands are read straight from
// global memory with the tile column on the lane -- 32 consecutive elements of one row of Linv, one 128-byte (fp32) or
// 256-byte (fp64) segment per half wave, no LDS, no transposes; the factor is cache resident (1 MiB at N = 512 in fp32).
// The mirror tile (J, I) is written from the same accumulators.
#include <type_traits>
#include "bcbf_common.h"
#include <stdlib.h>

namespace bcbf {

constexpr int SY_DEPTH = 8;        // k-steps whose operand loads are issued together

__global__ void __launch_bounds__(64)
syrk_lt_kernel_f32(const float* __restrict__ Linv, float* __restrict__ Kinv, int N, int nb) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    // tile index t -> (I, J), J <= I
    int t = blockIdx.x, I = 0;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const float* A = Linv + (size_t)blockIdx.y * N * N;
    float* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 31, kk = lane >> 5;          // v_mfma_f32_32x32x2: lane = (column, k of 2)
    const int ci = I * 32 + col, cj = J * 32 + col;
    const bool vi = ci < N, vj = cj < N;
    f32x16 acc = {0};
    for (int k0 = I * 32; k0 < N; k0 += 2 * SY_DEPTH) {
        float a[SY_DEPTH], b[SY_DEPTH];
#pragma unroll
        for (int s = 0; s < SY_DEPTH; ++s) {
            const int k = k0 + 2 * s + kk;
            a[s] = (vi && k < N) ? A[(size_t)k * N + ci] : 0.0f;
            b[s] = (vj && k < N) ? A[(size_t)k * N + cj] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < SY_DEPTH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    // accumulator: lane (column j = lane % 32, group g = lane / 32), register r -> row i = 8 (r / 4) + 4 g + r % 4
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r >> 2) + 4 * kk + (r & 3);
        const int gi = I * 32 + i;
        if (gi < N && vj) {
            C[(size_t)gi * N + cj] = acc[r];
            if (I != J) C[(size_t)cj * N + gi] = acc[r];
        }
    }
}

__global__ void __launch_bounds__(64)
syrk_lt_kernel_f64(const double* __restrict__ Linv, double* __restrict__ Kinv, int N, int nb) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    int t = blockIdx.x, I = 0;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const double* A = Linv + (size_t)blockIdx.y * N * N;
    double* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 15, kk = lane >> 4;          // v_mfma_f64_16x16x4: lane = (column, k of 4)
    f64x4 acc[2][2] = {};
    for (int k0 = I * 32; k0 < N; k0 += 4 * (SY_DEPTH / 2)) {
        double a[SY_DEPTH / 2][2], b[SY_DEPTH / 2][2];
#pragma unroll
        for (int s = 0; s < SY_DEPTH / 2; ++s) {
            const int k = k0 + 4 * s + kk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ci = I * 32 + 16 * h + col, cj = J * 32 + 16 * h + col;
                a[s][h] = (ci < N && k < N) ? A[(size_t)k * N + ci] : 0.0;
                b[s][h] = (cj < N && k < N) ? A[(size_t)k * N + cj] : 0.0;
            }
        }
#pragma unroll
        for (int s = 0; s < SY_DEPTH / 2; ++s)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int hj = 0; hj < 2; ++hj)
                    acc[hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][hi], b[s][hj], acc[hi][hj], 0, 0, 0);
    }
    // accumulator of one 16 x 16 tile: lane (column j = lane % 16, group g = lane / 16), register r -> row i = g + 4 r
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int hj = 0; hj < 2; ++hj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = I * 32 + 16 * hi + kk + 4 * r, gj = J * 32 + 16 * hj + col;
                if (gi < N && gj < N) {
                    C[(size_t)gi * N + gj] = acc[hi][hj][r];
                    if (I != J) C[(size_t)gj * N + gi] = acc[hi][hj][r];
                }
            }
}

// Batches (round 6): 2 x 2 output tiles per wave.  A wave that owns tiles (I, I+1) x (J, J+1) reads FOUR column blocks of Linv for FOUR tiles where the
// one-tile form reads two per tile -- half the operand loads, and eight (fp64: sixteen) independent accumulators per MFMA stream.  Super-tile (SI, SJ), SJ <= SI;
// the contraction starts at row 64 SI (the rows above are zero in column blocks 2 SI and 2 SI + 1: Linv is lower triangular and stored with its zeros).  In a
// diagonal super-tile the tile above the diagonal is the mirror of the one below it: not computed, written from the same accumulators like every mirror.
__global__ void __launch_bounds__(64)
syrk_lt2_kernel_f64(const double* __restrict__ Linv, double* __restrict__ Kinv, int N, int nb2) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    int t = blockIdx.x, SI = 0;
    while ((SI + 1) * (SI + 2) / 2 <= t) ++SI;
    const int SJ = t - SI * (SI + 1) / 2;
    const double* A = Linv + (size_t)blockIdx.y * N * N;
    double* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 15, kk = lane >> 4;
    f64x4 acc[2][2][2][2] = {};                                  // [ti][tj][hi][hj]
    constexpr int D = 4;                                         // k-steps (of 4 rows) whose operand loads are issued together
    for (int k0 = SI * 64; k0 < N; k0 += 4 * D) {
        double a[D][2][2], b[D][2][2];                           // [step][tile][half]
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const int k = k0 + 4 * q + kk;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ci = (2 * SI + ti) * 32 + 16 * h + col, cj = (2 * SJ + ti) * 32 + 16 * h + col;
                    a[q][ti][h] = (ci < N && k < N) ? A[(size_t)k * N + ci] : 0.0;
                    b[q][ti][h] = (cj < N && k < N) ? A[(size_t)k * N + cj] : 0.0;
                }
        }
#pragma unroll
        for (int q = 0; q < D; ++q)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    if (SI == SJ && tj > ti) continue;           // (the mirror of tile (1, 0))
#pragma unroll
                    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                        for (int hj = 0; hj < 2; ++hj)
                            acc[ti][tj][hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][ti][hi], b[q][tj][hj], acc[ti][tj][hi][hj], 0, 0, 0);
                }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            if (SI == SJ && tj > ti) continue;
            const int I = 2 * SI + ti, J = 2 * SJ + tj;
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int hj = 0; hj < 2; ++hj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gi = I * 32 + 16 * hi + kk + 4 * r, gj = J * 32 + 16 * hj + col;
                        if (gi < N && gj < N) {
                            C[(size_t)gi * N + gj] = acc[ti][tj][hi][hj][r];
                            if (I != J) C[(size_t)gj * N + gi] = acc[ti][tj][hi][hj][r];
                        }
                    }
        }
}

__global__ void __launch_bounds__(64)
syrk_lt2_kernel_f32(const float* __restrict__ Linv, float* __restrict__ Kinv, int N, int nb2) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    int t = blockIdx.x, SI = 0;
    while ((SI + 1) * (SI + 2) / 2 <= t) ++SI;
    const int SJ = t - SI * (SI + 1) / 2;
    const float* A = Linv + (size_t)blockIdx.y * N * N;
    float* C = Kinv + (size_t)blockIdx.y * N * N;
    const int lane = threadIdx.x, col = lane & 31, kk = lane >> 5;
    f32x16 acc[2][2] = {};
    for (int k0 = SI * 64; k0 < N; k0 += 2 * 4) {
        float a[4][2], b[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + 2 * q + kk;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int ci = (2 * SI + ti) * 32 + col, cj = (2 * SJ + ti) * 32 + col;
                a[q][ti] = (ci < N && k < N) ? A[(size_t)k * N + ci] : 0.0f;
                b[q][ti] = (cj < N && k < N) ? A[(size_t)k * N + cj] : 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    if (SI == SJ && tj > ti) continue;
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][ti], b[q][tj], acc[ti][tj], 0, 0, 0);
                }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            if (SI == SJ && tj > ti) continue;
            const int I = 2 * SI + ti, J = 2 * SJ + tj;
            const int cj = J * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gi = I * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
                if (gi < N && cj < N) {
                    C[(size_t)gi * N + cj] = acc[ti][tj][r];
                    if (I != J) C[(size_t)cj * N + gi] = acc[ti][tj][r];
                }
            }
        }
}

// Batches, N a multiple of 128 (round 6): a 128 x 128 output block per WORKGROUP of four waves (a 64 x 64 quadrant each), the two operand panels -- rows k of L^-1,
// columns of block row BI and of block row BJ -- streamed through a four-slot LDS ring by `buffer_load ... lds` (1 KB per copy: two fp32 rows / one fp64 row of 128
// columns; a wave issues a quarter of a slab's copies, a counted vmcnt and one workgroup barrier per slab publish them).  Against the 2 x 2-tile form above: half the
// operand bytes per flop, an eighth of the vector-memory instructions, and the whole read sequence in flight ahead of the MFMAs instead of one exposed round trip per
// sixteen of them.  Workgroups are dealt so that the blocks of a model share an XCD (its L2 holds the model's L^-1 rows for all ten blocks).
constexpr int SY_RING = 4, SY_KC = 8;
#ifndef BCBF_SY_KC32
#define BCBF_SY_KC32 8
#endif
constexpr int SY_KC32 = BCBF_SY_KC32;     // fp32: rows per slab (8: 32 KB of LDS per workgroup; 16: 64 KB, measured the same 3.1 ms at 4096 x 512)
__global__ void __launch_bounds__(256, 2)
syrk_tile_kernel_f32(const float* __restrict__ Linv, float* __restrict__ Kinv, int Bt, int N, int nb) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    __shared__ __attribute__((aligned(16))) float ring[SY_RING][2][SY_KC32][128];
    const int nblocks = nb * (nb + 1) / 2;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int b = (pos / nblocks) * 8 + xcd;
    if (b >= Bt) return;
    int t = pos % nblocks, BI = 0;
    while ((BI + 1) * (BI + 2) / 2 <= t) ++BI;
    const int BJ = t - BI * (BI + 1) / 2;
    const float* A = Linv + (size_t)b * N * N;
    float* C = Kinv + (size_t)b * N * N;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, kh = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)N * N * 4), 0x00020000);
    // slab s: rows k0 + 8 s .. + 7; copy p of 8: p < 4: rows 2 p, 2 p + 1 of the BI panel, p >= 4: of the BJ panel; this wave issues copies `wave` and `wave + 4`
    const int k0 = 128 * BI, nslab = (N - k0) / SY_KC32;
    const int voff = (kh * N + 4 * col) * 4;
    int sissue = 0, wslot = 0;
    auto issue = [&]() {
        const bool valid = sissue < nslab;
#pragma unroll
        for (int h = 0; h < SY_KC32 / 8; ++h) {
            const int row = k0 + SY_KC32 * sissue + 2 * wave + 8 * h;
            __attribute__((address_space(3))) float* dst = (__attribute__((address_space(3))) float*)&ring[wslot][0][2 * wave + 8 * h][0];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, valid ? voff : 0x7ffffff0, valid ? (row * N + 128 * BI) * 4 : 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + SY_KC32 * 128), 16, valid ? voff : 0x7ffffff0, valid ? (row * N + 128 * BJ) * 4 : 0, 0, 0);
        }
        wslot = wslot + 1 == SY_RING ? 0 : wslot + 1;
        ++sissue;
    };
    for (int a = 0; a < SY_RING - 1; ++a) issue();
    f32x16 acc[2][2] = {};
    int rslot = 0;
    // L^-1 is lower triangular: column c is zero above row c, so tile (ti, tj) of this wave starts contributing at row max(first column of its A tile, of its B
    // tile) -- a multiple of 32, hence of the slab; a quadrant above the diagonal of a diagonal block contributes nothing that is stored (the wave only copies).
    // Without this a 128-row granularity costs 1.9 x the algorithmic flops (84 against 45 MFLOP per model at N = 512), with it 1.2 x.
    const bool dead = BI == BJ && wj > wi, diagq = BI == BJ && wi == wj;
    const int cA = 128 * BI + 64 * wi, cB = 128 * BJ + 64 * wj;
    for (int s = 0; s < nslab; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (SY_KC32 / 8) * (SY_RING - 2)) : "memory");      // this wave's copies of slab s have landed
        __builtin_amdgcn_s_barrier();                                                  // ... everybody's; slab s - 1 has been read by all
        asm volatile("" ::: "memory");
        issue();
        const __attribute__((address_space(3))) float* as = (const __attribute__((address_space(3))) float*)&ring[rslot][0][kh][64 * wi + col];
        const __attribute__((address_space(3))) float* bs = (const __attribute__((address_space(3))) float*)&ring[rslot][1][kh][64 * wj + col];
        rslot = rslot + 1 == SY_RING ? 0 : rslot + 1;
        const int ks = k0 + SY_KC32 * s;
        if (dead || ks < cA) continue;                                                 // (wave-uniform)
        const bool t1 = ks >= cA + 32;                                                 // the second row of tiles has started
        if (t1 && !diagq) {
#pragma unroll
            for (int q = 0; q < SY_KC32 / 2; ++q) {
                const float a0 = as[2 * q * 128], a1 = as[2 * q * 128 + 32], b0 = bs[2 * q * 128], b1 = bs[2 * q * 128 + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else {                                                                       // the head of a quadrant / a diagonal quadrant: per-tile tests (wave-uniform)
#pragma unroll
            for (int q = 0; q < SY_KC32 / 2; ++q) {
                const float a0 = as[2 * q * 128], a1 = as[2 * q * 128 + 32], b0 = bs[2 * q * 128], b1 = bs[2 * q * 128 + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                if (!diagq) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);       // (diagonal quadrant: tile (0, 1) lies above the diagonal)
                if (t1) {
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
        }
    }
    // ---- stores.  Direct image: lane = column: 128-byte row segments.  Mirror image (row gj, columns = this tile's rows): through LDS (the ring is free now), so
    //      that eight lanes cover a 128-byte row segment (element-wise mirrored stores -- one request per 4 bytes -- took longer than the MFMAs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // (the cursor's trailing dummy copies have landed: they would overwrite what follows)
    __syncthreads();                                                                   // every wave is done with the ring
    if (dead) return;
    __attribute__((address_space(3))) float* tb = (__attribute__((address_space(3))) float*)&ring[0][0][0][0] + wave * (32 * 36);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int gi0 = 128 * BI + 64 * wi + 32 * ti, gj0 = 128 * BJ + 64 * wj + 32 * tj;
            if (gi0 < gj0) continue;                                                   // (tile above the diagonal inside a diagonal quadrant)
#pragma unroll
            for (int r = 0; r < 16; ++r) C[(size_t)(gi0 + 8 * (r >> 2) + 4 * kh + (r & 3)) * N + gj0 + col] = acc[ti][tj][r];
            if (gi0 != gj0) {
                using f4 = float __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f4 v = {acc[ti][tj][4 * a], acc[ti][tj][4 * a + 1], acc[ti][tj][4 * a + 2], acc[ti][tj][4 * a + 3]};
                    *(__attribute__((address_space(3))) f4*)&tb[col * 36 + 8 * a + 4 * kh] = v;       // [column n][row m]
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n_ = 8 * j + (lane >> 3), m4 = lane & 7;
                    const f4 v = *(const __attribute__((address_space(3))) f4*)&tb[n_ * 36 + 4 * m4];
                    *reinterpret_cast<float4*>(C + (size_t)(gj0 + n_) * N + gi0 + 4 * m4) = float4{v.x, v.y, v.z, v.w};
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
}

__global__ void __launch_bounds__(256, 2)
syrk_tile_kernel_f64(const double* __restrict__ Linv, double* __restrict__ Kinv, int Bt, int N, int nb) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    __shared__ __attribute__((aligned(16))) double ring[SY_RING][2][SY_KC][128];
    const int nblocks = nb * (nb + 1) / 2;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int b = (pos / nblocks) * 8 + xcd;
    if (b >= Bt) return;
    int t = pos % nblocks, BI = 0;
    while ((BI + 1) * (BI + 2) / 2 <= t) ++BI;
    const int BJ = t - BI * (BI + 1) / 2;
    const double* A = Linv + (size_t)b * N * N;
    double* C = Kinv + (size_t)b * N * N;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c16 = lane & 15, kq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A), 0, (unsigned)min((size_t)0xfffffff0u, (size_t)N * N * 8), 0x00020000);
    // slab s: rows k0 + 8 s .. + 7 of both panels = 16 copies of one row (128 doubles); this wave issues rows 2 wave, 2 wave + 1 of each panel
    const int k0 = 128 * BI, nslab = (N - k0) / SY_KC;
    int sissue = 0, wslot = 0;
    auto issue = [&]() {
        const bool valid = sissue < nslab;
        const int voff = valid ? 16 * lane : 0x7ffffff0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = k0 + SY_KC * sissue + 2 * wave + h;
            __attribute__((address_space(3))) double* dst = (__attribute__((address_space(3))) double*)&ring[wslot][0][2 * wave + h][0];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, valid ? (row * N + 128 * BI) * 8 : 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + SY_KC * 128), 16, voff, valid ? (row * N + 128 * BJ) * 8 : 0, 0, 0);
        }
        wslot = wslot + 1 == SY_RING ? 0 : wslot + 1;
        ++sissue;
    };
    for (int a = 0; a < SY_RING - 1; ++a) issue();
    f64x4 acc[4][4] = {};
    int rslot = 0;
    // (triangular structure as in the fp32 kernel, at the 16-row granularity of the fp64 MFMA: row group ui of this wave starts at row cA + 16 ui; in a diagonal
    //  quadrant the tiles above the diagonal are never formed)
    const bool dead = BI == BJ && wj > wi, diagq = BI == BJ && wi == wj;
    const int cA = 128 * BI + 64 * wi;
    for (int s = 0; s < nslab; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (SY_RING - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const __attribute__((address_space(3))) double* as = (const __attribute__((address_space(3))) double*)&ring[rslot][0][kq][64 * wi + c16];
        const __attribute__((address_space(3))) double* bs = (const __attribute__((address_space(3))) double*)&ring[rslot][1][kq][64 * wj + c16];
        rslot = rslot + 1 == SY_RING ? 0 : rslot + 1;
        const int ks = k0 + SY_KC * s;
        if (dead || ks < cA) continue;
        const int nrow = min(4, (ks - cA) / 16 + 1);                                   // row groups that have started (wave-uniform)
#pragma unroll
        for (int q = 0; q < SY_KC / 4; ++q) {
            double bb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bb[u] = bs[4 * q * 128 + 16 * u];
#pragma unroll
            for (int ui = 0; ui < 4; ++ui)
                if (ui < nrow) {
                    const double a = as[4 * q * 128 + 16 * ui];
#pragma unroll
                    for (int uj = 0; uj < 4; ++uj)
                        if (uj <= ui || !diagq) acc[ui][uj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb[uj], acc[ui][uj], 0, 0, 0);
                }
        }
    }
    // ---- stores.  Direct image: a lane's column, rows kq + 4 r: 128-byte row segments.  Mirror image (row gj, columns = this tile's rows): a lane's four rows are
    //      4 apart, so the tile goes through LDS (the ring is free now: 16 x 18 doubles per wave) and comes back as four consecutive rows per lane: 32-byte runs
    //      (element-wise mirrored stores -- one request per 8 bytes -- took longer than the MFMAs).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // (the cursor's trailing dummy copies have landed: they would overwrite what follows)
    __syncthreads();                                                                   // every wave is done with the ring
    if (dead) return;
    __attribute__((address_space(3))) double* tb = (__attribute__((address_space(3))) double*)&ring[0][0][0][0] + wave * (16 * 18);
#pragma unroll
    for (int ui = 0; ui < 4; ++ui)
#pragma unroll
        for (int uj = 0; uj < 4; ++uj) {
            const int gi0 = 128 * BI + 64 * wi + 16 * ui, gj0 = 128 * BJ + 64 * wj + 16 * uj;
            if (gi0 < gj0) continue;                                                   // (above the diagonal inside a diagonal quadrant: never formed)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(gi0 + kq + 4 * r) * N + gj0 + c16] = acc[ui][uj][r];
            if (gi0 != gj0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tb[c16 * 18 + kq + 4 * r] = acc[ui][uj][r];      // [column n][row m]
                __builtin_amdgcn_wave_barrier();
                using d2 = double __attribute__((ext_vector_type(2)));
                const d2 lo = *(const __attribute__((address_space(3))) d2*)&tb[c16 * 18 + 4 * kq], hi = *(const __attribute__((address_space(3))) d2*)&tb[c16 * 18 + 4 * kq + 2];
                __builtin_amdgcn_wave_barrier();
                double* dst = C + (size_t)(gj0 + c16) * N + gi0 + 4 * kq;               // row gj0 + c16, columns gi0 + 4 kq .. + 3
                *reinterpret_cast<double2*>(dst) = double2{lo.x, lo.y};
                *reinterpret_cast<double2*>(dst + 2) = double2{hi.x, hi.y};
            }
        }
}

}  // namespace bcbf

extern "C" int bcbf_syrk_lt_f32(const float* Linv, float* Kinv, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Linv || !Kinv || N < 1 || Linv == Kinv) return BCBF_EINVAL;
    const int nb = (N + 31) / 32;
    static const int tile_form = [] { const char* e = getenv("BCBF_SYRK_TILE"); return e ? atoi(e) : 1; }();      // (development: 0 = the 2 x 2-tile form)
    if (Bt >= 16 && N % 128 == 0 && N <= 8192 && tile_form) {     // batches: a 128 x 128 block per workgroup, operands through LDS
        const int nb128 = N / 128;
        const long long groups = ((long long)Bt + 7) / 8, wgs = groups * 8 * (nb128 * (nb128 + 1) / 2);
        if (wgs > 0x7fffffffLL) return BCBF_EINVAL;
        hipLaunchKernelGGL(bcbf::syrk_tile_kernel_f32, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, Linv, Kinv, Bt, N, nb128);
        return bcbf::check_launch("syrk_tile");
    }
    if (Bt >= 16 && nb >= 4) {                     // batches: 2 x 2 tiles per wave (a single model keeps the one-tile form: more waves)
        const int nb2 = (nb + 1) / 2;
        hipLaunchKernelGGL(bcbf::syrk_lt2_kernel_f32, dim3(nb2 * (nb2 + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb2);
        return bcbf::check_launch("syrk_lt");
    }
    hipLaunchKernelGGL(bcbf::syrk_lt_kernel_f32, dim3(nb * (nb + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb);
    return bcbf::check_launch("syrk_lt");
}
extern "C" int bcbf_syrk_lt_f64(const double* Linv, double* Kinv, int Bt, int N, void* stream) {
    if (Bt <= 0) return BCBF_OK;
    if (!Linv || !Kinv || N < 1 || Linv == Kinv) return BCBF_EINVAL;
    const int nb = (N + 31) / 32;
    static const int tile_form = [] { const char* e = getenv("BCBF_SYRK_TILE"); return e ? atoi(e) : 1; }();
    if (Bt >= 16 && N % 128 == 0 && N <= 8192 && tile_form) {
        const int nb128 = N / 128;
        const long long groups = ((long long)Bt + 7) / 8, wgs = groups * 8 * (nb128 * (nb128 + 1) / 2);
        if (wgs > 0x7fffffffLL) return BCBF_EINVAL;
        hipLaunchKernelGGL(bcbf::syrk_tile_kernel_f64, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, Linv, Kinv, Bt, N, nb128);
        return bcbf::check_launch("syrk_tile");
    }
    if (Bt >= 16 && nb >= 4) {
        const int nb2 = (nb + 1) / 2;
        hipLaunchKernelGGL(bcbf::syrk_lt2_kernel_f64, dim3(nb2 * (nb2 + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb2);
        return bcbf::check_launch("syrk_lt");
    }
    hipLaunchKernelGGL(bcbf::syrk_lt_kernel_f64, dim3(nb * (nb + 1) / 2, Bt), dim3(64), 0, (hipStream_t)stream, Linv, Kinv, N, nb);
    return bcbf::check_launch("syrk_lt");
}
