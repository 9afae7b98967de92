// Rel-degree-2 jets (bcbf_posterior_jets, fp32, N <= 512, (1 + m)(1 + n) + n <= 16) on the MATRIX CORES, the operator through an LDS-DMA ring (round 6).
//
// The streaming kernel (posterior_step.hip) forms  r -= L[:, j] w_j  with packed VALU multiply-adds: 12 right-hand-side columns are 96 accumulators and 48
// v_pk_fma per lane and operator column, a 192-term mat-vec plus shuffles in every diagonal step, and a register ring of two column groups beside them:
// 0.62 of the HBM rate for three rounds (4096 x 512, n = 3, m = 2: 0.447 ms).  Here one wave owns an instance and
//   * the residual R [N x 16] (the right-hand sides padded to 16 columns) lives in MFMA accumulators: 64-row group g = four 16x16x4 tiles acc[g][v] whose
//     row mu is row 64 g + 4 mu + v -- so that the A operand of the four tiles (L[64 g + 4 i + v][column c], lane (i, c)) is ONE 16-byte load of four
//     consecutive rows of a column: a 1 KB copy = 64 rows x 4 columns, four MFMAs (one per v) with the B operand w[column c][rhs] read once per four columns;
//   * all global reads are `buffer_load ... lds` into a ring of 1 KB slots: the read sequence -- per block column J: its packed inverted diagonal block
//     (3 slots), its Vw rows (1), then the column groups' pieces for the live row groups -- is known up front, a run-time cursor runs RING - 1 pieces ahead,
//     a consumer waits with a counted vmcnt and reads back the 16 bytes its own lane copied (no barrier: one wave; no s_waitcnt the compiler places);
//   * the diagonal step is 16 MFMAs (w_J = inv(L_JJ) r_J: A from the packed triangle in its slot, B = r_J published to LDS) instead of 192 multiply-adds
//     and a cross-half shuffle per column; the Gram / mean sums are the eight MFMAs of the streaming kernel's step 2b.
// Rows above a block column's diagonal are not stored: their lanes' copies are out of range (nothing lands), the stale slot contents they multiply go into
// residual rows that were solved already and are never read again.
#include <type_traits>
#include "bcbf_common.h"

namespace bcbf {

using jf4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
#define JM_AS3 __attribute__((address_space(3)))
#ifndef BCBF_JM_RING
#define BCBF_JM_RING 16
#endif
#ifndef BCBF_JM_AUX
#define BCBF_JM_AUX 2        // cache policy of the operator copies: 2 = non-temporal (every byte is read once): 0.431 -> 0.399 ms at 4096 x 512, n = 3, m = 2
#endif
#ifndef BCBF_JM_OCC
#define BCBF_JM_OCC 2        // waves per SIMD: one wave alone (32 slots, 512 registers) streams at 3.4 TB/s, two (16 slots each) at 5.2 -- the copies of one wave do not keep the CU's memory path busy
#endif
constexpr int JM_RING = BCBF_JM_RING;            // 1 KB slots
constexpr int JM_LAG = 4;                        // a copy refills the slot consumed JM_LAG pieces ago: the four header pieces of a block column (diagonal block, Vw rows)
                                                 // are read after all four have been taken; JM_RING - JM_LAG pieces (28 KB per wave) in flight
constexpr int JM_NG = 8;                         // 64-row groups (N <= 512)

template <int C, int NJ>
__global__ void __launch_bounds__(64, BCBF_JM_OCC)
posterior_jets_mfma_kernel(const float* __restrict__ Lop, const float* __restrict__ Vw, const float* __restrict__ X,
                           const float* __restrict__ UHB, const float* __restrict__ ell, const float* __restrict__ s2p,
                           const float* __restrict__ Bm, const float* __restrict__ M0, const float* __restrict__ xq,
                           float* __restrict__ Mk, float* __restrict__ Bk, float* __restrict__ Wout, float* __restrict__ Gfull,
                           float* __restrict__ Mfull, int shared, int N, int Np, int n, int kind) {
    constexpr int CT = C * (1 + NJ);
    static_assert(CT + NJ <= 16, "[W, Vw] is one 16-column tile");
    __shared__ __attribute__((aligned(16))) float ring[JM_RING][256];
    __shared__ __attribute__((aligned(16))) float stage[64][16];       // Phi staging per row group; then rbuf = rows 0..31, wbuf = rows 32..63
    JM_AS3 float (&rbuf)[64][16] = *(JM_AS3 float (*)[64][16])&stage[0][0];
    const int b = blockIdx.x, gb = shared ? 0 : b, lane = threadIdx.x, i16 = lane & 15, kc = lane >> 4;
    const float* __restrict__ lop = Lop + (size_t)gb * lop_elems<4>(Np);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lop), 0, (unsigned)(lop_elems<4>(Np) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vw + (size_t)gb * N * n), 0, (unsigned)(N * n * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (size_t)gb * N * n), 0, (unsigned)(N * n * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(UHB + (size_t)gb * N * C), 0, (unsigned)(N * C * 4), 0x00020000);
    const int nblk = Np / NB, ngrp = (Np + 63) / 64;
    const float s2 = s2p[gb];
    float xqr[4], iell[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        xqr[d] = d < n ? xq[(size_t)b * n + d] : 0.f;
        iell[d] = d < n ? 1.f / ell[(size_t)gb * n + d] : 0.f;
    }

    // ---- the read sequence.  Cursor: block column cJ, phase cp (0..2: diagonal block pieces, 3: Vw rows, 4: the column groups), column group ccg, row group cgp;
    //      per block column: cs4 = 16 cs (bytes between column groups), scur = scalar offset of the current column group, vbase = per-lane offset of row group 0
    //      (relative to the column's first stored row: negative = far out of range above it), vcur = that of the current piece.  The common case (a stream
    //      piece) is a handful of scalar instructions: formed per piece from (cJ, cg, cgp) the packed layout's polynomial cost more issue time than the four MFMAs.
    static_assert((JM_RING & (JM_RING - 1)) == 0, "ring positions wrap by masking");
    int cJ = 0, cp = 0, ccg = 0, cgp = 0, wslot = 0, cs4 = 0, scur = 0, cg0 = 0;
    int vbase = 0, vcur = 0;
    auto issue_next = [&]() {
        JM_AS3 void* dst = (JM_AS3 void*)&ring[wslot][0];
        wslot = (wslot + 1) & (JM_RING - 1);
        if (cp == 4) {                                                 // a stream piece: row group cgp of column group ccg
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, dst, 16, vcur, scur, 0, BCBF_JM_AUX);
            vcur += 256;
            if (++cgp >= ngrp) {
                cgp = cg0; vcur = vbase + 256 * cg0; scur += cs4;
                if (++ccg == 8) { ++cJ; cp = 0; }
            }
            return;
        }
        if (cJ >= nblk) {                                              // past the end: a dummy keeps the counted wait's arithmetic
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, dst, 16, 0x7ffffff0, 0, 0, 0);
            return;
        }
        if (cp < 3) {
            // (the packed block is 544 words: the third piece needs its first eight lanes only)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, dst, 16, (cp < 2 || lane < (LOP_DB - 512) / 4) ? 16 * lane : 0x7ffffff0, (lop_dinv_block(cJ, Np) + 256 * cp) * 4, 0, 0);
            ++cp;
            return;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, dst, 16, lane < 8 * n ? 16 * lane : 0x7ffffff0, cJ * NB * n * 4, 0, 0);      // (32 rows x n words)
        cg0 = (cJ + 1) >> 1;
        if (cg0 >= ngrp) { ++cJ; cp = 0; return; }                     // (last block column: nothing below it)
        const int cs = Np - NB * (cJ + 1);
        cp = 4; ccg = 0; cgp = cg0; cs4 = 16 * cs;
        scur = (lop_base<4>(cJ * NB, Np) + NB * (cJ + 1)) * 4;
        vbase = (kc * cs + 4 * i16 - NB * (cJ + 1)) * 4;
        vcur = vbase + 256 * cg0;
    };
    for (int a = 0; a < JM_RING - JM_LAG; ++a) issue_next();
    int rslot = 0;
    // the next piece of the sequence has landed; returns its slot (and issues one more copy: the ring stays JM_RING - JM_LAG ahead)
    auto next_piece = [&]() -> JM_AS3 const float* {
        issue_next();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(JM_RING - JM_LAG) : "memory");
        JM_AS3 const float* p = (JM_AS3 const float*)&ring[rslot][0];
        rslot = (rslot + 1) & (JM_RING - 1);
        return p;
    };

    // ---- right-hand sides: lane = row computes Phi and its jets, the group's 64 x 16 tile goes through LDS into the accumulator layout
    //      acc[g][v][r] (lane (i16, kc)) = R[64 g + 16 kc + 4 r + v][i16]
    jf4 acc[JM_NG][4];
#pragma unroll
    for (int g = 0; g < JM_NG; ++g) {
        if (g < ngrp) {
            const int row = 64 * g + lane;
            float xv[4], uv[C];
#pragma unroll
            for (int d = 0; d < 4; ++d) xv[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, (row < N && d < n) ? (row * n + d) * 4 : 0x7ffffff0, 0, 0));
#pragma unroll
            for (int c = 0; c < C; ++c) uv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsU, row < N ? (row * C + c) * 4 : 0x7ffffff0, 0, 0));
            float d2 = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d) { const float z = (xv[d] - xqr[d]) * iell[d]; d2 += z * z; }
            float shape, dshape;
            if (kind != 0) kernel_shape(kind, d2, [](float q_) { return expf(q_); }, shape, dshape);
            else { shape = expf(-0.5f * d2); dshape = shape; }
            const float k = s2 * shape, kd = s2 * dshape;
            float phi[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) phi[c] = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                phi[c] = k * uv[c];
#pragma unroll
                for (int d = 0; d < NJ; ++d) phi[(1 + d) * C + c] = (xv[d] - xqr[d]) * iell[d] * iell[d] * kd * uv[c];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *(JM_AS3 jf4*)&rbuf[lane][4 * q] = jf4{phi[4 * q], phi[4 * q + 1], phi[4 * q + 2], phi[4 * q + 3]};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][v][r] = rbuf[16 * kc + 4 * r + v][i16];
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[g][v] = jf4{0.f, 0.f, 0.f, 0.f};
        }
    }
    jf4 gacc = {0.f, 0.f, 0.f, 0.f}, gacc2 = {0.f, 0.f, 0.f, 0.f};         // [W, Vw]' [W, Vw] summed over all rows (two chains)
    // per-lane offsets into a packed inverted diagonal block: inv(L_JJ)[row 16 h + i16][column 4 s + kc]
    int doff[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int r_ = 16 * h + i16, c_ = 4 * s + kc;
            doff[h][s] = r_ >= c_ ? lop_dinv_col(c_) + r_ : -1;
        }

    for (int J = 0; J < nblk; ++J) {
        // ---- 1. publish r_J: rows 32 J .. of group J / 2, held by the lanes with kc / 2 == J % 2 as row 16 (kc % 2) + 4 r + v of the block
        JM_AS3 const float* d0 = next_piece();
        JM_AS3 const float* d1 = next_piece();
        JM_AS3 const float* d2p = next_piece();
        JM_AS3 const float* vwp = next_piece();
        (void)d1; (void)d2p;                                           // (the three diagonal pieces are consecutive slots unless the ring wraps: read by index below)
#pragma unroll
        for (int g = 0; g < JM_NG; ++g)
            if (g == (J >> 1) && (kc >> 1) == (J & 1)) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rbuf[16 * (kc & 1) + 4 * r + v][i16] = acc[g][v][r];
            }
        __builtin_amdgcn_wave_barrier();
        // ---- 2. w_J = inv(L_JJ) r_J: two 16-row halves x eight k-steps
        jf4 w[2] = {jf4{0.f, 0.f, 0.f, 0.f}, jf4{0.f, 0.f, 0.f, 0.f}};
        const int dslot = (int)((d0 - (JM_AS3 const float*)&ring[0][0]) >> 8);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float bq = rbuf[4 * s + kc][i16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int o = doff[h][s];                               // element o of the block: slot dslot + o / 256 (mod the ring), word o % 256
                const int sl = dslot + (o >> 8);
                const float a = o >= 0 ? ring[sl >= JM_RING ? sl - JM_RING : sl][o & 255] : 0.f;
                w[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, w[h], 0, 0, 0);
            }
        }
        // w[h][r] (lane (i16, kc)) = W[32 J + 16 h + 4 kc + r][i16]; with the block's Vw rows beside it the tile T_J = [w, Vw, 0] (32 x 16) goes to LDS
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = 16 * h + 4 * kc + r;
                float val = w[h][r];
                if (i16 >= CT) val = (i16 < CT + NJ && i16 - CT < n && J * NB + rw < N) ? vwp[rw * n + (i16 - CT)] : 0.f;      // (rows of the padding: the copy's lanes were out of range, the slot holds stale words)
                rbuf[32 + rw][i16] = val;
                if (Wout != nullptr && i16 < CT) Wout[((size_t)b * Np + J * NB + rw) * CT + i16] = val;
            }
        __builtin_amdgcn_wave_barrier();
        // ---- 2b. Gram and mean sums: gacc += T_J' T_J  (two chains: an MFMA on the accumulator of the one before waits for it)
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            const float t0 = rbuf[32 + 4 * s + kc][i16], t1 = rbuf[32 + 4 * s + 4 + kc][i16];
            gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0, t0, gacc, 0, 0, 0);
            gacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1, t1, gacc2, 0, 0, 0);
        }
        // ---- 3. rows below the block: R -= L[:, J] w_J, four columns and one 64-row group per piece.  The piece after the current one is read from its
        //      slot BEFORE the current piece's MFMAs issue (a wave issues in order and stalls at an MFMA while the pipe is busy: what sits in front of the
        //      MFMAs -- the cursor, the copy, the counted wait, the LDS read -- runs under the previous piece's; read behind them it is exposed)
        const int g0 = (J + 1) >> 1;
        if (g0 < ngrp) {
            int left = 8 * (ngrp - g0);                                  // pieces of this block column's stream
            jf4 cur = *(JM_AS3 const jf4*)(next_piece() + 4 * lane);
            float wn = -rbuf[32 + kc][i16];                              // B operand of column group 0: -w[column kc][rhs i16]
            for (int cg = 0; cg < 8; ++cg) {
                const float wnn = cg + 1 < 8 ? -rbuf[32 + 4 * (cg + 1) + kc][i16] : 0.f;
#pragma unroll
                for (int g = 0; g < JM_NG; ++g)
                    if (g >= g0 && g < ngrp) {
                        jf4 nxt = cur;
#ifndef JM_ABL_NOMFMA
                        acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0], wn, acc[g][0], 0, 0, 0);
#else
                        acc[g][0][0] += cur[0] * wn;
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        if (--left > 0) nxt = *(JM_AS3 const jf4*)(next_piece() + 4 * lane);
                        __builtin_amdgcn_sched_barrier(0);
#ifndef JM_ABL_NOMFMA
                        acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[1], wn, acc[g][1], 0, 0, 0);
                        acc[g][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[2], wn, acc[g][2], 0, 0, 0);
                        acc[g][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[3], wn, acc[g][3], 0, 0, 0);
#endif
                        cur = nxt;
                    }
                wn = wnn;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) gacc[r] += gacc2[r];
    // ---- epilogue: lane (i16, kc) holds gacc rows 4 kc + r, column i16 of [W, Vw]'[W, Vw]
    float* Gb = Gfull + (size_t)b * CT * CT;
    float* Mb = Mfull + (size_t)b * n * CT;
    const float* M0b = M0 + (size_t)gb * C * n;
    const float* Bmb = Bm + (size_t)gb * C * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * kc + r, j = i16;
        const float val = gacc[r];
        if (j < CT) {
            if (i < CT) {
                Gb[i * CT + j] = val;
                if (i < C && j < C) Bk[(size_t)b * C * C + i * C + j] = (float)((double)s2 * (double)Bmb[i * C + j] - (double)val);
            } else if (i < CT + n) {
                const int d = i - CT;
                Mb[d * CT + j] = val;
                if (j < C) Mk[(size_t)b * n * C + d * C + j] = M0b[j * n + d] + val;
            }
        }
    }
}

bool posterior_jets_mfma_fits(int N, int n, int m) {
    const int C = m + 1;
    return N <= 64 * JM_NG && n <= 4 && C * (1 + n) + n <= 16 && (n == 3 || n == 2) && (m == 1 || m == 2);
}
// ... and is the faster form: measured at 4096 x 512 (ms, streaming kernel / this one): (n, m) = (2, 1): 0.345 / 0.395, (2, 2): 0.370 / 0.399, (3, 1): 0.371 / 0.399,
// (3, 2): 0.445 / 0.401 -- this form's time does not depend on the column count (it is bound by what two waves per SIMD get out of the copy path: 5.6 TB/s), the
// streaming kernel's does (its multiply-adds): from twelve columns on this one wins
bool posterior_jets_mfma_preferred(int N, int n, int m) { return posterior_jets_mfma_fits(N, n, m) && (m + 1) * (1 + n) >= 12; }

// 0 = launched, 1 = a shape this form does not take
int launch_posterior_jets_mfma(const float* Lop, const float* Vw, const float* X, const float* UHB, const float* ell, const float* s2,
                               const float* Bm, const float* M0, const float* xq, float* Mk, float* Bk, float* Wout, float* Gfull, float* Mfull,
                               int shared, int Bt, int N, int n, int m, int kind, hipStream_t st) {
    if (!posterior_jets_mfma_fits(N, n, m)) return 1;
    const int Np = round_up(N, NB);
#define BCBF_JM_LAUNCH(CC, NN) hipLaunchKernelGGL((posterior_jets_mfma_kernel<CC, NN>), dim3(Bt), dim3(64), 0, st, Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, Mk, Bk, Wout, Gfull, Mfull, shared, N, Np, n, kind)
    switch (10 * n + m) {
        case 21: BCBF_JM_LAUNCH(2, 2); break;
        case 22: BCBF_JM_LAUNCH(3, 2); break;
        case 31: BCBF_JM_LAUNCH(2, 3); break;
        case 32: BCBF_JM_LAUNCH(3, 3); break;
        default: return 1;
    }
#undef BCBF_JM_LAUNCH
    return 0;
}

}  // namespace bcbf
