// Shared device/host helpers for libbcbf (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "bcbf.h"

namespace bcbf {

constexpr int NB = BCBF_NB;  // diagonal block of the packed operator

template <typename T> struct Vec;
template <> struct Vec<float> { using type = float4; static constexpr int V = 4; };
template <> struct Vec<double> { using type = double2; static constexpr int V = 2; };

__host__ __device__ inline int round_up(int x, int q) { return (x + q - 1) / q * q; }

// Packed operator.  Column j stores rows lop_first(j) = 32*floor(j/32) .. Np-1 (the whole column from the top of
// its diagonal block), columns back to back: every column starts on a 128-byte boundary and has a multiple of
// 128 bytes, so a wave's 16-byte-per-lane loads never straddle a line shared with another column and the row-256
// split between a lane's two row blocks is line aligned.  Element (i, j) lives at lop_base(j) + i.
// (The template parameter V is the vector width of the caller; the layout itself does not depend on it.)
__host__ __device__ inline int lop_first(int j) { return (j / NB) * NB; }
template <int V>
__host__ __device__ inline int lop_base(int j, int Np) {
    const int J = j / NB, c = j - J * NB;
    return NB * (J * Np - NB * (J * (J - 1) / 2)) + c * (Np - NB * J) - NB * J;
}
template <int V>
__host__ __device__ inline size_t lop_elems(int Np) { return (size_t)Np * (Np + NB) / 2; }

void set_error(const char* what, hipError_t err);
int check_launch(const char* what);
bool posterior_shared_fits(int N, int n, int m);   // regime-S MFMA kernel: staging + one W slab fit in LDS

template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace bcbf
