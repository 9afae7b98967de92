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

// Packed operator: element (i, j) of column j lives at lop_base(j) + i  (valid for i >= V*(j/V)).
template <int V>
__host__ __device__ inline int lop_base(int j, int Np) {
    const int g = j / V, c = j - g * V;
    return V * (g * Np - V * (g * (g - 1) / 2)) + c * (Np - V * g) - V * g;
}
template <int V>
__host__ __device__ inline size_t lop_elems(int Np) { return (size_t)Np * (Np + V) / 2; }

void set_error(const char* what, hipError_t err);
int check_launch(const char* what);

template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace bcbf
