// Device helpers shared by the dense (dct.hip) and even/odd-folded (dct_folded.hip) basis GEMMs.
#pragma once
#include "ssw_internal.hpp"

namespace ssw {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// XCD-aware, L2-friendly block -> tile map.  Blocks b, b+8, b+16, ... share an XCD (observed
// round-robin placement; speed only).  First give each XCD a contiguous run of tile ids, then
// walk tiles in groups of GROUP_M tile-rows, column-major inside a group, so that the ~64 blocks
// resident on one XCD cover a compact rectangle of tiles and share A/B panels in its L2
// (GROUP_M 4 measured 2.5 % faster than 8 on the 9-tile-wide column pass at 4K, equal elsewhere).
__device__ inline void tile_of_block(unsigned bid, unsigned nblk, unsigned tiles_m, unsigned tiles_n,
                                     unsigned& tm, unsigned& tn, unsigned group_m = 4) {
    const unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8;
    const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    const unsigned GROUP_M = group_m;
    const unsigned per_group = GROUP_M * tiles_n;
    const unsigned g = id / per_group;
    const unsigned first_m = g * GROUP_M;
    const unsigned gm = (tiles_m - first_m < GROUP_M) ? tiles_m - first_m : GROUP_M;
    const unsigned in_g = id % per_group;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
}

// The factors are resolved on the host (plain 1 / 1, orthonormal s0 / sn, inverse corr / corr): a
// per-element `if (mode ...)` compiled to scalar branches around every store and, for the scaled
// modes, to a dependent load of the factor per element.  acc * 1.0f is exact.
__device__ inline float apply_epilogue(const Epilogue& ep, float acc, unsigned out_idx) {
    return acc * (out_idx == 0 ? ep.first : ep.base);
}

// Load 4 consecutive floats of a k-contiguous row, zero beyond K.
template <bool ALIGNED>
__device__ inline f32x4 load_k4(const float* __restrict__ row, unsigned k, unsigned K) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ALIGNED) {
        if (k < K) v = *reinterpret_cast<const f32x4*>(row + k);
    } else {
        if (k + 0 < K) v[0] = row[k + 0];
        if (k + 1 < K) v[1] = row[k + 1];
        if (k + 2 < K) v[2] = row[k + 2];
        if (k + 3 < K) v[3] = row[k + 3];
    }
    return v;
}

template <bool ALIGNED>
__device__ inline f32x4 load_n4(const float* __restrict__ row, unsigned n, unsigned N, bool row_ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!row_ok) return v;
    if (ALIGNED) {
        if (n < N) v = *reinterpret_cast<const f32x4*>(row + n);
    } else {
        if (n + 0 < N) v[0] = row[n + 0];
        if (n + 1 < N) v[1] = row[n + 1];
        if (n + 2 < N) v[2] = row[n + 2];
        if (n + 3 < N) v[3] = row[n + 3];
    }
    return v;
}


// Image operand, k-contiguous row: aligned instances load unconditionally from a clamped address
// (what lies past the end of the sum axis multiplies a zero-padded basis entry).
template <bool ALIGNED>
__device__ inline f32x4 load_img_k4(const float* __restrict__ row, unsigned k, unsigned K) {
    if (ALIGNED) {
        const unsigned kc = k + 4 <= K ? k : K - 4;
        return *reinterpret_cast<const f32x4*>(row + kc);
    }
    return load_k4<false>(row, k, K);
}
// Basis operand (f32), rows zero-padded to dense_basis_kpad.
template <bool ALIGNED>
__device__ inline f32x4 load_basis_k4(const float* __restrict__ row, unsigned k, unsigned K) {
    if (ALIGNED) return *reinterpret_cast<const f32x4*>(row + k);
    return load_k4<false>(row, k, K);
}
// Image operand of the column pass: row kk of a K x N plane, 4 columns from n.
template <bool ALIGNED>
__device__ inline f32x4 load_img_n4(const float* __restrict__ plane, unsigned kk, unsigned K, unsigned n, unsigned N) {
    if (ALIGNED) {
        const unsigned kc = kk < K ? kk : K - 1;
        const unsigned nc = n + 4 <= N ? n : N - 4;
        return *reinterpret_cast<const f32x4*>(plane + (size_t)kc * N + nc);
    }
    return load_n4<false>(plane + (size_t)(kk < K ? kk : 0) * N, n, N, kk < K);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ssw
