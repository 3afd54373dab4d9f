// Shared by the operand-ready GEMMs (dct_pair_f64.hip, dct_pair_f32.hip) and their pre-passes
// (dct_pair_prep.hip).
#pragma once
#include "dct_common.hpp"

namespace ssw {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T> using vec4_t = T __attribute__((ext_vector_type(4)));
template <typename T> using vec2_t = T __attribute__((ext_vector_type(2)));

// k-blocked operand planes: [Kp / KB][lines][KB] elements with KB * sizeof(T) = 64 bytes, i.e. the
// 64-byte piece of every line that one GEMM k-step needs lies next to its neighbours'.
template <typename T> struct KBlock { static constexpr unsigned KB = 64 / sizeof(T); };
// element (line, k) of a k-blocked plane with `rows` lines
template <typename T>
__host__ __device__ inline size_t blk_index(size_t line, unsigned k, size_t rows) {
    constexpr unsigned KB = KBlock<T>::KB;
    return ((size_t)(k / KB) * rows + line) * KB + (k % KB);
}

// Epilogues.  n = transform length, idx = output index along the transformed axis:
//   EPI_FWD    out[c1 + cs pair] = acc1, out[c2 + cs pair] = acc2          (forward, any folding level)
//   EPI_FWD_ADJ  the same with c1 = 0, c2 = 1, cs = 2 on a row pass: one 8-byte store
//   EPI_INV    out[pair] = acc1 + acc2, out[n-1-pair] = acc1 - acc2        (inverse, one level)
//   EPI_INV_E  T[pair] = acc1 + acc2, T[n/2-1-pair] = acc1 - acc2, unrounded (inverse level 2: the even half E)
//   EPI_INV_O  with n1 = pair, n2 = pair + n/4:  out[n1] = T[n1] + acc1, out[n-1-n1] = T[n1] - acc1,
//              out[n2] = T[n2] + acc2, out[n-1-n2] = T[n2] - acc2          (inverse level 2: odd part + combine)
enum { EPI_FWD = 0, EPI_FWD_ADJ = 1, EPI_INV = 2, EPI_INV_E = 3, EPI_INV_O = 4 };

template <typename T>
struct PairOutT {
    float* out;          // f32 plane(s)
    T* tmp;              // E planes in the accumulation type (EPI_INV_E / EPI_INV_O)
    unsigned W, H;       // plane dims
    unsigned n;          // transform length (W for a row pass, H for a column pass)
    unsigned c1, c2, cs; // EPI_FWD
};

// row stride (in elements) of the operand planes / half bases of a length-n axis for precision T:
// n/2 rounded up so that the GEMM runs an even number of k-steps (f64: 8 per step, f32: 16)
template <typename T> inline size_t pair_kpad(size_t n) {
    const size_t m = 2 * KBlock<T>::KB;
    return ((n / 2 + m - 1) / m) * m;
}

}  // namespace ssw
