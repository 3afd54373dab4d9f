// Default build (make without ALL_STRATEGIES=1): the transform strategies that the default path never takes -- in-kernel
// folding in f32 and f64 (dct_folded.hip, dct_folded_f64.hip: r1) and the f32 twin of the operand-ready GEMMs
// (dct_pair_f32.hip: non-parity and slower than the f64 path since r3) -- are not compiled in.  What remains is the f64
// pair path (dct_pair_f64*.hip + the pre-passes) and the dense kernels of dct.hip for everything else: shapes the pair path
// does not take (src/dct2d.rs:268-524's 3x3, 4x5 ... test shapes, the 444-row cat) and SSW_PRECISION_F32.
// `make ALL_STRATEGIES=1` builds the diagnostic library with every strategy; ssw_build_all_strategies() tells which one is loaded.
#include "ssw_internal.hpp"

namespace ssw {

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// the shape / alignment conditions of the folded transforms (the operand-ready path asks for them too)
bool dct_rows_can_fold(size_t w, const float* in, const float* out) { return w >= 16 && (w % 8 == 0) && aligned16(in) && aligned16(out); }
bool dct_cols_can_fold(size_t w, size_t h, const float* in, const float* out) {
    return h >= 16 && (h % 8 == 0) && (w % 4 == 0) && aligned16(in) && aligned16(out);
}

size_t half_basis_kpad(size_t n) { return ((n / 2 + 15) / 16) * 16; }        // sizes an allocation only (get_basis kinds 1 / 2 are never requested)
int launch_make_half_basis_f32(hipStream_t, size_t, bool, int, float*) { return SSW_ERR_UNSUPPORTED; }
int launch_make_half_basis_f64(hipStream_t, size_t, bool, int, double*) { return SSW_ERR_UNSUPPORTED; }
int launch_dct_rows_folded_f32(hipStream_t, bool, const float*, float*, size_t, size_t, const float*, const float*, Epilogue) { return SSW_ERR_UNSUPPORTED; }
int launch_dct_cols_folded_f32(hipStream_t, bool, const float*, float*, size_t, size_t, size_t, const float*, const float*, Epilogue) { return SSW_ERR_UNSUPPORTED; }
int launch_dct_rows_folded_f64(hipStream_t, bool, const float*, float*, size_t, size_t, const double*, const double*, Epilogue) { return SSW_ERR_UNSUPPORTED; }
int launch_dct_cols_folded_f64(hipStream_t, bool, const float*, float*, size_t, size_t, size_t, const double*, const double*, Epilogue) { return SSW_ERR_UNSUPPORTED; }
int launch_dct_pair_gemm_f32(hipStream_t, bool, bool, int, int, const float*, const float*, const float*, const float*, float*, float*, size_t, size_t,
                             size_t, Epilogue, const RgbSink*) { return SSW_ERR_UNSUPPORTED; }
int launch_dct_pair_gemm_rows_subset_f32(hipStream_t, const float*, const float*, unsigned, unsigned, float*, unsigned, unsigned, size_t) { return SSW_ERR_UNSUPPORTED; }

bool build_all_strategies() { return false; }

}  // namespace ssw
