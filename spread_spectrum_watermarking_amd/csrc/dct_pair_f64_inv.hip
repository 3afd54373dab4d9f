// Translation unit 0 of the inverse instances of the operand-ready f64 GEMM (dct_pair_f64_kernel.hpp, dct_pair_f64_inv.inc).
#ifndef SSW_TILE_TRACE
#define SSW_INV_PART 0
#include "dct_pair_f64_kernel.hpp"
#include "dct_pair_f64_inv.inc"
#endif
