#!/bin/bash
# Diagnostic variants of the library: dct_pair_f64.hip (the forward GEMM instances + launch logic) recompiled with extra -D flags and
# linked with the normal objects of csrc/ into spread_spectrum_watermarking_amd/lib/libssw_<name>.so (git-ignored; travels with gpurun;
# load with SSW_LIB_PATH).  Run `make -C csrc` first.  Timing-only ablations compute wrong values by design.
#   tools/build_variant.sh trace -DSSW_TILE_TRACE -DSSW_TILE_TRACE_FWD_ONLY     per-tile / per-k-step stamps (tools/tile_trace.py)
#   tools/build_variant.sh x0 -DSSW_ABL_X0                                      every block stages tile 0's lines (L2-resident operands)
#   tools/build_variant.sh noepi -DSSW_ABL_NOEPI                                no epilogue
# usage: tools/build_variant.sh NAME [-D flags]
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/spread_spectrum_watermarking_amd/csrc
mkdir -p "$R/build_tmp/variants"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" \
    -c "$C/dct_pair_f64.hip" -o "$R/build_tmp/variants/dct_pair_f64_$NAME.o"
OBJS=$(ls "$C"/*.o | grep -v -e '/dct_pair_f64.o' -e dct_folded -e dct_pair_f32.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/spread_spectrum_watermarking_amd/lib/libssw_$NAME.so" $OBJS "$R/build_tmp/variants/dct_pair_f64_$NAME.o"
echo built "libssw_$NAME.so"
