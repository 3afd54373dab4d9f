#!/bin/bash
# Diagnostic variants of the library: the GEMM translation units recompiled with extra -D flags and linked with the normal objects
# of csrc/ into spread_spectrum_watermarking_amd/lib/libssw_<name>.so (git-ignored; travels with gpurun; load with SSW_LIB_PATH).
# Run `make -C csrc` first.  Timing-only ablations compute wrong values by design.
#   tools/build_variant.sh trace -DSSW_TILE_TRACE -DSSW_TILE_TRACE_FWD_ONLY     per-tile / per-k-step stamps of the forward instances (tools/tile_trace.py)
#   tools/build_variant.sh x0 -DSSW_ABL_X0                                      every block stages tile 0's lines (L2-resident operands)
#   ALL=1 tools/build_variant.sh regs -DSSW_GEMM_DMA=0                          forward AND inverse units (five compiles in parallel)
# usage: [ALL=1] tools/build_variant.sh NAME [-D flags]
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/spread_spectrum_watermarking_amd/csrc
D=$R/build_tmp/variants/$NAME
mkdir -p "$D"
UNITS="dct_pair_f64"
[ -n "$ALL" ] && UNITS="dct_pair_f64 dct_pair_f64_inv dct_pair_f64_inv2 dct_pair_f64_inv3 dct_pair_f64_inv4"
for u in $UNITS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" -c "$C/$u.hip" -o "$D/$u.o" &
done
wait
EXCL="-e dct_folded -e dct_pair_f32.o"
for u in $UNITS; do EXCL="$EXCL -e /$u.o"; done
OBJS=$(ls "$C"/*.o | grep -v $EXCL)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/spread_spectrum_watermarking_amd/lib/libssw_$NAME.so" $OBJS "$D"/*.o
echo built "libssw_$NAME.so"
