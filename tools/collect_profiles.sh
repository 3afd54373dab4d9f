#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/profiles_$TAG/ (copied to profiles/ afterwards):
#   bench JSON lines of the four BASELINE configs, rocprofv3 kernel-trace stats of the same commands, and the
#   PMC passes (HBM traffic, VALU / LDS activity) of a short 4K run -- separate --pmc passes, never combined with
#   trace domains other than the kernel trace (MI355X_MICROARCH.md, rocprofv3 section).
# usage (in the container, so that the commit is recorded): git rev-parse --short HEAD > .commit_stamp; gpurun -- bash tools/collect_profiles.sh r5
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
COMMIT=$(cat "$R/.commit_stamp" 2>/dev/null || echo unknown)
echo "$COMMIT" > "$OUT/commit.txt"

run_bench() {   # name, args...
    local name=$1; shift
    python3 "$R/bench.py" "$@" > "$OUT/${TAG}_bench_${name}.json" 2> "$OUT/${TAG}_bench_${name}.err" || echo "bench $name failed" >&2
    tail -c 300 "$OUT/${TAG}_bench_${name}.json"; echo
}
stats() {       # name, args...: rocprofv3 kernel trace + stats of the same command (short)
    local name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name --output-format csv -- python3 "$R/bench.py" "$@" > /dev/null 2>&1
    cp /tmp/prof_$name/${name}_kernel_stats.csv "$OUT/${TAG}_${name}_kernel_stats.csv" 2>/dev/null || find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_${name}_kernel_stats.csv" \;
}
pmc() {         # name, counters..., then "--" and bench args
    local name=$1; shift
    local counters=()
    while [ "$1" != "--" ]; do counters+=("$1"); shift; done
    shift
    rm -rf /tmp/pmc_$name
    rocprofv3 --kernel-trace --pmc "${counters[@]}" -d /tmp/pmc_$name -o $name --output-format csv -- python3 "$R/bench.py" "$@" > /dev/null 2>&1
    python3 "$R/tools/pmc_summary.py" /tmp/pmc_$name > "$OUT/${TAG}_pmc_${name}.txt" 2>&1
}

# 1. bench lines (driver-style steps for the headline, shorter for the rest)
run_bench config3_n1 --steps 20 --warmup 5
run_bench config2_n1 --config 2 --steps 10 --warmup 3
run_bench config1_n1 --config 1 --steps 20 --warmup 5
run_bench config4_n1 --config 4 --steps 3 --warmup 1
# 2. kernel stats of the same commands (one chunk-pair worth of steps)
SHORT="--no-cpu-baseline --no-alt --no-serial-leg --no-timers-off-leg --no-handle-leg --no-full-transform-leg"
stats config3 --steps 3 --warmup 1 $SHORT
stats config3_serial --steps 3 --warmup 1 --no-overlap $SHORT
stats config2 --config 2 --steps 3 --warmup 1 $SHORT
stats config1 --config 1 --steps 5 --warmup 2 $SHORT
stats config4 --config 4 --steps 1 --warmup 1 $SHORT
# 3. PMC passes on one serial 4K step of 128 frames (one pass)
PM="--batch 128 --steps 1 --warmup 1 --no-overlap $SHORT"      # one 128-frame pass: the automatic pass size of the default run
pmc fetch FETCH_SIZE -- $PM
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $PM
pmc valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $PM
pmc mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $PM
python3 "$R/tools/pmc_traffic_json.py" "$OUT" "$TAG" 3840 2160 128 "$COMMIT" 2 > "$OUT/${TAG}_pmc_traffic.json"
# the same two traffic passes on one 32-frame pass of configs[4] (8K, 8-bit): roofline.traffic of `--config 4`
PA="--config 4 --batch 32 --steps 1 --warmup 1 --no-overlap $SHORT"
mv "$OUT/${TAG}_pmc_fetch.txt" "$OUT/${TAG}_pmc_fetch_4k.txt"; mv "$OUT/${TAG}_pmc_write.txt" "$OUT/${TAG}_pmc_write_4k.txt"
pmc fetch FETCH_SIZE -- $PA
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $PA
python3 "$R/tools/pmc_traffic_json.py" "$OUT" "$TAG" 7680 4320 32 "$COMMIT" 2 > "$OUT/${TAG}_pmc_traffic_8k.json"
mv "$OUT/${TAG}_pmc_fetch.txt" "$OUT/${TAG}_pmc_fetch_8k.txt"; mv "$OUT/${TAG}_pmc_write.txt" "$OUT/${TAG}_pmc_write_8k.txt"
mv "$OUT/${TAG}_pmc_fetch_4k.txt" "$OUT/${TAG}_pmc_fetch.txt"; mv "$OUT/${TAG}_pmc_write_4k.txt" "$OUT/${TAG}_pmc_write.txt"
python3 "$R/tools/handle_bench.py" > "$OUT/${TAG}_handle_api.txt" 2>&1
python3 "$R/tools/resize_bench.py" > "$OUT/${TAG}_resize_bench.txt" 2>&1
python3 "$R/tools/sort_bench.py" > "$OUT/${TAG}_sort_bench.txt" 2>&1
python3 "$R/tools/select_bench.py" 7680 4320 32 > "$OUT/${TAG}_select_bench.txt" 2>&1
python3 "$R/tools/select_bench.py" 3840 2160 1 >> "$OUT/${TAG}_select_bench.txt" 2>&1
ls -la "$OUT"
