mkdir -p gpurun_out
run() {
  python bench.py $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-handle-leg --no-full-transform-leg --no-timers-off-leg --no-serial-leg > gpurun_out/p.json 2>/dev/null
  python - <<PY
import json
r=json.load(open("gpurun_out/p.json")); k=r["kernels"]; s=r["stage_ms_per_step"]
print("$1 value", r["value"], "ms", r["ms_per_step"], "row", s["dct_row"], "col", s["dct_col"], "rgb", s["rgb_to_yiq"], "prep", s["dct_prep"], "sel", s["select"])
PY
}
EXTRA=""; run "chunk128 serial     "
EXTRA="--chunk 64"; run "chunk64 two lanes   "
EXTRA="--chunk 64 --no-overlap"; run "chunk64 one lane    "
export SSW_ONEBLOCK=60
EXTRA="--chunk 64"; run "chunk64 2 lanes 1blk "
export SSW_ONEBLOCK=44
EXTRA="--chunk 64"; run "chunk64 2 lanes 1blk44"
EXTRA="--chunk 43"; run "chunk43 2 lanes 1blk44"
