mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t10.log 2>&1; tail -4 gpurun_out/t10.log
run() {
  python bench.py $EXTRA --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-handle-leg --no-full-transform-leg --no-timers-off-leg --no-serial-leg > gpurun_out/p.json 2>/dev/null
  python - <<PY
import json
r=json.load(open("gpurun_out/p.json")); k=r["kernels"]
print("$1 value", r["value"], "rows", k["dct_rows"]["frac_mfma"], "cols", k["dct_cols"]["frac_mfma"], "prep", r["stage_ms_per_step"]["dct_prep"], "row ms", r["stage_ms_per_step"]["dct_row"], "col ms", r["stage_ms_per_step"]["dct_col"])
PY
}
for rep in 1 2; do
EXTRA=""
SSW_TPERM=1 run "tperm wide  "
SSW_TPERM=0 run "natural wide"
SSW_TPERM=0 SSW_PREP_NARROW=1 run "natural old "
done
EXTRA="--config 2"
SSW_TPERM=1 run "1080p tperm wide  "
SSW_TPERM=0 SSW_PREP_NARROW=1 run "1080p natural old "
