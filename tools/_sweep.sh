mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in 2 3; do
python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-handle-leg --no-full-transform-leg --no-timers-off-leg --no-serial-leg > gpurun_out/p.json 2>/dev/null
python - <<PY
import json
r=json.load(open("gpurun_out/p.json")); k=r["kernels"]
print("config $c value", r["value"], "rows", k["dct_rows"]["frac_mfma"], "cols", k["dct_cols"]["frac_mfma"], "col ms", r["stage_ms_per_step"]["dct_col"], "row ms", r["stage_ms_per_step"]["dct_row"])
PY
done
