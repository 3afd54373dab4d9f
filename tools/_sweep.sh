mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for f in 0 1 -; do
  if [ "$f" = "-" ]; then unset SSW_BN32; else export SSW_BN32=$f; fi
  python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-handle-leg --no-timers-off-leg --no-serial-leg > gpurun_out/p.json 2>/dev/null
  python - <<PY
import json
r=json.load(open("gpurun_out/p.json")); k=r["kernels"]
print("BN32=$f config1 value", r["value"], "rows", k["dct_rows"]["frac_mfma"], "cols", k["dct_cols"]["frac_mfma"], r["stage_ms_per_step"]["dct_row"], r["stage_ms_per_step"]["dct_col"])
PY
done
unset SSW_BN32
python tools/handle_bench.py 2>&1 | grep -E "u8 pinned|host thread"
