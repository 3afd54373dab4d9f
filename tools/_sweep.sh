mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t5.log 2>&1; tail -6 gpurun_out/t5.log
python tools/dct_microbench.py 3840 2160 64 2 f64 0
python tools/dct_microbench.py 3840 2160 64 2 f64 2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-handle-leg > gpurun_out/b5.json 2>gpurun_out/b5.err
python - <<PY
import json
r=json.load(open("gpurun_out/b5.json"))
k=r["kernels"]
print("value", r["value"], "rows", k["dct_rows"]["frac_mfma"], "cols", k["dct_cols"]["frac_mfma"], r["stage_ms_per_step"])
print("full", r["full_transform"]["value"], "serial", r["serialized"]["value"])
PY
