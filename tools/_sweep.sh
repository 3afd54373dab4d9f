mkdir -p gpurun_out
python tools/dct_microbench.py 3840 2160 64 2 f64 0
python tools/dct_microbench.py 3840 2160 64 2 f64 2
python tools/dct_microbench.py 1920 1080 128 2 f64 0
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-handle-leg --no-full-transform-leg --no-timers-off-leg --no-serial-leg > gpurun_out/b6.json 2>gpurun_out/b6.err
python - <<PY
import json
r=json.load(open("gpurun_out/b6.json"))
k=r["kernels"]
print("value", r["value"], "rows", k["dct_rows"]["frac_mfma"], "cols", k["dct_cols"]["frac_mfma"], r["roofline"]["frac"], r["stage_ms_per_step"])
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
