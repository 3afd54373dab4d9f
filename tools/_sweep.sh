mkdir -p gpurun_out
for i in 1 2; do
for o in 0 1; do
  if [ $o = 1 ]; then export SSW_PREP_OLD=1; else unset SSW_PREP_OLD; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-handle-leg --no-full-transform-leg --no-timers-off-leg --no-serial-leg > gpurun_out/p.json 2>/dev/null
  python - <<PY
import json
r=json.load(open("gpurun_out/p.json")); k=r["kernels"]
print("old=$o value", r["value"], "rgb", k["rgb_to_yiq"], "prep", k["dct_prep"]["gbs"])
PY
done
done
