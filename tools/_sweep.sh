mkdir -p gpurun_out
S=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b12.json 2> gpurun_out/b12.err
E=$(date +%s.%N); echo "bench wall seconds: $(echo "$E - $S" | bc)"
python - <<PY
import json
r=json.load(open("gpurun_out/b12.json")); print(r["value"], r["kernels"]["dct_cols"]["frac_mfma"], r["handle_api"]["rgb8_pinned"]["embed_extract_mpix_s"], r["full_transform"]["value"])
PY
