mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 > gpurun_out/b11.json 2> gpurun_out/b11.err; tail -3 gpurun_out/b11.err
python - <<PY
import json
r=json.load(open("gpurun_out/b11.json"))
print(r["value"], r["handle_api"]["rgb8_pinned_two_threads"], {k:(v["embed_extract_mpix_s"]) for k,v in r["handle_api"].items() if isinstance(v,dict) and "embed_ms" in v})
print(r["cpu_baseline"]["value"], r["parity"]["frames"][0]["extracted_max_abs_diff_vs_cpu_exact"])
PY
