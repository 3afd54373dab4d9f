#!/usr/bin/env python3
"""Prints the numbers table of DESIGN.md section 6.3 from the committed bench lines profiles/<tag>_bench_config*_n1.json.
usage: python tools/design_numbers.py [tag = r5]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
names = {3: "3: 256 × 4K, k = 1000 (the metric's configuration)", 2: "2: 256 × 1080p", 1: "1: one 4K frame, embed + IDCT round trip",
         4: "4: 64 × 8K 8-bit, k = 10000, ⅛ resize attack"}
print("| config | Mpix/s | ms / step | `roofline_step` frac (bound) | `roofline` GEMM family two lanes / one | `roofline_hbm` pre-passes two lanes / one | one lane Mpix/s |")
print("|---|---|---|---|---|---|---|")
for c in (3, 2, 1, 4):
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_config{c}_n1.json")))
    s = d.get("serialized", {})
    rs = d["roofline_step"]
    print(f"| {names[c]} | **{d['value']:.0f}** | {d['ms_per_step']:.3f} | {rs['frac']:.2f} ({rs['bound']}: {rs['algorithmic_bytes_per_step'] / 1e9:.0f} GB, "
          f"{rs['executed_flop_per_step'] / 1e12:.2f} TF) | {d['roofline']['frac']:.2f} / {s.get('roofline', {}).get('frac', float('nan')):.2f} | "
          f"{d['roofline_hbm']['frac']:.2f} / {s.get('roofline_hbm', {}).get('frac', float('nan')):.2f} | {s.get('value', float('nan')):.0f} |")
d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_config3_n1.json")))
h = d["handle_api"]
print()
print(f"configs[3] legs: `full_transform` (prune off, four full transforms) {d['full_transform']['value']:.0f}; timers off {d['timers_off']['value']:.0f}; "
      f"best launch {d['roofline']['best_launch']['frac']:.2f} of the MFMA peak ({d['roofline']['best_launch']['avg_ms']:.3f} ms); "
      f"serialized stages (ms per step): " + ", ".join(f"{k} {v:.1f}" for k, v in d["serialized"]["stage_ms_per_step"].items() if v >= 0.05 and not k.endswith("_main")) + ".")
print(f"`handle_api` (4K 8-bit host images, PCIe included): one image per call pinned {h['rgb8_pinned']['embed_extract_mpix_s']:.0f} Mpix/s "
      f"(pageable {h['rgb8_pageable']['embed_extract_mpix_s']:.0f}, two host threads {h.get('rgb8_pinned_two_threads', {}).get('embed_extract_mpix_s', 0):.0f}); "
      f"streaming 64 pinned images per call {h['rgb8_pinned_stream']['embed_extract_mpix_s']:.0f}, pageable {h['rgb8_pageable_stream']['embed_extract_mpix_s']:.0f}; "
      f"link {h['pinned_link_gbs']}.")
c = d["cpu_baseline"]
p = d["parity"]
print(f"`cpu_baseline`: {c['value']} Mpix/s on {c['cores']} core ({c['kind']}; {c['sample'].split(',')[-1].strip()}); frame-parallel {d.get('cpu_baseline_parallel', {}).get('value')} Mpix/s on "
      f"{d.get('cpu_baseline_parallel', {}).get('cores')} threads.  `parity` (frames 0 and {p['frames'][-1]['frame']}): sim delta "
      f"{max(f['sim_delta_vs_cpu_exact'] for f in p['frames']):.1e}, extracted max |diff| {max(f['extracted_max_abs_diff_vs_cpu_exact'] for f in p['frames']):.1e}, "
      f"marked frame bit-identical {min(f['marked_frame_bit_identical_fraction'] for f in p['frames']):.5f}; against the f32-FFT stand-in for rustdct: "
      f"{p['extracted_max_abs_diff_vs_cpu_f32fft']:.1e}.")
try:
    pm = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")))
    fam = pm["families"]
    print(f"PMC (`profiles/{tag}_pmc_traffic.json`, one serial 128-frame 4K step, commit {pm['commit']}): GEMM family {fam['gemm']['hbm_bytes_per_step'] / 1e9:.1f} GB = "
          f"{fam['gemm']['traffic_over_algorithmic']} × its algorithmic bytes; pre-pass family {fam['prepass']['hbm_bytes_per_step'] / 1e9:.1f} GB = {fam['prepass']['traffic_over_algorithmic']} ×; per kernel: "
          + "; ".join(f"{k.split('<', 1)[1][:-1] if '<' in k else k} {v['traffic_over_algorithmic']}" + (f" (MFMA busy {v['mfma_busy_over_active_cycles']:.2f})" if v.get('mfma_busy_over_active_cycles') else "")
                      for k, v in pm["kernels"].items() if "gemm" in k) + ".")
except OSError:
    pass
