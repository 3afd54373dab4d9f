#!/usr/bin/env python3
"""Full coefficient ordering (Reader::indices() without a limit, src/algorithm.rs:200-210): the library's batched
radix sort on device-resident planes.  usage: sort_bench.py [W H FRAMES REPS]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check

W, H, N, REPS = (int(a) for a in (sys.argv[1:5] + ["3840", "2160", "8", "3"][len(sys.argv) - 1:]))
ctx = wm.Context(0)
lib = L.load()
planes = np.random.default_rng(0).standard_normal((N, H, W)).astype(np.float32)
d = ctx.to_device(planes)
k = W * H - 1
idx = ctx.alloc(N * k * 4)
check(lib.ssw_topk_indices(ctx.handle, d.ptr, N, W, H, L.ORDER_ENERGY, k, idx.ptr), "warm")
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    check(lib.ssw_topk_indices(ctx.handle, d.ptr, N, W, H, L.ORDER_ENERGY, k, idx.ptr), "sort")
ctx.synchronize()
dt = (time.perf_counter() - t0) / REPS
print(f"full order of {N} planes {W}x{H}: {dt * 1e3:.2f} ms per call, {N * k / dt / 1e9:.2f} G keys/s, "
      f"{N * k * 8 / dt / 1e9:.0f} GB/s of coefficients in + indices out (4 passes of 16 B/key inside: "
      f"{N * k * (8 + 4 * 32) / dt / 1e9:.0f} GB/s moved)")
ctx.close()
