#!/usr/bin/env python3
"""Time of one batch_extract's derived-frame row pass (stage timers) -- used with SSW_LIB_PATH to compare builds of
csrc/dct_pair_derived.hip.  usage: python tools/derived_time.py [FRAMES = 128]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gpu_util as G
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check
import ctypes as C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W, H, K = 3840, 2160, 1000
ctx = G.ctx()
lib = ctx._lib
rgb = ctx.alloc(N * H * W * 12)
check(lib.ssw_synth_frames(ctx.handle, 1, 0, N, W, H, rgb.ptr), "synth")
marks = ctx.to_device(np.random.default_rng(0).standard_normal((N, K)).astype(np.float32))
ext, sims = ctx.alloc(N * K * 4), ctx.alloc(N * 4)
cfg = G.default_config(L.PRECISION_F64)
ctx.set_overlap(False)
for rep in range(3):
    ctx.enable_timing(True); ctx.reset_timing()
    check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), rgb.ptr, rgb.ptr, N, W, H, K, ext.ptr, marks.ptr, sims.ptr), "extract")
    ctx.synchronize()
    t = ctx.timing(); ctx.enable_timing(False)
print({k: round(v["ms"], 2) for k, v in t.items() if v["ms"] > 0.05}, "(row stage = base frame's eight launches [~9.5 ms] + the derived frame's pass)")
