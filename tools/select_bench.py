#!/usr/bin/env python3
"""Times ssw_topk_indices (sample + threshold + compaction + finish) for several mark lengths on coefficient planes
of synthetic frames.  usage: python tools/select_bench.py [W H FRAMES [K ...]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check

W, H, N = (int(a) for a in (sys.argv[1:4] + ["7680", "4320", "8"][len(sys.argv) - 1:]))
KS = [int(a) for a in sys.argv[4:]] or [500, 1000, 1024, 1025, 2000, 4000, 8000, 10000, 16384]
ctx = wm.Context(0)
lib = ctx._lib
rgb = ctx.alloc(N * H * W * 12)
check(lib.ssw_synth_frames(ctx.handle, 1, 0, N, W, H, rgb.ptr), "synth")
y = ctx.alloc(N * H * W * 4)
check(lib.ssw_rgb_to_yiq(ctx.handle, rgb.ptr, N, W, H, y.ptr, None, None), "yiq")
rgb.free()
check(lib.ssw_dct2d(ctx.handle, L.DCT2, L.PRECISION_F64, N, W, H, y.ptr), "dct")
idx = ctx.alloc(N * 16384 * 4)
for k in KS:
    check(lib.ssw_topk_indices(ctx.handle, y.ptr, N, W, H, L.ORDER_ENERGY, k, idx.ptr), "topk")
    ctx.enable_timing(True); ctx.reset_timing()
    for _ in range(5):
        check(lib.ssw_topk_indices(ctx.handle, y.ptr, N, W, H, L.ORDER_ENERGY, k, idx.ptr), "topk")
    t = ctx.timing()["select"]; ctx.enable_timing(False)
    print(f"k={k:6d}: {t['ms'] / 5:.3f} ms per {N} frames {W}x{H}  ({t['work'] / 5 / (t['ms'] / 5) / 1e6:.0f} GB/s algorithmic)")
ctx.close()
