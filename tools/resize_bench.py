#!/usr/bin/env python3
"""Times ssw_resize_rgb8 per direction (hipEvent stage timer of the library) on device-resident 8-bit frames.
usage: python tools/resize_bench.py [W H FRAMES REPS]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd.api import check

W, H, N, REPS = (int(a) for a in (sys.argv[1:5] + ["3840", "2160", "32", "5"][len(sys.argv) - 1:]))
ctx = wm.Context(0)
lib = ctx._lib
rgb = ctx.alloc(N * H * W * 12)
check(lib.ssw_synth_frames(ctx.handle, 1, 0, N, W, H, rgb.ptr), "synth")
big = ctx.alloc(N * H * W * 3)
check(lib.ssw_convert_f32_to_rgb8(ctx.handle, rgb.ptr, N * H * W * 3, big.ptr), "to_u8")
rgb.free()
small = ctx.alloc(N * (H // 8) * (W // 8) * 3)
back = ctx.alloc(N * H * W * 3)
for name, (src, sw, sh, dst, dw, dh) in {"down /8": (big, W, H, small, W // 8, H // 8), "up x8": (small, W // 8, H // 8, back, W, H),
                                         "down /2": (big, W, H, back, W // 2, H // 2)}.items():
    check(lib.ssw_resize_rgb8(ctx.handle, src.ptr, N, sw, sh, dw, dh, dst.ptr), name)
    ctx.enable_timing(True)
    ctx.reset_timing()
    for _ in range(REPS):
        check(lib.ssw_resize_rgb8(ctx.handle, src.ptr, N, sw, sh, dw, dh, dst.ptr), name)
    t = ctx.timing()["resize"]
    ctx.enable_timing(False)
    ms = t["ms"] / REPS
    print(f"{name}: {ms:.3f} ms per {N} frames {sw}x{sh}->{dw}x{dh}: {t['work'] / REPS / ms / 1e6:.0f} GB/s algorithmic "
          f"({t['work'] / REPS / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")
ctx.close()
