#!/usr/bin/env python3
"""Times ssw_dct2d (forward and inverse, f64) on a batch of planes and prints the library's stage timers.
usage: python tools/dct_bench.py [W H FRAMES REPS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check

W, H, N, R = (int(a) for a in (sys.argv[1:5] + ["3840", "2160", "128", "3"][len(sys.argv) - 1:]))
ctx = wm.Context(0)
lib = ctx._lib
rgb = ctx.alloc(N * H * W * 12)
check(lib.ssw_synth_frames(ctx.handle, 1, 0, N, W, H, rgb.ptr), "synth")
y = ctx.alloc(N * H * W * 4)
check(lib.ssw_rgb_to_yiq(ctx.handle, rgb.ptr, N, W, H, y.ptr, None, None), "yiq")
rgb.free()
for kind, name in ((L.DCT2, "forward"), (L.DCT3, "inverse")):
    check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, N, W, H, y.ptr), "dct")
    ctx.enable_timing(True); ctx.reset_timing()
    for _ in range(R):
        check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, N, W, H, y.ptr), "dct")
    t = ctx.timing(); ctx.enable_timing(False)
    parts = {k: round(v["ms"] / R, 3) for k, v in t.items() if v["ms"] > 0}
    tot = sum(v for k, v in parts.items() if k in ("dct_row", "dct_col", "dct_prep", "rgb_to_yiq"))
    print(f"{name} {N} x {W}x{H}: {tot:.2f} ms  {parts}")
ctx.close()
