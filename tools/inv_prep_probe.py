#!/usr/bin/env python3
"""Stage times of one inverse transform (f64) over shapes of about one gigapixel: is the inverse row pre-pass's rate a function
of the row length?  usage: python tools/inv_prep_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check

ctx = wm.Context(0)
lib = ctx._lib
for (W, H) in ((3840, 2160), (5120, 2880), (6144, 3456), (7680, 4320), (7808, 4320), (7680, 2160), (3840, 4320), (8192, 4096), (4096, 2048)):
    N = max(1, (1 << 30) // (W * H))
    y = ctx.alloc(N * H * W * 4)
    check(lib.ssw_synth_frames(ctx.handle, 1, 0, max(1, N // 3), W, H, y.ptr), "synth")        # any finite data
    for kind, name in ((L.DCT2, "fwd"), (L.DCT3, "inv")):
        check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, N, W, H, y.ptr), "dct")
        ctx.enable_timing(True); ctx.reset_timing()
        for _ in range(2):
            check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, N, W, H, y.ptr), "dct")
        t = ctx.timing(); ctx.enable_timing(False)
        parts = {k: round(v["ms"] / 2, 2) for k, v in t.items() if v["ms"] > 0 and not k.endswith("_main")}
        gb = N * W * H * 12 / 1e9
        print(f"{W:5d} x {H:5d} n={N:4d} {name}: {parts}  (a pre-pass moves {gb:.1f} GB)", flush=True)
    y.free()
ctx.close()
