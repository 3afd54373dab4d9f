#!/usr/bin/env python3
"""Times ssw_dct2d alone (no correctness checks) with the library's hipEvent stage timers.
usage: dct_microbench.py [W H frames reps precision(f32|f64) type(0|2)]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check
a = sys.argv[1:]
W = int(a[0]) if len(a) > 0 else 3840; H = int(a[1]) if len(a) > 1 else 2160
n = int(a[2]) if len(a) > 2 else 16; reps = int(a[3]) if len(a) > 3 else 3
prec = L.PRECISION_F64 if (len(a) > 4 and a[4] == "f64") else L.PRECISION_F32
typ = int(a[5]) if len(a) > 5 else 0
ctx = wm.Context(0); lib = L.load(); ctx.set_chunk_frames(n)
if os.environ.get("SSW_NO_FOLD"): ctx.set_dct_folding(False)
if os.environ.get("SSW_FOLD_LEVEL"): lib.ssw_ctx_set_dct_folding(ctx.handle, int(os.environ["SSW_FOLD_LEVEL"]))
host = np.random.default_rng(0).random((n, H, W), dtype=np.float32)
buf = ctx.to_device(host)
check(lib.ssw_dct2d(ctx.handle, typ, prec, n, W, H, buf.ptr), "warm")
ctx.enable_timing(True); ctx.reset_timing()
for _ in range(reps):
    buf = ctx.to_device(host)          # fresh finite data each time (the transform is in place and unnormalised)
    check(lib.ssw_dct2d(ctx.handle, typ, prec, n, W, H, buf.ptr), "dct")
t = ctx.timing()
# executed fraction of the dense flop per axis (mirrors dct2d_planes): none 1, one folding level 1/2, two 3/8
level = 0 if os.environ.get("SSW_NO_FOLD") else int(os.environ.get("SSW_FOLD_LEVEL", L.DCT_FOLDING_DEFAULT))
def frac(length):
    if level == 0 or length % 8: return 1.0
    return 0.375 if (level >= 4 and length % 16 == 0 and length >= 64) else 0.5
rf = 2.0 * n * H * W * W * reps * frac(W); cf = 2.0 * n * W * H * H * reps * frac(H)
print("rows %.2f ms %.1f TF | cols %.2f ms %.1f TF (executed flop)" % (
    t["dct_row"]["ms"] / reps, rf / t["dct_row"]["ms"] / 1e9, t["dct_col"]["ms"] / reps, cf / t["dct_col"]["ms"] / 1e9))
