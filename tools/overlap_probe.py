#!/usr/bin/env python3
"""What a low-register streaming kernel gets when it runs BESIDE the basis GEMMs (r5 probe: is a pre-pass that fits the
registers two GEMM blocks leave free worth building?).  Forward transforms of 128 4K planes (ssw_dct2d, f64) on the context's
stream, alone and with a device-to-device copy loop (torch: vectorised elementwise copy, < 32 VGPRs) on a second stream; prints
the library's GEMM stage times and the copy's rate in both situations.
usage: python tools/overlap_probe.py [FRAMES = 128]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check

W, H = 3840, 2160
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = 3
ctx = wm.Context(0)
lib = ctx._lib
rgb = ctx.alloc(N * H * W * 12)
check(lib.ssw_synth_frames(ctx.handle, 1, 0, N, W, H, rgb.ptr), "synth")
y = ctx.alloc(N * H * W * 4)
check(lib.ssw_rgb_to_yiq(ctx.handle, rgb.ptr, N, W, H, y.ptr, None, None), "yiq")
rgb.free()
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GiB
b = torch.empty_like(a)
side = torch.cuda.Stream()


def transforms():
    ctx.enable_timing(True); ctx.reset_timing()
    for _ in range(R):
        check(lib.ssw_dct2d(ctx.handle, L.DCT2, L.PRECISION_F64, N, W, H, y.ptr), "dct")
        check(lib.ssw_dct2d(ctx.handle, L.DCT3, L.PRECISION_F64, N, W, H, y.ptr), "dct")
    ctx.synchronize()
    t = ctx.timing(); ctx.enable_timing(False)
    return {k: round(v["ms"] / R, 2) for k, v in t.items() if v["ms"] > 0}


def copies(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        for _ in range(n):
            b.copy_(a)
        e1.record()
    return e0, e1


check(lib.ssw_dct2d(ctx.handle, L.DCT2, L.PRECISION_F64, N, W, H, y.ptr), "dct")
check(lib.ssw_dct2d(ctx.handle, L.DCT3, L.PRECISION_F64, N, W, H, y.ptr), "dct")
ctx.synchronize()
e0, e1 = copies(20); torch.cuda.synchronize()
alone_copy = 20 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12
alone = transforms()
print(f"alone: copy {alone_copy:.2f} TB/s (read + write); transforms per forward + inverse (ms): {alone}")
# together: enough copies to outlast the transforms
t_tr = sum(v for k, v in alone.items() if k in ("dct_row", "dct_col", "dct_prep"))
n_copies = int(1.3 * R * t_tr * 1e-3 * alone_copy * 1e12 / (2 * a.numel() * 4)) + 4
e0, e1 = copies(n_copies)
t0 = time.perf_counter()
both = transforms()
wall = time.perf_counter() - t0
torch.cuda.synchronize()
both_copy = n_copies * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12
print(f"together: copy {both_copy:.2f} TB/s over {e0.elapsed_time(e1):.1f} ms ({n_copies} x 2 GiB); transforms (ms): {both}; host wall of the transforms {wall * 1e3:.1f} ms")
ga, gb = alone.get("dct_row", 0) + alone.get("dct_col", 0), both.get("dct_row", 0) + both.get("dct_col", 0)
print(f"GEMM stages {ga:.2f} -> {gb:.2f} ms ({gb / ga:.2f} x); bytes the copy moved while the transforms ran: ~{both_copy * R * sum(both.get(k, 0) for k in ('dct_row', 'dct_col', 'dct_prep')) * 1e-3 * 1e3:.0f} GB")
ctx.close()
