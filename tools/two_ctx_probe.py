#!/usr/bin/env python3
"""Probe: do the GEMM stream's idle gaps of a step (HISTORY r6 section 12) close when the NEXT step runs beside them?  Two contexts on one
GPU, each with its own lanes and streams; step i = ssw_batch_embed + ssw_batch_extract of the same resident frames on context i % 2,
no host wait between steps.  Prints Mpix/s for one context (the bench's step) and for two.
usage: python tools/two_ctx_probe.py [FRAMES = 256] [STEPS = 8]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, K = 3840, 2160, 1000
ctxs = [wm.Context(0), wm.Context(0)]
lib = ctxs[0]._lib
rgb = ctxs[0].alloc(N * H * W * 12)
check(lib.ssw_synth_frames(ctxs[0].handle, 1, 0, N, W, H, rgb.ptr), "synth")
marks = ctxs[0].to_device(np.random.default_rng(0).standard_normal((N, K)).astype(np.float32))
ctxs[0].synchronize()
cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64)
bufs = []
for c in ctxs:
    bufs.append((c.alloc(N * H * W * 12), c.alloc(N * K * 4), c.alloc(N * 4)))


def step(i, c):
    out, ext, sims = bufs[i]
    check(lib.ssw_batch_embed(c.handle, C.byref(cfg), rgb.ptr, N, W, H, marks.ptr, K, out.ptr, None, None), "embed")
    check(lib.ssw_batch_extract(c.handle, C.byref(cfg), rgb.ptr, out.ptr, N, W, H, K, ext.ptr, marks.ptr, sims.ptr), "extract")


for mode in (1, 2, 1, 2):
    for i in range(2):
        step(i % mode, ctxs[i % mode])
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        step(i % mode, ctxs[i % mode])
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    s0 = bufs[0][2].to_host(np.float32, (N,))
    print(f"{mode} context(s): {STEPS * N * W * H / dt / 1e6:.0f} Mpix/s, {dt / STEPS * 1e3:.2f} ms per step, pass frames {ctxs[mode - 1].pass_frames(N, W, H) if hasattr(ctxs[0], 'pass_frames') else '?'}, sim[0] {s0[0]:.4f}")
