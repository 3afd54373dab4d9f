#!/usr/bin/env python3
"""PCIe-inclusive rate of the single-image handle API (host buffers in, host buffers out), one frame at a time:
Writer::new + mark, Reader::base + Reader::derived + extract, Tester::similarity.  usage: handle_bench.py [W H REPS]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm

W, H, REPS = (int(a) for a in (sys.argv[1:4] + ["3840", "2160", "5"][len(sys.argv) - 1:]))
ctx = wm.Context(0)
rng = np.random.default_rng(1)
rgb = rng.random((H, W, 3), dtype=np.float32)
mark = rng.standard_normal(1000).astype(np.float32)


def embed():
    return wm.Writer(rgb, ctx=ctx).mark([mark])


def extract(marked):
    ext = wm.Reader.base(rgb, ctx=ctx).extract(wm.Reader.derived(marked, ctx), 1000)
    return wm.Tester(ext, ctx).similarity(mark).similarity


marked = embed(); extract(marked)
t0 = time.perf_counter()
for _ in range(REPS):
    marked = embed()
t1 = time.perf_counter()
for _ in range(REPS):
    sim = extract(marked)
t2 = time.perf_counter()
px = W * H / 1e6
print(f"{W}x{H} handles, host buffers: embed {(t1 - t0) / REPS * 1e3:.1f} ms/frame ({px * REPS / (t1 - t0):.0f} Mpix/s), "
      f"extract+similarity {(t2 - t1) / REPS * 1e3:.1f} ms/frame ({px * REPS / (t2 - t1):.0f} Mpix/s), "
      f"embed+extract {px * REPS / (t2 - t0):.0f} Mpix/s; sim {sim:.3f}")
ctx.close()
