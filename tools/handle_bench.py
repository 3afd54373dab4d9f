#!/usr/bin/env python3
"""PCIe-inclusive rate of the single-image handle API (host buffers in, host buffers out), one frame at a time:
Writer::new + mark, Reader::base + Reader::derived + extract, Tester::similarity -- 8-bit and f32 host frames,
pageable (through the pinned staging ring, swept over the number of copy threads) and pinned.
usage: handle_bench.py [W H REPS]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm

W, H, REPS = (int(a) for a in (sys.argv[1:4] + ["3840", "2160", "10"][len(sys.argv) - 1:]))
K = 1000
ctx = wm.Context(0)
rng = np.random.default_rng(1)
# the bench's synthetic frames (multi-octave value noise: a natural-like spectrum), made on the device
import ctypes as C
from spread_spectrum_watermarking_amd.api import check
_d = ctx.alloc(H * W * 12)
check(ctx._lib.ssw_synth_frames(ctx.handle, 1, 0, 1, W, H, _d.ptr), "ssw_synth_frames")
ctx.synchronize()
rgb8 = np.floor(np.clip(_d.to_host(np.float32, (H, W, 3)), 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)
rgb32 = rgb8.astype(np.float32) / np.float32(255)
_d.free()
mark = rng.standard_normal(K).astype(np.float32)
px = W * H / 1e6


def bench(name, img, u8_out, out_buf=None, fresh_out=False):
    def embed():
        wr = wm.Writer(img, ctx=ctx)
        o = None if fresh_out else out_buf
        return wr.mark_rgb8([mark], out=o) if u8_out else wr.mark([mark], out=o)

    def extract(marked):
        ext = wm.Reader.base(img, ctx=ctx).extract(wm.Reader.derived(marked, ctx), K)
        return wm.Tester(ext, ctx).similarity(mark).similarity
    marked = embed(); extract(marked)
    t0 = time.perf_counter()
    for _ in range(REPS):
        marked = embed()
    t1 = time.perf_counter()
    for _ in range(REPS):
        sim = extract(marked)
    t2 = time.perf_counter()
    print(f"{name:34s} embed {(t1 - t0) / REPS * 1e3:6.2f} ms ({px * REPS / (t1 - t0):6.0f} Mpix/s)  extract+sim "
          f"{(t2 - t1) / REPS * 1e3:6.2f} ms ({px * REPS / (t2 - t1):6.0f} Mpix/s)  embed+extract {px * REPS / (t2 - t0):6.0f} Mpix/s  sim {sim:.3f}")


print(f"{W}x{H} handles, host buffers, {REPS} reps, host has {os.cpu_count()} logical cores")
out8, out32 = np.empty_like(rgb8), np.empty_like(rgb32)
for thr in (1, 2, 4, 8, 16):
    ctx.set_copy_threads(thr)
    bench(f"u8 pageable, {thr} copy thread(s)", rgb8, True, out8)
ctx.set_copy_threads(0)
bench("u8 pageable, fresh output arrays", rgb8, True, None, fresh_out=True)
bench("f32 pageable", rgb32, False, out32)
p_in, p_out = ctx.pinned_empty(rgb8.shape, np.uint8), ctx.pinned_empty(rgb8.shape, np.uint8)
p_in[...] = rgb8
bench("u8 pinned", p_in, True, p_out)
p32_in, p32_out = ctx.pinned_empty(rgb32.shape, np.float32), ctx.pinned_empty(rgb32.shape, np.float32)
p32_in[...] = rgb32
bench("f32 pinned", p32_in, False, p32_out)
# two host threads, one context each (contexts are independent; ctypes releases the GIL inside the library): the
# transfers of one thread's image overlap the kernels of the other's
import threading


def worker(n, out):
    c = wm.Context(0)
    pi, po = c.pinned_empty(rgb8.shape, np.uint8), c.pinned_empty(rgb8.shape, np.uint8)
    pi[...] = rgb8
    for _ in range(2):                                   # warm-up: bases, staging ring, plane pool of this context
        m = wm.Writer(pi, ctx=c).mark_rgb8([mark], out=po)
        wm.Reader.base(pi, ctx=c).extract(wm.Reader.derived(m, c), K)
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(n):
        m = wm.Writer(pi, ctx=c).mark_rgb8([mark], out=po)
        e = wm.Reader.base(pi, ctx=c).extract(wm.Reader.derived(m, c), K)
        wm.Tester(e, c).similarity(mark)
    out.append(time.perf_counter() - t0)
    del pi, po
    c.close()


for n_thr in (1, 2, 3):
    res = []
    barrier = threading.Barrier(n_thr)
    th = [threading.Thread(target=worker, args=(REPS, res)) for _ in range(n_thr)]
    for t in th: t.start()
    for t in th: t.join()
    dt = max(res)
    print(f"{n_thr} host thread(s) x own context, u8 pinned: embed+extract {px * REPS * n_thr / dt:6.0f} Mpix/s")
ctx.enable_timing(True); ctx.reset_timing()
wr = wm.Writer(p_in, ctx=ctx); m = wr.mark_rgb8([mark], out=p_out)
b = wm.Reader.base(p_in, ctx=ctx); e = b.extract(wm.Reader.derived(m, ctx), K)
t = ctx.timing()
print("device stages of one embed + extract (ms):", {k: round(v["ms"], 3) for k, v in t.items() if v["launches"]})
del p_in, p_out, p32_in, p32_out, wr, b
ctx.close()
