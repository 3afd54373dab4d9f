#!/usr/bin/env python3
"""Timeline of one pinned 8-bit embed + extract through the single-image handles (what the PCIe-inclusive rate is made of).
Run under rocprofv3 (kernel + memory-copy trace), then summarise:
    rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ht -o ht --output-format csv -- python3 tools/handle_trace.py run
    python3 tools/handle_trace.py summary /tmp/ht
The run prints host-side wall times of each call of the last repetition (perf_counter, us)."""
import csv
import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import spread_spectrum_watermarking_amd as wm
    from spread_spectrum_watermarking_amd.api import check
    W, H, K, REPS = 3840, 2160, 1000, 6
    ctx = wm.Context(0)
    d = ctx.alloc(H * W * 12)
    check(ctx._lib.ssw_synth_frames(ctx.handle, 1, 0, 1, W, H, d.ptr), "ssw_synth_frames")
    ctx.synchronize()
    rgb8 = np.floor(np.clip(d.to_host(np.float32, (H, W, 3)), 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)
    d.free()
    mark = np.random.default_rng(1).standard_normal(K).astype(np.float32)
    p_in, p_out = ctx.pinned_empty(rgb8.shape, np.uint8), ctx.pinned_empty(rgb8.shape, np.uint8)
    p_in[...] = rgb8
    for rep in range(REPS):
        t = [time.perf_counter()]
        wr = wm.Writer(p_in, ctx=ctx); t.append(time.perf_counter())
        marked = wr.mark_rgb8([mark], out=p_out); t.append(time.perf_counter())
        rb = wm.Reader.base(p_in, ctx=ctx); t.append(time.perf_counter())
        rd = wm.Reader.derived(marked, ctx); t.append(time.perf_counter())
        ext = rb.extract(rd, K); t.append(time.perf_counter())
        sim = wm.Tester(ext, ctx).similarity(mark).similarity; t.append(time.perf_counter())
    names = ["Writer::new", "mark_rgb8", "Reader::base", "Reader::derived", "extract", "similarity"]
    print("host wall of the last repetition (us):", {n: round((b - a) * 1e6) for n, a, b in zip(names, t, t[1:])},
          "total", round((t[-1] - t[0]) * 1e6), "sim %.3f" % sim)


def summary(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
    ev.sort()
    # the last repetition: events after the last gap of more than 300 us that precedes a large host-to-device copy ... simpler: the
    # last 1/6 of the events by count of 24.9-MB downloads
    big_d2h = [i for i, e in enumerate(ev) if e[2].startswith("C") and "DEVICE_TO_HOST" in e[2] and any(int(x) > 20e6 for x in e[2].split() if x.isdigit())]
    start = big_d2h[-2] + 1 if len(big_d2h) >= 2 else 0
    t0 = ev[start][0]
    for s, e, n in ev[start:]:
        print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summary":
        summary(sys.argv[2])
    else:
        run()
