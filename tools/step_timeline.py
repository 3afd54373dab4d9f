#!/usr/bin/env python3
"""Where the GEMM stream idles: reads a rocprofv3 kernel trace of bench.py and prints, for the last steps of the run, the time
covered by `pair_gemm` launches, by every other kernel and by either, and every gap of more than 50 us between GEMM launches
with the kernels that ran inside it (HISTORY.md r6 section 12: the two-lane step idles its GEMM stream at the head of both
calls and at the end of the extract call).
usage (on the GPU box): cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python3 $REPO/bench.py --steps 3
           --warmup 2 --no-cpu-baseline --no-alt --no-timers-off-leg --no-handle-leg --no-full-transform-leg --no-serial-leg --no-stage-timers
       python3 tools/step_timeline.py /tmp/kt [window_ms = 410]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 410.0
f = d if d.endswith(".csv") else sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))


def short(name):
    name = name.replace("void ", "").replace("ssw::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:48]


def union(iv):
    iv = sorted(iv)
    if not iv:
        return 0
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


gemm = [k for k in ks if "pair_gemm" in k[2]]
end = gemm[-1][1]
start = end - int(win_ms * 1e6)
win = [k for k in ks if k[0] >= start and k[1] <= end]
g = sorted((k[0], k[1]) for k in win if "pair_gemm" in k[2])
h = [(k[0], k[1]) for k in win if "pair_gemm" not in k[2]]
print(f"window {win_ms:.0f} ms: GEMM launches cover {union(g) / 1e6:.1f} ms, other kernels {union(h) / 1e6:.1f} ms, either {union(g + h) / 1e6:.1f} ms")
gaps = [(e1, s2) for (s1, e1), (s2, e2) in zip(g, g[1:]) if s2 - e1 > 50_000]
print(f"{len(gaps)} gaps of more than 50 us between GEMM launches, {sum(b - a for a, b in gaps) / 1e6:.1f} ms in all")
for a, b in gaps:
    during = collections.Counter()
    for k in win:
        if "pair_gemm" in k[2]:
            continue
        ov = min(k[1], b) - max(k[0], a)
        if ov > 0:
            during[short(k[2])] += ov
    print(f"  at {(a - start) / 1e6:8.2f} ms, {(b - a) / 1e3:8.1f} us:", {k: round(v / 1e3) for k, v in during.most_common(4)})
