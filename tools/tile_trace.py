#!/usr/bin/env python3
"""Diagnostic (library built with -DSSW_TILE_TRACE): per-tile phase times of the operand-ready f64 GEMMs, and with --ksteps the
duration of every k-step of every tile's main loop, classified by what the other resident block of the same CU was doing.
Build: `make -C spread_spectrum_watermarking_amd/csrc && tools/build_variant.sh trace -DSSW_TILE_TRACE -DSSW_TILE_TRACE_FWD_ONLY`
(forward instances only, 3 min; without _FWD_ONLY the inverse instances trace too, 9 min), then run with
SSW_LIB_PATH=spread_spectrum_watermarking_amd/lib/libssw_trace.so.
usage: SSW_LIB_PATH=... python tools/tile_trace.py [W H FRAMES forward|inverse] [--ksteps]"""
import ctypes
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check

W, H, N = (int(a) for a in (sys.argv[1:4] + ["3840", "2160", "128"][len(sys.argv) - 1:]))
kind = L.DCT3 if (len(sys.argv) > 4 and sys.argv[4] == "inverse") else L.DCT2
ctx = wm.Context(0)
lib = ctx._lib
rgb = ctx.alloc(N * H * W * 12)
check(lib.ssw_synth_frames(ctx.handle, 1, 0, N, W, H, rgb.ptr), "synth")
y = ctx.alloc(N * H * W * 4)
check(lib.ssw_rgb_to_yiq(ctx.handle, rgb.ptr, N, W, H, y.ptr, None, None), "yiq")
rgb.free()
check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, N, W, H, y.ptr), "dct")      # warm-up (bases, workspaces)
ctx.synchronize()
CAP = 400000
buf = ctx.alloc(CAP * 64)
raw = lib
raw.ssw_debug_set_tile_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint]
raw.ssw_debug_get_tile_trace_count.argtypes = [ctypes.POINTER(ctypes.c_uint)]
assert raw.ssw_debug_set_tile_trace(buf.ptr, CAP) == 0
KSTEPS = "--ksteps" in sys.argv
if KSTEPS:
    kbuf = ctx.to_device(np.zeros(CAP * 32, np.uint32))
    raw.ssw_debug_set_tile_kstep.argtypes = [ctypes.c_void_p]
    assert raw.ssw_debug_set_tile_kstep(kbuf.ptr) == 0
check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, N, W, H, y.ptr), "dct")
ctx.synchronize()
n = ctypes.c_uint(0)
raw.ssw_debug_get_tile_trace_count(ctypes.byref(n))
cnt = min(n.value, CAP)
t = buf.to_host(np.uint64, (cnt, 8))
tag = (t[:, 5] & 0xFFFFFFFF).astype(np.int64)
cyc = ((t[:, 5] >> 36) & 0xFFFFFFF).astype(np.int64)
kp = (t[:, 6] >> 32).astype(np.int64)
npairs = (t[:, 6] & 0xFFFFFFFF).astype(np.int64)
hw = (t[:, 4] & 0xFFFFFFFF).astype(np.int64)
xcc = (t[:, 4] >> 32).astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"{cnt} tiles traced; distinct CU ids {len(set(cuid.tolist()))}")
groups = defaultdict(list)
for i in range(cnt):
    groups[(int(tag[i]), int(kp[i]), int(npairs[i]))].append(i)
for key, ids in sorted(groups.items(), key=lambda kv: t[kv[1][0], 0]):
    ids = np.array(ids)
    a = t[ids].astype(np.int64)
    pro, main, epi = (a[:, 1] - a[:, 0]) / 100.0, (a[:, 2] - a[:, 1]) / 100.0, (a[:, 3] - a[:, 2]) / 100.0   # us (100 MHz)
    issue = (a[:, 7] - a[:, 2]) / 100.0
    span = (a[:, 3].max() - a[:, 0].min()) / 100.0
    mhz = np.median(cyc[ids] / np.maximum((a[:, 3] - a[:, 0]) / 100.0, 1e-3))
    # per CU: time covered by 0 / 1 / 2+ blocks between first start and last end of the launch
    cov = np.zeros(3)
    gaps = []
    for c in set(cuid[ids].tolist()):
        sel = ids[cuid[ids] == c]
        ev = []
        for i in sel:
            ev.append((int(t[i, 1]), 1)); ev.append((int(t[i, 2]), -1))       # in the main loop
        ev.sort()
        level, last = 0, int(t[sel, 0].min())
        end = int(t[sel, 3].max())
        for when, d in ev:
            cov[min(level, 2)] += when - last
            last = when
            level += d
        cov[0] += end - last
    cov = cov / cov.sum()
    print(f"[{mhz:6.0f} MHz] tag {key[0]:4d} Kp {key[1]:5d} NP {key[2]:4d}: {len(ids):6d} tiles, launch {span / 1000:7.3f} ms | prologue {pro.mean():6.2f} us, "
          f"main {main.mean():7.2f} (min {main.min():7.2f}), epilogue {epi.mean():6.2f} (issued after {issue.mean():6.2f}) | CU time with 0 / 1 / 2+ blocks in the main loop: "
          f"{cov[0]:.3f} / {cov[1]:.3f} / {cov[2]:.3f}")

if KSTEPS:
    # per k-step durations of every tile, classified by what the OTHER blocks of the same CU were doing at the step's midpoint
    ks = kbuf.to_host(np.uint32, (cnt, 32)).astype(np.int64)
    T = t[:, :4].astype(np.int64)
    for key, ids in sorted(groups.items(), key=lambda kv: t[kv[1][0], 0]):
        ids = np.array(ids)
        nk = key[1] // 8
        nst = min(nk - 2, 32)                       # stamped steps: t = 0 .. nk - 3
        if nst < 3:
            continue
        acc = defaultdict(list)
        for c in set(cuid[ids].tolist()):
            sel = ids[cuid[ids] == c]
            if len(sel) < 2:
                continue
            for i in sel:
                hi = T[i, 1] & ~0xFFFFFFFF
                st = hi | ks[i, :nst]
                st = np.where(st < T[i, 1], st + (1 << 32), st)
                others = sel[sel != i]
                for j in range(nst - 1):
                    mid = (st[j] + st[j + 1]) // 2
                    n_main = int(((T[others, 1] <= mid) & (mid < T[others, 2])).sum())
                    n_epi = int(((T[others, 2] <= mid) & (mid < T[others, 3])).sum())
                    n_pro = int(((T[others, 0] <= mid) & (mid < T[others, 1])).sum())
                    cls = "partner main" if n_main else "partner epilogue" if n_epi else "partner prologue" if n_pro else "alone"
                    acc[cls].append((st[j + 1] - st[j]) / 100.0)
        line = f"tag {key[0]:4d} Kp {key[1]:5d} NP {key[2]:4d} k-step us:"
        for cls in ("partner main", "partner epilogue", "partner prologue", "alone"):
            v = np.array(acc.get(cls, []))
            if len(v):
                line += f" | {cls}: n {len(v)} mean {v.mean():.3f} p10 {np.percentile(v, 10):.3f} p50 {np.percentile(v, 50):.3f} p90 {np.percentile(v, 90):.3f}"
        print(line)
ctx.close()
