#!/usr/bin/env python3
"""r5: the level-2 row pre-pass in its co-resident form (csrc/dct_pair_prep_light.hip, tuning prep_light = 1) against
pair_prep16_rows_kernel (prep_light = 0): whole batch pipelines (coefficients, index lists, marked frames, extracted marks,
similarities) bit for bit, from f32 / 8-bit / 16-bit frames (writer: I and Q planes written; readers: not), in natural line
order and in the fused transform's unit order, frame heights that do and do not fill their last k-block of units, a
width whose last tile of units is partial, and plain planes through ssw_dct2d.  Small shapes take the level-2 kernels through
lowered thresholds.  tests/test_fuzz_gpu.py calls run() in-process.
usage: python tools/prep_light_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

# (h, w, frames, k): 4K x 8 fused (unit order), 4K x 4 unfused (natural order), 1080p, 8K, and small shapes (lowered thresholds)
BATCH = [(2160, 3840, 8, 1000), (2160, 3840, 4, 1000), (1080, 1920, 5, 500), (4320, 7680, 1, 2000), (272, 512, 40, 100), (144, 1088, 9, 64),
         (720, 1280, 48, 300)]
PLANES = [(2160, 3840, 2), (272, 576, 30), (1088, 2048, 9)]
# r5, the derived frame's pruned row pass in one kernel (csrc/dct_pair_derived.hip, tuning derived_fused) against pre-pass + launches:
# (h, w, frames, k) -- k <= 1024 takes the kernel; 64 .. 1024 covers class tiles of 4 .. 32 gathered columns
DERIVED = [(2160, 3840, 8, 1000), (2160, 3840, 3, 1024), (1080, 1920, 5, 500), (1080, 1920, 40, 64), (272, 512, 40, 100), (144, 1088, 9, 300),
           (4320, 7680, 1, 1000), (1088, 2048, 17, 785)]
INVERSE = [(2160, 3840, 3), (1080, 1920, 3), (272, 512, 30), (4320, 7680, 1), (144, 1280, 7), (2160, 3840, 1)]      # inv_prep_light: rows of a multiple of 128 columns
LOW = dict(efold_min=256, efold_inv_min=256, efold_cols_min=64)


def run(out=print, batch=BATCH, planes=PLANES, derived=DERIVED, inverse=INVERSE):
    import gpu_util as G
    from conftest import f32_to_u8
    from spread_spectrum_watermarking_amd import _lib as L, tuning
    bad = 0
    for (h, w, n, k) in batch:
        rgb = G.synth(5, 0, n, w, h)
        marks = np.random.default_rng(k).standard_normal((n, k)).astype(np.float32)
        u8 = f32_to_u8(rgb)
        u16 = np.floor(np.clip(rgb, 0, 1) * np.float32(65535) + np.float32(0.5)).astype(np.uint16)
        for kind in ("f32", "u8", "u16"):
            res = []
            for light in (1, 0):
                with tuning(prep_light=light, **LOW), G.fresh_ctx():
                    cfg = G.default_config(L.PRECISION_F64)
                    if kind == "f32":
                        r = G.batch_embed(rgb, marks, cfg, want_coef=True, want_idx=True)
                        ext, sims = G.batch_extract(rgb, r["rgb"], k, marks, cfg)
                        res.append((r["coef"], r["idx"], r["rgb"], ext, sims))
                    elif kind == "u8":
                        m = G.batch_embed_rgb8(u8, marks, cfg)
                        ext, sims = G.batch_extract_rgb8(u8, m, k, marks, cfg)
                        res.append((m, ext, sims))
                    else:
                        m = G.batch_embed_rgb16(u16, marks, cfg)
                        m16 = np.floor(np.clip(m, 0, 1) * np.float32(65535) + np.float32(0.5)).astype(np.uint16)
                        ext, sims = G.batch_extract_rgb16(u16, m16, k, marks, cfg)
                        res.append((m, ext, sims))
            same = all(np.array_equal(p, q) for p, q in zip(*res))
            bad += not same
            out(f"batch {h} x {w} n={n} k={k} {kind}: light == pair_prep16_rows_kernel: {same}{'' if same else '   <-- FAIL'}")
        del rgb, u8, u16
    for (h, w, n) in planes:
        x = np.random.default_rng(h + w).random((n, h, w)).astype(np.float32)
        res = []
        for light in (1, 0):
            with tuning(prep_light=light, **LOW), G.fresh_ctx():
                res.append(G.dct2d(x, L.DCT2, L.PRECISION_F64))
        same = np.array_equal(*res)
        bad += not same
        out(f"planes {h} x {w} n={n}: light == pair_prep16_rows_kernel: {same}{'' if same else '   <-- FAIL'}")
    for (h, w, n, k) in derived:
        rgb = G.synth(9, 0, n, w, h)
        marks = np.random.default_rng(k + 1).standard_normal((n, k)).astype(np.float32)
        cfg = G.default_config(L.PRECISION_F64)
        with tuning(**LOW), G.fresh_ctx():
            marked = G.batch_embed(rgb, marks, cfg)["rgb"]
        u8, m8 = f32_to_u8(rgb), f32_to_u8(marked)
        to16 = lambda a: np.floor(np.clip(a, 0, 1) * np.float32(65535) + np.float32(0.5)).astype(np.uint16)
        u16, m16 = to16(rgb), to16(marked)
        for kind in ("f32", "u8", "u16"):
            res = []
            for fused in (1, 0):
                with tuning(derived_fused=fused, merge_max_lines=256, **LOW), G.fresh_ctx() as c:      # (passes of more lines than that take the kernel)
                    if kind == "f32":
                        res.append(G.batch_extract(rgb, marked, k, marks, cfg))
                    elif kind == "u8":
                        res.append(G.batch_extract_rgb8(u8, m8, k, marks, cfg))
                    else:
                        res.append(G.batch_extract_rgb16(u16, m16, k, marks, cfg))
                    ps = c.prune_stats()
                    assert ps["redone_chunks"] == 0, ps
            same = all(np.array_equal(p, q) for p, q in zip(*res))
            bad += not same
            out(f"derived {h} x {w} n={n} k={k} {kind}: one kernel == pre-pass + launches (extracted marks, similarities): {same}{'' if same else '   <-- FAIL'}")
        del rgb, marked, u8, m8, u16, m16
    for (h, w, n) in inverse:
        x = np.random.default_rng(h * 3 + w).standard_normal((n, h, w)).astype(np.float32)
        res = []
        for light in (1, 0):
            with tuning(inv_prep_light=light, **LOW), G.fresh_ctx():
                res.append(G.dct2d(x, L.DCT3, L.PRECISION_F64))
        same = np.array_equal(*res)
        bad += not same
        out(f"inverse planes {h} x {w} n={n}: light inverse row pre-pass == prep16_inv_rows_l2_kernel: {same}{'' if same else '   <-- FAIL'}")
    out("light row pre-pass: " + ("FAILED" if bad else "all good"))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(lambda s: print(s, flush=True)) else 0)
