#!/usr/bin/env python3
"""r5: the FUSED forward transform (csrc/dct_pair_f64_kernel.hpp EPI_FWD_COLOP: the row GEMMs' epilogue applies the column
pre-pass's arithmetic to its accumulators and writes the column operands; no f32 plane between the passes,
src/dct2d.rs:152-168 stays the rounding point) and its mirror image for the inverse transform (EPI_INV_O_COLOP) against the unfused path (tuning fuse_cols = 0: row
launches -> f32 plane -> prep16_cols_l2_kernel / prep16_inv_cols_l2_kernel): planes of ssw_dct2d bit for bit, forward,
orthonormal and inverse, on shapes that cover every
tail mode of the row launches (16 / 32 / 48 / 64 pairs in the last tile column) and frames whose units do not fill their
last k-block; the first and the last frame also against the oracle's correctly rounded transform; then whole batch pipelines
(embed + extract) fused against unfused.  Small shapes take the level-2 kernels through lowered thresholds (ssw_tuning_set).
tests/test_fuzz_gpu.py calls run() in-process.
usage: python tools/fuse_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

# (h, w, frames): frames chosen so that both passes run on 128-line tiles (dct_pair_can_fuse_cols)
SHAPES = [(256, 256, 224), (272, 512, 150), (720, 1280, 46), (1088, 2048, 25), (2160, 3840, 8), (4320, 7680, 2),
          (2160, 3840, 1), (4320, 7680, 1), (2160, 3840, 2)]           # single frames: the eight classes of a pass in ONE launch
BATCH = [(720, 1280, 48, 300), (2160, 3840, 8, 1000)]           # (h, w, frames, k)
LOW = dict(efold_min=256, efold_inv_min=256, efold_cols_min=64)


def run(out=print, shapes=SHAPES, batch=BATCH):
    import fuzz_dct
    import gpu_util as G
    from oracle import oracle as O
    from spread_spectrum_watermarking_amd import _lib as L, tuning
    bad = 0
    for (h, w, n) in shapes:
        x = np.random.default_rng(h * 7 + w).random((n, h, w)).astype(np.float32)
        for name, kind in (("fwd", L.DCT2), ("ortho", L.DCT2_ORTHOGONAL), ("inv", L.DCT3)):
            preps = []
            for fuse in (1, 0):
                with tuning(fuse_cols=fuse, fuse_inv_cols=fuse, **LOW), G.fresh_ctx() as c:
                    if fuse:
                        planned = c.transform_plan(n, w, h, kind)["fused_cols"]
                    c.enable_timing(True)
                    c.reset_timing()
                    r = G.dct2d(x, kind, L.PRECISION_F64)
                    preps.append(c.timing()["dct_prep"]["launches"])
                    c.enable_timing(False)
                if fuse:
                    a = r
                else:
                    b = r
            # the fused transform has ONE pre-pass stage (rows), the unfused one two (rows, columns): the path under test ran
            # every forward case of the list must take the fused path; the inverse of one or two 4K frames stays unfused (its
            # dependent launches run one class each, on 64-line tiles) -- ssw_ctx_transform_plan says which, the stage count confirms
            same = bool(np.array_equal(a, b)) and preps == ([1, 2] if planned else [2, 2]) and (planned or kind == L.DCT3)
            worst = 1.0
            for f in (0, n - 1):
                worst = min(worst, float(np.mean(a[f] == O.dct2d(x[f], kind))))
            ok = same and worst >= fuzz_dct.BAR_IDENTICAL
            bad += not ok
            out(f"{h:5d} x {w:5d} n={n:3d} {name:5s} fused == unfused: {same}  identical to the oracle {worst:.6f}"
                f"{'' if ok else '   <-- FAIL (' + str(int(np.sum(a != b))) + ' coefficients differ; pre-pass stages ' + str(preps) + ')'}")
        del x
    for (h, w, n, k) in batch:
        rgb = G.synth(5, 0, n, w, h)
        marks = np.random.default_rng(k).standard_normal((n, k)).astype(np.float32)
        res = []
        for fuse in (1, 0):
            with tuning(fuse_cols=fuse, fuse_inv_cols=fuse, **LOW), G.fresh_ctx():
                cfg = G.default_config(L.PRECISION_F64)
                r = G.batch_embed(rgb, marks, cfg, want_coef=True, want_idx=True)
                ext, sims = G.batch_extract(rgb, r["rgb"], k, marks, cfg)
                res.append((r["coef"], r["idx"], r["rgb"], ext, sims))
        same = all(np.array_equal(p, q) for p, q in zip(*res))
        bad += not same
        out(f"batch {h} x {w} n={n} k={k}: fused == unfused (coefficients, indices, marked frames, extracted marks, similarities): {same}"
            f"{'' if same else '   <-- FAIL'}")
    out("fused forward transform: " + ("FAILED" if bad else "all good"))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(lambda s: print(s, flush=True)) else 0)
