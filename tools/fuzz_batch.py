#!/usr/bin/env python3
"""Random frame shapes through ssw_batch_embed + ssw_batch_extract (f64): the pruned derived transform against the full one
(bit-identical), two lanes against one (bit-identical), and frame 0 against the oracle's pipeline.
usage: python tools/fuzz_batch.py [N_SHAPES SEED]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from oracle import oracle as O
from spread_spectrum_watermarking_amd import _lib as L

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = G.ctx()
cfg = G.default_config(L.PRECISION_F64)


def run(rgb, marks, overlap, prune, chunk):
    ctx.set_overlap(overlap); ctx.set_prune(prune); ctx.set_chunk_frames(chunk)
    try:
        res = G.batch_embed(rgb, marks, cfg, want_coef=False, want_idx=True)
        ext, sims = G.batch_extract(rgb, res["rgb"], marks.shape[1], marks, cfg)
        return res["rgb"], res["idx"], ext, sims
    finally:
        ctx.set_overlap(True); ctx.set_prune(True); ctx.set_chunk_frames(0)


for t in range(n_shapes):
    step = int(rng.choice([8, 16, 32, 64, 128]))
    h = int(rng.integers(128 // step, 700 // step + 1)) * step
    w = int(rng.integers(max(h, 512) // step, 1400 // step + 1)) * step        # rows first, wide enough for the pruned path
    n, k = int(rng.integers(2, 6)), int(rng.integers(50, 400))
    rgb = G.synth(int(rng.integers(1, 1000)), 0, n, w, h)
    marks = rng.standard_normal((n, k)).astype(np.float32)
    a = run(rgb, marks, True, True, 2)
    b = run(rgb, marks, False, False, 2)
    same = all(np.array_equal(x, y) for x, y in zip(a, b))
    o_marked = O.embed_frame(rgb[0], marks[0])
    o_ext, o_sim = O.extract_frame(rgb[0], a[0][0], marks[0])
    ok = (same and np.abs(a[0][0] - o_marked).max() <= 2.4e-7 and np.abs(a[2][0] - o_ext).max() <= 1e-5 * max(1.0, float(np.abs(o_ext).max()))
          and abs(float(a[3][0]) - o_sim) <= 1e-4 * max(1.0, abs(o_sim)))
    print(f"{h:5d} x {w:5d} n={n} k={k:3d} pruned+lanes == full+serial: {same}; vs oracle: marked {np.abs(a[0][0] - o_marked).max():.1e} "
          f"ext {np.abs(a[2][0] - o_ext).max():.1e} sim {abs(float(a[3][0]) - o_sim):.1e}{'' if ok else '   <-- FAIL'}")
    if not ok:
        sys.exit(1)
print("all good; prune stats", ctx.prune_stats())
