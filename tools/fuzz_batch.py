#!/usr/bin/env python3
"""Random frame shapes through ssw_batch_embed + ssw_batch_extract (f64): the pruned derived transform against the full one
(bit-identical), two lanes against one (bit-identical), and frame 0 against the oracle's pipeline.
tests/test_fuzz_gpu.py runs a fixed-seed leg of it in `pytest -m gpu`.
usage: python tools/fuzz_batch.py [N_SHAPES SEED]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def shapes(n_shapes, seed):
    """(h, w, n_frames, k, frame seed, mark seed): rows first, wide enough for the pruned path."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_shapes):
        step = int(rng.choice([8, 16, 32, 64, 128]))
        h = int(rng.integers(128 // step, 700 // step + 1)) * step
        w = int(rng.integers(max(h, 512) // step, 1400 // step + 1)) * step
        out.append((h, w, int(rng.integers(2, 6)), int(rng.integers(50, 400)), int(rng.integers(1, 1000)), int(rng.integers(0, 2 ** 31))))
    return out


def check(h, w, n, k, frame_seed, mark_seed):
    """One case: dict(same, marked_err, ext_err, ext_scale, sim_err, sim_scale)."""
    import gpu_util as G
    from oracle import oracle as O
    from spread_spectrum_watermarking_amd import _lib as L
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

    rgb = G.synth(frame_seed, 0, n, w, h)
    marks = np.random.default_rng(mark_seed).standard_normal((n, k)).astype(np.float32)
    a = run(rgb, marks, True, True, 2)
    b = run(rgb, marks, False, False, 2)
    same = all(np.array_equal(x, y) for x, y in zip(a, b))
    o_marked = O.embed_frame(rgb[0], marks[0])
    o_ext, o_sim = O.extract_frame(rgb[0], a[0][0], marks[0])
    return {"same": same, "marked_err": float(np.abs(a[0][0] - o_marked).max()),
            "ext_err": float(np.abs(a[2][0] - o_ext).max()), "ext_scale": max(1.0, float(np.abs(o_ext).max())),
            "sim_err": abs(float(a[3][0]) - o_sim), "sim_scale": max(1.0, abs(o_sim))}


def passes(r):
    return r["same"] and r["marked_err"] <= 2.4e-7 and r["ext_err"] <= 1e-5 * r["ext_scale"] and r["sim_err"] <= 1e-4 * r["sim_scale"]


if __name__ == "__main__":
    import gpu_util as G
    n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for case in shapes(n_shapes, int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        r = check(*case)
        h, w, n, k = case[:4]
        print(f"{h:5d} x {w:5d} n={n} k={k:3d} pruned+lanes == full+serial: {r['same']}; vs oracle: marked {r['marked_err']:.1e} "
              f"ext {r['ext_err']:.1e} sim {r['sim_err']:.1e}{'' if passes(r) else '   <-- FAIL'}")
        if not passes(r):
            sys.exit(1)
    print("all good; prune stats", G.ctx().prune_stats())
