#!/usr/bin/env python3
"""Prints DCT / extraction error statistics of both GPU precisions against the CPU oracle
(f64 backend = correctly rounded transform; f32 backend = FFT-class f32 like rustdct).
Diagnostic for DESIGN.md's numerics section; run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_util as G
from oracle import oracle as O
from spread_spectrum_watermarking_amd import _lib as L
import spread_spectrum_watermarking_amd as wm

def stats(name, rgb, k):
    mark = np.random.default_rng(1).standard_normal(k).astype(np.float32)
    y, i, q = O.rgb_to_yiq(rgb)
    ref = O.dct2d(y, O.DCT2, O.BACKEND_F64)
    idx = O.indices(ref, k=k)
    ii = idx.astype(np.int64)
    dc = abs(float(ref.ravel()[0]))
    # oracle canonical pipeline with the fixed index list
    emb = O.embed(ref, idx, [mark]); yb = O.dct2d(emb, O.DCT3, O.BACKEND_F64); out = O.yiq_to_rgb(yb, i, q)
    cd = O.dct2d(O.rgb_to_yiq(out)[0], O.DCT2, O.BACKEND_F64)
    ext_ref = O.extract(ref, cd, idx, k)
    rows = []
    for label, getc in (("cpu_f32fft", lambda p: O.dct2d(p, O.DCT2, O.BACKEND_F32)),
                        ("gpu_f32", lambda p: G.dct2d(p, L.DCT2, L.PRECISION_F32)),
                        ("gpu_f64", lambda p: G.dct2d(p, L.DCT2, L.PRECISION_F64))):
        c = getc(y)
        err = np.abs(c.astype(np.float64) - ref)
        rel_top = np.abs((c.ravel()[ii].astype(np.float64) - ref.ravel()[ii]) / ref.ravel()[ii])
        cdd = getc(O.rgb_to_yiq(out)[0])
        ext = O.extract(c, cdd, idx, k)
        d = np.abs(ext - ext_ref)
        mism = int((O.indices(c, k=k) != idx).sum())
        print(f"{name:10s} {label:10s} coef err/DC max {err.max()/dc:.2e}  exact {np.mean(c==ref):.5f}  top-k rel err med {np.median(rel_top):.2e} max {rel_top.max():.2e}"
              f"  | extracted err med {np.median(d):.2e} max {d.max():.2e}  | rank mismatches {mism}/{k}"
              f"  | sim delta {abs(O.similarity(ext, mark)-O.similarity(ext_ref, mark)):.2e}")

if __name__ == "__main__":
    g = np.load(os.path.join(ROOT, "tests/golden/cat_decoded_u8.npz"))
    stats("cat640x444", g["cat"].astype(np.float32) / np.float32(255), 1000)
    stats("synth1080p", G.synth(1, 0, 1, 1920, 1080)[0], 1000)
    stats("synth4K", G.synth(1, 0, 1, 3840, 2160)[0], 1000)
    if len(sys.argv) > 1 and sys.argv[1] == "8k":
        stats("synth8K", G.synth(1, 0, 1, 7680, 4320)[0], 10000)
