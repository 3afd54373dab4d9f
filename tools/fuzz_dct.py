#!/usr/bin/env python3
"""Random frame shapes (multiples of 8 and a few that are not) through ssw_dct2d in f64 against the CPU oracle's
correctly rounded transform: exercises every strategy branch of build_pass (deep / semi-deep / first-level split /
exact-operand folding / in-kernel folding / dense; rows first and columns first; class-major or natural planes).
usage: python tools/fuzz_dct.py [N_SHAPES SEED]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from oracle import oracle as O
from spread_spectrum_watermarking_amd import _lib as L

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 1.0
for t in range(n_shapes):
    step = int(rng.choice([8, 8, 16, 32, 64, 128, 4, 2]))
    h = int(rng.integers(128 // step, 1200 // step + 1)) * step
    w = int(rng.integers(128 // step, 1300 // step + 1)) * step
    n = int(rng.integers(1, 4))
    x = rng.random((n, h, w)).astype(np.float32)
    for kind, name in ((L.DCT2, "fwd"), (L.DCT2_ORTHOGONAL, "ortho"), (L.DCT3, "inv")):
        src = np.stack([O.dct2d(p, O.DCT2) for p in x]) if kind == L.DCT3 else x
        got = G.dct2d(src, kind, L.PRECISION_F64)
        ref = np.stack([O.dct2d(p, kind) for p in src])
        same = float(np.mean(got == ref))
        err = float(np.abs(got.astype(np.float64) - ref).max() / max(np.abs(ref[:, 1:, 1:]).max(), 1.0))
        worst = min(worst, same)
        flag = "" if same >= 0.998 and err <= 2e-7 else "   <-- FAIL"
        print(f"{h:5d} x {w:5d} n={n} {name:5s} identical {same:.6f} err/ACmax {err:.2e}{flag}")
        if flag:
            sys.exit(1)
print("all within the bars; worst identical fraction", worst)
