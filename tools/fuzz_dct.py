#!/usr/bin/env python3
"""Random frame shapes (multiples of 8 and a few that are not) through ssw_dct2d in f64 against the CPU oracle's
correctly rounded transform: exercises every strategy branch of build_pass (deep / semi-deep / first-level split /
exact-operand folding / in-kernel folding / dense; rows first and columns first; class-major or natural planes).
tests/test_fuzz_gpu.py runs a fixed-seed leg of it in `pytest -m gpu`.
usage: python tools/fuzz_dct.py [N_SHAPES SEED]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

BAR_IDENTICAL, BAR_ERR = 0.998, 2e-7          # fraction of coefficients bit-identical to the oracle; max error / AC max


def shapes(n_shapes, seed):
    """(h, w, n_frames, data seed) of the random cases: steps of 8 .. 128 and a few that are not multiples of 8."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_shapes):
        step = int(rng.choice([8, 8, 16, 32, 64, 128, 4, 2]))
        h = int(rng.integers(128 // step, 1200 // step + 1)) * step
        w = int(rng.integers(128 // step, 1300 // step + 1)) * step
        out.append((h, w, int(rng.integers(1, 4)), int(rng.integers(0, 2 ** 31))))
    return out


def check(h, w, n, data_seed, kind_name):
    """One case: (identical fraction, err / AC max) of ssw_dct2d against the oracle."""
    import gpu_util as G
    from oracle import oracle as O
    from spread_spectrum_watermarking_amd import _lib as L
    kind = {"fwd": L.DCT2, "ortho": L.DCT2_ORTHOGONAL, "inv": L.DCT3}[kind_name]
    x = np.random.default_rng(data_seed).random((n, h, w)).astype(np.float32)
    src = np.stack([O.dct2d(p, O.DCT2) for p in x]) if kind == L.DCT3 else x
    got = G.dct2d(src, kind, L.PRECISION_F64)
    ref = np.stack([O.dct2d(p, kind) for p in src])
    same = float(np.mean(got == ref))
    err = float(np.abs(got.astype(np.float64) - ref).max() / max(np.abs(ref[:, 1:, 1:]).max(), 1.0))
    return same, err


if __name__ == "__main__":
    n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    worst = 1.0
    for (h, w, n, ds) in shapes(n_shapes, int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        for name in ("fwd", "ortho", "inv"):
            same, err = check(h, w, n, ds, name)
            worst = min(worst, same)
            flag = "" if same >= BAR_IDENTICAL and err <= BAR_ERR else "   <-- FAIL"
            print(f"{h:5d} x {w:5d} n={n} {name:5s} identical {same:.6f} err/ACmax {err:.2e}{flag}")
            if flag:
                sys.exit(1)
    print("all within the bars; worst identical fraction", worst)
