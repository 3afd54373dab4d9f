#!/usr/bin/env python3
"""The level-2 row passes (forward: dct_pair_efold, inverse: dct_pair_efold_inv -- by default rows of 1280 columns or more)
and column passes (dct_pair_efold_cols: 720 rows or more) on SMALL shapes the oracle finishes in seconds: the thresholds
are lowered through ssw_tuning_set (efold_min = efold_inv_min = 256, efold_cols_min = 64) so that every row of a multiple of
64 (forward) / 256 (inverse) columns and every column of a multiple of 16 rows takes them; tests/test_fuzz_gpu.py calls
run() in-process.  Transforms against the oracle's correctly rounded one, and two batch pipelines (pruned + two lanes
against full + one lane, both against the oracle).
usage: python tools/level2_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import fuzz_batch  # noqa: E402
import fuzz_dct  # noqa: E402

# (h, w, frames): class-major tiles behind deep columns (h % 16 == 0), natural order behind others, one and several
# tiles of region sets per block of the inverse pre-pass (w / 64 = 4 .. 60), ragged last blocks and groups that straddle
# k-block pieces (w = 384, 640, 896, 1920: w / 64 % 4 == 2; w / 128 odd)
SHAPES = [(128, 256, 1), (128, 512, 2), (256, 768, 1), (136, 1024, 3), (480, 1280, 2), (512, 1536, 1), (100, 2048, 2),
          (1080, 256, 1), (64, 3840, 2), (272, 320, 2), (2160, 512, 1), (1104, 200, 2), (336, 132, 1), (72, 1920, 2), (1080, 1920, 1), (144, 384, 3), (80, 640, 1), (96, 896, 2)]
BATCH = [(256, 512, 3, 150, 11, 21), (144, 1024, 2, 200, 12, 22)]      # (h, w, frames, k, frame seed, mark seed)

def run(out=print):
    """All level-2 cases under lowered thresholds, on a fresh context; returns the number of failures."""
    import gpu_util as G
    from spread_spectrum_watermarking_amd import tuning
    bad = 0
    with tuning(efold_min=256, efold_inv_min=256, efold_cols_min=64), G.fresh_ctx():
        for (h, w, n) in SHAPES:
            for name in ("fwd", "ortho", "inv"):
                same, err = fuzz_dct.check(h, w, n, 1234 + h + w, name)
                ok = same >= fuzz_dct.BAR_IDENTICAL and err <= fuzz_dct.BAR_ERR
                bad += not ok
                out(f"{h:5d} x {w:5d} n={n} {name:5s} identical {same:.6f} err/ACmax {err:.2e}{'' if ok else '   <-- FAIL'}")
        for case in BATCH:
            r = fuzz_batch.check(*case)
            ok = r["same"] and fuzz_batch.passes(r)
            bad += not ok
            out(f"batch {case}: {r}{'' if ok else '   <-- FAIL'}")
    out("level-2 checks: " + ("FAILED" if bad else "all good"))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(lambda s: print(s, flush=True)) else 0)
