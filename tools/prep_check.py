#!/usr/bin/env python3
"""A/B check of the LDS-staged deep pre-passes (csrc/dct_pair_prep_staged.hip, class-major tiles) against the r3 kernels
(SSW_PREP_STAGED=0 SSW_CLASS_TILE=0): the same operations per operand element, so forward and inverse transforms of the
same planes must agree bit for bit.  Each variant runs in its own process (the switches are read once).  All variants run
at LEVEL 1 (SSW_EFOLD_MIN / SSW_EFOLD_INV_MIN / SSW_EFOLD_COLS_MIN out of reach): the level-2 passes (r4b / r4c) exist in the
staged form only and round differently (one more rotation); tools/level2_check.py holds them against the oracle.
usage: python tools/prep_check.py            # compare
       python tools/prep_check.py dump       # one variant: sha256 of every result (internal)"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (W, H, frames): deep both passes with class-major tiles; semi-deep columns (natural order); 8K; small tiles; one-tile
# class-major (W % 128 != 0); natural order with a short last column tile; tall frames (column pass first)
SHAPES = [(3840, 2160, 3), (1920, 1080, 4), (7680, 4320, 1), (1024, 512, 2), (256, 256, 5), (320, 256, 3),
          (1000, 1088, 2), (520, 264, 3), (512, 1024, 2), (384, 272, 2), (2048, 1040, 1)]


def dump():
    import numpy as np
    import spread_spectrum_watermarking_amd as wm
    from spread_spectrum_watermarking_amd import _lib as L
    from spread_spectrum_watermarking_amd.api import check
    ctx = wm.Context(0)
    lib = ctx._lib
    out = {}
    for (w, h, n) in SHAPES:
        rgb = ctx.alloc(n * h * w * 12)
        check(lib.ssw_synth_frames(ctx.handle, 7, 0, n, w, h, rgb.ptr), "synth")
        y = ctx.alloc(n * h * w * 4)
        check(lib.ssw_rgb_to_yiq(ctx.handle, rgb.ptr, n, w, h, y.ptr, None, None), "yiq")
        rgb.free()
        for kind, name in ((L.DCT2, "fwd"), (L.DCT3, "inv")):
            check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, n, w, h, y.ptr), "dct")
            host = y.to_host(np.float32, (n * h * w,))
            out[f"{w}x{h}x{n}:{name}"] = hashlib.sha256(host.tobytes()).hexdigest()
            if not np.isfinite(host).all():
                out[f"{w}x{h}x{n}:{name}"] += " NONFINITE"
        y.free()
    ctx.close()
    print("PREPCHECK " + json.dumps(out))


def run_variant(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "dump"], env=env, capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("PREPCHECK "):
            return json.loads(line[len("PREPCHECK "):])
    raise SystemExit(f"variant {env_extra} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dump":
        dump()
        sys.exit(0)
    l1 = {"SSW_EFOLD_MIN": "1000000", "SSW_EFOLD_INV_MIN": "1000000", "SSW_EFOLD_COLS_MIN": "1000000"}
    new = run_variant(dict(l1))
    old = run_variant(dict(l1, SSW_PREP_STAGED="0", SSW_CLASS_TILE="0"))
    mid = run_variant(dict(l1, SSW_PREP_STAGED="0"))          # r3 kernels on the tiled class-major order
    bad = 0
    for k in new:
        flag = "ok" if new[k] == old[k] == mid[k] else "MISMATCH"
        bad += flag != "ok"
        print(f"{k:24s} {flag}  staged {new[k][:12]}  r3 {old[k][:12]}  r3-on-tiles {mid[k][:12]}")
    print("all bit-identical" if not bad else f"{bad} mismatches")
    sys.exit(1 if bad else 0)
