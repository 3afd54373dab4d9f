#!/usr/bin/env python3
"""Bit-for-bit A/B of two builds of the library (e.g. lib/libssw_hip.so against a variant of tools/build_variant.sh): every build
runs in a child process (SSW_LIB_PATH), transforms the same synthetic frames -- ssw_dct2d forward, orthonormal and inverse on
shapes of every strategy, one batch embed + extract -- and prints a digest per case; the parent compares the digests.
usage: python tools/lib_ab_check.py LIB_A LIB_B
       python tools/lib_ab_check.py --golden LIB OUT.json     (writes the digests of one build: tests/golden/gemm_digests.json)"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(2160, 3840, 8), (2160, 3840, 1), (1080, 1920, 12), (720, 1280, 20), (4320, 7680, 2), (272, 512, 40), (444, 640, 3), (2160, 3840, 3)]


def child():
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    import spread_spectrum_watermarking_amd as wm
    from spread_spectrum_watermarking_amd import _lib as L
    from spread_spectrum_watermarking_amd.api import check
    ctx = wm.Context(0)
    lib = ctx._lib
    for (h, w, n) in SHAPES:
        rgb = ctx.alloc(n * h * w * 12)
        check(lib.ssw_synth_frames(ctx.handle, 7, 0, n, w, h, rgb.ptr), "synth")
        y = ctx.alloc(n * h * w * 4)
        check(lib.ssw_rgb_to_yiq(ctx.handle, rgb.ptr, n, w, h, y.ptr, None, None), "yiq")
        y0 = y.to_host(np.float32, (n, h, w))
        for kind, name in ((L.DCT2, "fwd"), (L.DCT2_ORTHOGONAL, "ortho"), (L.DCT3, "inv")):
            t = ctx.to_device(y0)
            check(lib.ssw_dct2d(ctx.handle, kind, L.PRECISION_F64, n, w, h, t.ptr), "dct")
            out = t.to_host(np.float32, (n, h, w))
            print(f"DIGEST dct {h}x{w}x{n} {name} {hashlib.sha256(out.tobytes()).hexdigest()[:16]}", flush=True)
            t.free()
        if (h, w, n) in ((2160, 3840, 8), (1080, 1920, 12), (272, 512, 40)):
            k = 500
            marks = np.random.default_rng(3).standard_normal((n, k)).astype(np.float32)
            dm = ctx.to_device(marks)
            out, ext, sims = ctx.alloc(n * h * w * 12), ctx.alloc(n * k * 4), ctx.alloc(n * 4)
            cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64)
            check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg), rgb.ptr, n, w, h, dm.ptr, k, out.ptr, None, None), "embed")
            check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), rgb.ptr, out.ptr, n, w, h, k, ext.ptr, dm.ptr, sims.ptr), "extract")
            for nm, b, shp in (("marked", out, (n, h, w, 3)), ("ext", ext, (n, k)), ("sims", sims, (n,))):
                print(f"DIGEST batch {h}x{w}x{n} {nm} {hashlib.sha256(b.to_host(np.float32, shp).tobytes()).hexdigest()[:16]}", flush=True)
            for b in (dm, out, ext, sims):
                b.free()
        rgb.free(); y.free()
    ctx.close()


def digests(lib=None, env=None):
    """{case: digest} of one build (None: the default library), computed in a child process."""
    e = dict(os.environ, **(env or {}))
    if lib:
        e["SSW_LIB_PATH"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-2000:] + r.stderr[-3000:])
    return {" ".join(l.split()[1:-1]): l.split()[-1] for l in r.stdout.splitlines() if l.startswith("DIGEST")}


def main():
    if sys.argv[1] == "--golden":
        import json
        d = digests(sys.argv[2])
        with open(sys.argv[3], "w") as f:
            json.dump({"_how": "python tools/lib_ab_check.py --golden <library built with -DSSW_GEMM_DMA=0: the register-staged r5 GEMM kernel> "
                               "tests/golden/gemm_digests.json -- sha256[:16] of the f32 outputs on ssw_synth_frames(seed 7) inputs",
                       "digests": d}, f, indent=1)
        print(f"{len(d)} digests -> {sys.argv[3]}")
        return 0
    libs = sys.argv[1:3]
    try:
        res = [digests(lib) for lib in libs]
    except RuntimeError as e:
        print(e)
        return 2
    bad = 0
    for key in res[0]:
        same = res[0][key] == res[1].get(key)
        bad += not same
        print(("same   " if same else "DIFFER ") + key, res[0][key], res[1].get(key))
    print(f"{len(res[0])} cases, {bad} differ")
    return 1 if bad else 0


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        sys.exit(main())
