"""ctypes front-end of the CPU oracle (oracle/libssw_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- never by the product package
(spread_spectrum_watermarking_amd/), which has no CPU path at all.

Every function cites the reference file:line it follows in oracle/ssw_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSW_ORACLE_LIB: load another build of the same source (the ASan/UBSan one of `make sanitize`)
_LIB_PATH = os.environ.get("SSW_ORACLE_LIB") or os.path.join(_HERE, "libssw_oracle.so")

DCT2, DCT2_ORTHOGONAL, DCT3 = 0, 1, 2
BACKEND_F64, BACKEND_F32, BACKEND_NAIVE_F64 = 0, 1, 2
ORDER_ENERGY, ORDER_ENERGY_ORTHOGONAL, ORDER_LEGACY = 0, 1, 2
OPTION1, OPTION2, OPTION3 = 1, 2, 3

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)


def build() -> str:
    """Compile the oracle with its committed Makefile (gcc, plain C)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def _load() -> C.CDLL:
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    lib.sswo_rgb_to_yiq.argtypes = [_f32p, C.c_size_t, _f32p, _f32p, _f32p]
    lib.sswo_rgb_to_yiq.restype = None
    lib.sswo_yiq_to_rgb.argtypes = [_f32p, _f32p, _f32p, C.c_size_t, _f32p]
    lib.sswo_yiq_to_rgb.restype = None
    lib.sswo_dct2d.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, _f32p]
    lib.sswo_dct2d.restype = C.c_int
    lib.sswo_dct1d.argtypes = [C.c_int, C.c_int, C.c_size_t, _f32p]
    lib.sswo_dct1d.restype = C.c_int
    lib.sswo_indices.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, _u64p]
    lib.sswo_indices.restype = C.c_size_t
    lib.sswo_order_key.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
    lib.sswo_order_key.restype = C.c_int32
    lib.sswo_embed.argtypes = [_f32p, C.c_size_t, _u64p, C.c_size_t, C.c_int, C.c_float,
                               C.POINTER(_f32p), C.POINTER(C.c_size_t), C.c_size_t]
    lib.sswo_embed.restype = None
    lib.sswo_extract.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _u64p, C.c_int, C.c_float,
                                 _f32p, C.c_size_t]
    lib.sswo_extract.restype = C.c_int
    lib.sswo_similarity.argtypes = [_f32p, _f32p, C.c_size_t]
    lib.sswo_similarity.restype = C.c_float
    lib.sswo_synth_frame.argtypes = [C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, _f32p]
    lib.sswo_synth_frame.restype = None
    _u8p = C.POINTER(C.c_uint8)
    lib.sswo_u8_to_f32.argtypes = [_u8p, C.c_size_t, _f32p]
    lib.sswo_u8_to_f32.restype = None
    lib.sswo_f32_to_u8.argtypes = [_f32p, C.c_size_t, _u8p]
    lib.sswo_f32_to_u8.restype = None
    _u16p = C.POINTER(C.c_uint16)
    lib.sswo_u16_to_f32.argtypes = [_u16p, C.c_size_t, _f32p]
    lib.sswo_u16_to_f32.restype = None
    lib.sswo_f32_to_u16.argtypes = [_f32p, C.c_size_t, _u16p]
    lib.sswo_f32_to_u16.restype = None
    lib.sswo_resize_rgb8.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _u8p]
    lib.sswo_resize_rgb8.restype = None
    lib.sswo_resize_taps.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32), _f32p, C.c_size_t]
    lib.sswo_resize_taps.restype = C.c_size_t
    lib.sswo_embed_frame.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                     C.c_float, _f32p, C.c_size_t, C.c_int, _f32p]
    lib.sswo_embed_frame.restype = None
    lib.sswo_extract_frame.argtypes = [_f32p, _f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                       C.c_int, C.c_float, _f32p, C.c_size_t, C.c_int, _f32p]
    lib.sswo_extract_frame.restype = C.c_float
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def rgb_to_yiq(rgb_hwc):
    """rgb [H,W,3] f32 -> (y, i, q) planes [H,W]."""
    rgb = _f32(rgb_hwc)
    h, w = rgb.shape[:2]
    y, i, q = (np.empty((h, w), np.float32) for _ in range(3))
    lib().sswo_rgb_to_yiq(_p(rgb), h * w, _p(y), _p(i), _p(q))
    return y, i, q


def yiq_to_rgb(y, i, q):
    y, i, q = _f32(y), _f32(i), _f32(q)
    h, w = y.shape
    rgb = np.empty((h, w, 3), np.float32)
    lib().sswo_yiq_to_rgb(_p(y), _p(i), _p(q), h * w, _p(rgb))
    return rgb


def dct2d(plane_hw, dct_type=DCT2, backend=BACKEND_F64):
    """Out-of-place convenience around the in-place reference semantics."""
    a = _f32(plane_hw).copy()
    h, w = a.shape
    rc = lib().sswo_dct2d(dct_type, backend, w, h, _p(a))
    if rc != 0:
        raise ValueError("sswo_dct2d: bad arguments")
    return a


def dct1d(x, dct_type=DCT2, backend=BACKEND_F64):
    a = _f32(x).copy()
    rc = lib().sswo_dct1d(dct_type, backend, a.size, _p(a))
    if rc != 0:
        raise ValueError("sswo_dct1d: bad arguments")
    return a


def indices(coef, ordering=ORDER_ENERGY, k=None, width=None, height=None):
    """First k entries of obtain_indices_by_function (k=None: all n-1)."""
    c = _f32(coef)
    if c.ndim == 2:
        height, width = c.shape
    n = c.size
    if width is None:
        width, height = n, 1
    if k is None:
        k = n - 1
    k = min(int(k), max(n - 1, 0))
    out = np.empty(max(k, 1), np.uint64)
    got = lib().sswo_indices(_p(c.reshape(-1)), n, ordering, width, height, k,
                             out.ctypes.data_as(_u64p))
    return out[:got].copy()


def order_keys(coef, ordering=ORDER_ENERGY, width=None, height=None):
    """int32 sort keys (larger == earlier) for every coefficient; for tie-aware checks."""
    c = _f32(coef)
    if c.ndim == 2:
        height, width = c.shape
    flat = c.reshape(-1)
    if width is None:
        width, height = flat.size, 1
    f = lib().sswo_order_key
    if ordering == ORDER_ENERGY:      # vectorised fast path, same arithmetic
        e = (flat * flat).astype(np.float32)
        b = e.view(np.int32)
        return b ^ ((b >> 31).view(np.uint32) >> np.uint32(1)).view(np.int32)
    return np.array([f(ordering, width, height, j, float(flat[j])) for j in range(flat.size)], np.int32)


def embed(coef, idx, marks, method=OPTION2, alpha=0.1):
    """Writer::embed_watermark on a copy of `coef` (flat or 2-D). `marks`: list of 1-D arrays."""
    c = _f32(coef).copy()
    flat = c.reshape(-1)
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    ms = [_f32(m) for m in marks]
    arr = (_f32p * len(ms))(*[_p(m) for m in ms])
    lens = (C.c_size_t * len(ms))(*[m.size for m in ms])
    lib().sswo_embed(_p(flat), flat.size, idx.ctypes.data_as(_u64p), idx.size, method,
                     C.c_float(alpha), arr, lens, len(ms))
    return c


def extract(base, derived, idx, k, method=OPTION2, alpha=0.1):
    b, d = _f32(base).reshape(-1), _f32(derived).reshape(-1)
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    out = np.empty(k, np.float32)
    rc = lib().sswo_extract(_p(b), b.size, _p(d), d.size, idx.ctypes.data_as(_u64p), method,
                            C.c_float(alpha), _p(out), k)
    if rc == 1:
        raise ValueError("Derived coefficient length not equal to base coefficient length.")
    if rc == 2:
        raise ValueError("Desired extraction length exceeds available coefficients.")
    return out


def similarity(extracted, mark) -> float:
    e, m = _f32(extracted), _f32(mark)
    if e.size != m.size:
        raise ValueError("length mismatch")
    return float(lib().sswo_similarity(_p(e), _p(m), e.size))


def synth_frame(seed: int, frame: int, w: int, h: int):
    rgb = np.empty((h, w, 3), np.float32)
    lib().sswo_synth_frame(seed & 0xFFFFFFFF, frame & 0xFFFFFFFF, w, h, _p(rgb))
    return rgb


def embed_frame(rgb, mark, backend=BACKEND_F64, ordering=ORDER_ENERGY, method=OPTION2, alpha=0.1,
                full_sort=False):
    rgb = _f32(rgb)
    h, w = rgb.shape[:2]
    mark = _f32(mark)
    out = np.empty_like(rgb)
    lib().sswo_embed_frame(_p(rgb), w, h, backend, ordering, method, C.c_float(alpha), _p(mark),
                           mark.size, int(full_sort), _p(out))
    return out


def extract_frame(base_rgb, derived_rgb, mark, backend=BACKEND_F64, ordering=ORDER_ENERGY,
                  method=OPTION2, alpha=0.1, full_sort=False):
    b, d = _f32(base_rgb), _f32(derived_rgb)
    h, w = b.shape[:2]
    mark = _f32(mark)
    ext = np.empty(mark.size, np.float32)
    sim = lib().sswo_extract_frame(_p(b), _p(d), w, h, backend, ordering, method, C.c_float(alpha),
                                   _p(mark), mark.size, int(full_sort), _p(ext))
    return ext, float(sim)


def u8_to_f32(img_u8):
    """`into_rgb32f` for 8-bit input: v / 255."""
    a = np.ascontiguousarray(img_u8, dtype=np.uint8)
    out = np.empty(a.shape, np.float32)
    lib().sswo_u8_to_f32(a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size, _p(out))
    return out


def f32_to_u8(img_f32):
    """`into_rgb8` from Rgb32F: round(clamp(v, 0, 1) * 255)."""
    a = _f32(img_f32)
    out = np.empty(a.shape, np.uint8)
    lib().sswo_f32_to_u8(_p(a), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def u16_to_f32(img_u16):
    """`into_rgb32f` for 16-bit input: v / 65535."""
    a = np.ascontiguousarray(img_u16, dtype=np.uint16)
    out = np.empty(a.shape, np.float32)
    lib().sswo_u16_to_f32(a.ctypes.data_as(C.POINTER(C.c_uint16)), a.size, _p(out))
    return out


def f32_to_u16(img_f32):
    """`into_rgb16` from Rgb32F: round(clamp(v, 0, 1) * 65535)."""
    a = _f32(img_f32)
    out = np.empty(a.shape, np.uint16)
    lib().sswo_f32_to_u16(_p(a), a.size, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def resize_rgb8(img_u8, new_w, new_h):
    """image::imageops::resize(img, new_w, new_h, CatmullRom) on an [H, W, 3] uint8 image."""
    a = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = a.shape[:2]
    out = np.empty((new_h, new_w, 3), np.uint8)
    u8p = C.POINTER(C.c_uint8)
    lib().sswo_resize_rgb8(a.ctypes.data_as(u8p), w, h, new_w, new_h, out.ctypes.data_as(u8p))
    return out


def resize_taps(in_len, out_len):
    """(left[out_len] uint32, ntaps[out_len] uint32, weights[out_len, max_taps] f32) exactly as the
    oracle computes them (f32 arithmetic); used to check the device's tap tables."""
    max_t = int(4 * max(1.0, in_len / out_len)) + 4
    left = np.zeros(out_len, np.uint32); nt = np.zeros(out_len, np.uint32)
    ws = np.zeros((out_len, max_t), np.float32)
    l = C.c_uint32()
    for o in range(out_len):
        row = np.zeros(max_t, np.float32)
        n = lib().sswo_resize_taps(in_len, out_len, o, C.byref(l), _p(row), max_t)
        left[o], nt[o], ws[o] = l.value, n, row
    return left, nt, ws
