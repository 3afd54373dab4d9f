"""numpy <-> device helpers for the GPU parity tests: every call goes through the C ABI."""
import ctypes as C

import numpy as np

from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import Context, check

_ctx = None


def ctx() -> Context:
    global _ctx
    if _ctx is None:
        _ctx = Context(0)
    return _ctx


class fresh_ctx:
    """`with fresh_ctx(): ...` -- the helpers of this module run on a new Context inside the block (tests that change a
    process-wide tuning value: an existing context's workspaces and plans were built under the old one)."""

    def __enter__(self):
        global _ctx
        self._prev, _ctx = _ctx, Context(0)
        return _ctx

    def __exit__(self, *exc):
        global _ctx
        _ctx.close()
        _ctx = self._prev
        return False


def lib():
    return L.load()


def dct2d(planes, dct_type=L.DCT2, precision=L.PRECISION_F32):
    """planes: [n, h, w] or [h, w] float32 -> same shape."""
    a = np.ascontiguousarray(planes, dtype=np.float32)
    single = a.ndim == 2
    if single:
        a = a[None]
    n, h, w = a.shape
    d = ctx().to_device(a)
    check(lib().ssw_dct2d(ctx().handle, dct_type, precision, n, w, h, d.ptr), "ssw_dct2d")
    out = d.to_host(np.float32, a.shape)
    d.free()
    return out[0] if single else out


def rgb_to_yiq(rgb, with_iq=True):
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    if a.ndim == 3:
        a = a[None]
    n, h, w, _ = a.shape
    d = ctx().to_device(a)
    y = ctx().alloc(n * h * w * 4)
    i = ctx().alloc(n * h * w * 4) if with_iq else None
    q = ctx().alloc(n * h * w * 4) if with_iq else None
    check(lib().ssw_rgb_to_yiq(ctx().handle, d.ptr, n, w, h, y.ptr, i.ptr if i else None, q.ptr if q else None), "ssw_rgb_to_yiq")
    outs = [b.to_host(np.float32, (n, h, w)) if b else None for b in (y, i, q)]
    for b in (d, y, i, q):
        if b:
            b.free()
    return outs


def yiq_to_rgb(y, i, q):
    y, i, q = (np.ascontiguousarray(v, dtype=np.float32) for v in (y, i, q))
    if y.ndim == 2:
        y, i, q = y[None], i[None], q[None]
    n, h, w = y.shape
    dy, di, dq = ctx().to_device(y), ctx().to_device(i), ctx().to_device(q)
    out = ctx().alloc(n * h * w * 12)
    check(lib().ssw_yiq_to_rgb(ctx().handle, dy.ptr, di.ptr, dq.ptr, n, w, h, out.ptr), "ssw_yiq_to_rgb")
    r = out.to_host(np.float32, (n, h, w, 3))
    for b in (dy, di, dq, out):
        b.free()
    return r


def topk(coef, k, ordering=L.ORDER_ENERGY):
    """coef [n, h, w] or [h, w] -> indices [n, k] (or [k]) uint32."""
    a = np.ascontiguousarray(coef, dtype=np.float32)
    single = a.ndim == 2
    if single:
        a = a[None]
    n, h, w = a.shape
    d = ctx().to_device(a)
    idx = ctx().alloc(max(n * k, 1) * 4)
    check(lib().ssw_topk_indices(ctx().handle, d.ptr, n, w, h, ordering, k, idx.ptr), "ssw_topk_indices")
    out = idx.to_host(np.uint32, (n, k))
    d.free(); idx.free()
    return out[0] if single else out


def embed(coef, indices, marks, method=L.OPTION2, alpha=0.1):
    """coef [n, plane], indices [n, k], marks [n, n_marks, k] -> embedded coef."""
    a = np.ascontiguousarray(coef, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    m = np.ascontiguousarray(marks, dtype=np.float32)
    n, plane = a.shape
    k = idx.shape[1]
    n_marks = m.shape[1]
    d, di, dm = ctx().to_device(a), ctx().to_device(idx), ctx().to_device(m)
    check(lib().ssw_embed_coefficients(ctx().handle, d.ptr, n, plane, di.ptr, k, method, C.c_float(alpha), dm.ptr, n_marks),
          "ssw_embed_coefficients")
    out = d.to_host(np.float32, a.shape)
    for b in (d, di, dm):
        b.free()
    return out


def extract(base, derived, indices, method=L.OPTION2, alpha=0.1):
    b, dv = np.ascontiguousarray(base, dtype=np.float32), np.ascontiguousarray(derived, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    n, plane = b.shape
    k = idx.shape[1]
    db, dd, di = ctx().to_device(b), ctx().to_device(dv), ctx().to_device(idx)
    out = ctx().alloc(max(n * k, 1) * 4)
    check(lib().ssw_extract_coefficients(ctx().handle, db.ptr, dd.ptr, n, plane, di.ptr, k, method, C.c_float(alpha), out.ptr),
          "ssw_extract_coefficients")
    r = out.to_host(np.float32, (n, k))
    for x in (db, dd, di, out):
        x.free()
    return r


def similarity_batch(extracted, marks):
    e, m = np.ascontiguousarray(extracted, dtype=np.float32), np.ascontiguousarray(marks, dtype=np.float32)
    n, k = e.shape
    de, dm = ctx().to_device(e), ctx().to_device(m)
    out = ctx().alloc(n * 4)
    check(lib().ssw_similarity_batch(ctx().handle, de.ptr, dm.ptr, n, k, out.ptr), "ssw_similarity_batch")
    r = out.to_host(np.float32, (n,))
    for x in (de, dm, out):
        x.free()
    return r


def synth(seed, first, n, w, h):
    d = ctx().alloc(n * h * w * 12)
    check(lib().ssw_synth_frames(ctx().handle, seed, first, n, w, h, d.ptr), "ssw_synth_frames")
    r = d.to_host(np.float32, (n, h, w, 3))
    d.free()
    return r


def batch_embed(rgb, marks, cfg=None, want_coef=False, want_idx=False):
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    m = np.ascontiguousarray(marks, dtype=np.float32)
    n, h, w, _ = a.shape
    k = m.shape[1]
    c = cfg or default_config()
    d, dm = ctx().to_device(a), ctx().to_device(m)
    out = ctx().alloc(a.nbytes)
    coef = ctx().alloc(n * h * w * 4) if want_coef else None
    idx = ctx().alloc(n * k * 4) if want_idx else None
    check(lib().ssw_batch_embed(ctx().handle, C.byref(c), d.ptr, n, w, h, dm.ptr, k, out.ptr,
                                coef.ptr if coef else None, idx.ptr if idx else None), "ssw_batch_embed")
    res = {"rgb": out.to_host(np.float32, a.shape)}
    if coef:
        res["coef"] = coef.to_host(np.float32, (n, h, w))
    if idx:
        res["idx"] = idx.to_host(np.uint32, (n, k))
    for b in (d, dm, out, coef, idx):
        if b:
            b.free()
    return res


def batch_extract(base_rgb, derived_rgb, k, marks=None, cfg=None):
    b, dv = np.ascontiguousarray(base_rgb, dtype=np.float32), np.ascontiguousarray(derived_rgb, dtype=np.float32)
    n, h, w, _ = b.shape
    c = cfg or default_config()
    db, dd = ctx().to_device(b), ctx().to_device(dv)
    ext = ctx().alloc(max(n * k, 1) * 4)
    dm = ctx().to_device(np.ascontiguousarray(marks, dtype=np.float32)) if marks is not None else None
    sims = ctx().alloc(n * 4) if marks is not None else None
    check(lib().ssw_batch_extract(ctx().handle, C.byref(c), db.ptr, dd.ptr, n, w, h, k, ext.ptr,
                                  dm.ptr if dm else None, sims.ptr if sims else None), "ssw_batch_extract")
    e = ext.to_host(np.float32, (n, k))
    s = sims.to_host(np.float32, (n,)) if sims else None
    for x in (db, dd, ext, dm, sims):
        if x:
            x.free()
    return e, s


def default_config(precision=L.PRECISION_F64, ordering=L.ORDER_ENERGY, method=L.OPTION2, alpha=0.1):
    return L.Config(ordering, method, alpha, precision)


def convert_u8_to_f32(a_u8):
    a = np.ascontiguousarray(a_u8, dtype=np.uint8)
    d = ctx().to_device(a)
    out = ctx().alloc(a.size * 4)
    check(lib().ssw_convert_rgb8_to_f32(ctx().handle, d.ptr, a.size, out.ptr), "ssw_convert_rgb8_to_f32")
    r = out.to_host(np.float32, a.shape)
    d.free(); out.free()
    return r


def convert_f32_to_u8(a_f32):
    a = np.ascontiguousarray(a_f32, dtype=np.float32)
    d = ctx().to_device(a)
    out = ctx().alloc(max(a.size, 16))
    check(lib().ssw_convert_f32_to_rgb8(ctx().handle, d.ptr, a.size, out.ptr), "ssw_convert_f32_to_rgb8")
    r = out.to_host(np.uint8, a.shape)
    d.free(); out.free()
    return r


def resize_rgb8(frames_u8, nw, nh):
    a = np.ascontiguousarray(frames_u8, dtype=np.uint8)
    single = a.ndim == 3
    if single:
        a = a[None]
    n, h, w, _ = a.shape
    d = ctx().to_device(a)
    out = ctx().alloc(max(n * nh * nw * 3, 16))
    check(lib().ssw_resize_rgb8(ctx().handle, d.ptr, n, w, h, nw, nh, out.ptr), "ssw_resize_rgb8")
    r = out.to_host(np.uint8, (n, nh, nw, 3))
    d.free(); out.free()
    return r[0] if single else r


def batch_embed_rgb8(rgb_u8, marks, cfg=None):
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    m = np.ascontiguousarray(marks, dtype=np.float32)
    n, h, w, _ = a.shape
    c = cfg or default_config()
    d, dm = ctx().to_device(a), ctx().to_device(m)
    out = ctx().alloc(a.nbytes)
    check(lib().ssw_batch_embed_rgb8(ctx().handle, C.byref(c), d.ptr, n, w, h, dm.ptr, m.shape[1], out.ptr), "ssw_batch_embed_rgb8")
    r = out.to_host(np.uint8, a.shape)
    for b in (d, dm, out):
        b.free()
    return r


def batch_extract_rgb8(base_u8, derived_u8, k, marks, cfg=None):
    b, dv = np.ascontiguousarray(base_u8, dtype=np.uint8), np.ascontiguousarray(derived_u8, dtype=np.uint8)
    n, h, w, _ = b.shape
    c = cfg or default_config()
    db, dd = ctx().to_device(b), ctx().to_device(dv)
    dm = ctx().to_device(np.ascontiguousarray(marks, dtype=np.float32))
    ext, sims = ctx().alloc(n * k * 4), ctx().alloc(n * 4)
    check(lib().ssw_batch_extract_rgb8(ctx().handle, C.byref(c), db.ptr, dd.ptr, n, w, h, k, ext.ptr, dm.ptr, sims.ptr),
          "ssw_batch_extract_rgb8")
    e, s = ext.to_host(np.float32, (n, k)), sims.to_host(np.float32, (n,))
    for x in (db, dd, dm, ext, sims):
        x.free()
    return e, s


def batch_embed_rgb16(rgb_u16, marks, cfg=None):
    """ssw_batch_embed_rgb16: u16 frames in, f32 marked frames out (what Writer::mark returns)."""
    a = np.ascontiguousarray(rgb_u16, dtype=np.uint16)
    m = np.ascontiguousarray(marks, dtype=np.float32)
    n, h, w, _ = a.shape
    c = cfg or default_config()
    d, dm = ctx().to_device(a), ctx().to_device(m)
    out = ctx().alloc(a.size * 4)
    check(lib().ssw_batch_embed_rgb16(ctx().handle, C.byref(c), d.ptr, n, w, h, dm.ptr, m.shape[1], out.ptr), "ssw_batch_embed_rgb16")
    r = out.to_host(np.float32, a.shape)
    for b in (d, dm, out):
        b.free()
    return r


def batch_extract_rgb16(base_u16, derived_u16, k, marks, cfg=None):
    b, dv = np.ascontiguousarray(base_u16, dtype=np.uint16), np.ascontiguousarray(derived_u16, dtype=np.uint16)
    n, h, w, _ = b.shape
    c = cfg or default_config()
    db, dd = ctx().to_device(b), ctx().to_device(dv)
    dm = ctx().to_device(np.ascontiguousarray(marks, dtype=np.float32))
    ext, sims = ctx().alloc(n * k * 4), ctx().alloc(n * 4)
    check(lib().ssw_batch_extract_rgb16(ctx().handle, C.byref(c), db.ptr, dd.ptr, n, w, h, k, ext.ptr, dm.ptr, sims.ptr),
          "ssw_batch_extract_rgb16")
    e, s = ext.to_host(np.float32, (n, k)), sims.to_host(np.float32, (n,))
    for x in (db, dd, dm, ext, sims):
        x.free()
    return e, s


def convert_rgb16(values_u16=None, values_f32=None):
    """ssw_convert_rgb16_to_f32 (u16 in) or ssw_convert_f32_to_rgb16 (f32 in) on a flat array."""
    if values_u16 is not None:
        a = np.ascontiguousarray(values_u16, dtype=np.uint16)
        d, out = ctx().to_device(a), ctx().alloc(a.size * 4)
        check(lib().ssw_convert_rgb16_to_f32(ctx().handle, d.ptr, a.size, out.ptr), "ssw_convert_rgb16_to_f32")
        r = out.to_host(np.float32, a.shape)
    else:
        a = np.ascontiguousarray(values_f32, dtype=np.float32)
        d, out = ctx().to_device(a), ctx().alloc(a.size * 2)
        check(lib().ssw_convert_f32_to_rgb16(ctx().handle, d.ptr, a.size, out.ptr), "ssw_convert_f32_to_rgb16")
        r = out.to_host(np.uint16, a.shape)
    d.free(); out.free()
    return r


def similarity_matrix(extracted, marks_db):
    e, m = np.ascontiguousarray(extracted, dtype=np.float32), np.ascontiguousarray(marks_db, dtype=np.float32)
    b, k = e.shape
    n = m.shape[0]
    de, dm = ctx().to_device(e), ctx().to_device(m)
    out = ctx().alloc(b * n * 4)
    check(lib().ssw_similarity_matrix(ctx().handle, de.ptr, b, dm.ptr, n, k, out.ptr), "ssw_similarity_matrix")
    r = out.to_host(np.float32, (b, n))
    for x in (de, dm, out):
        x.free()
    return r
