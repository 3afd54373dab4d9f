"""Host-side mirror of the crate's public surface (lib.rs:81-85) over the C ABI.

Same names, argument meaning and error behaviour as the Rust reference
(file:line relative to the reference tree):

  Writer / WriteConfig / Insertion        src/algorithm.rs:68-112, :285-433
  Reader / ReaderDerived / ReadConfig      src/algorithm.rs:114-140, :435-594
  OrderingMethod                           src/algorithm.rs:142-191
  MarkBuf (Mark)                           src/algorithm.rs:596-666
  Tester / Similarity                      src/algorithm.rs:668-715

Images are numpy arrays [H, W, 3]: float32 in [0, 1] (what `into_rgb32f()` yields,
algorithm.rs:308) or uint8 (handed to the library as they are: `into_rgb32f()`, v / 255 like the
image crate does, runs on the device and 3 instead of 12 bytes per pixel cross PCIe).
Where the reference panics, an SswError is raised.  All arithmetic runs on the
GPU through libssw_hip.so; there is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

from . import _lib as L
from ._lib import SswError, check


# ---- context ---------------------------------------------------------------------------------
class Context:
    """One per GPU (include/ssw.h: ssw_ctx).  Not thread-safe, like the reference's !Send types."""

    def __init__(self, device_id: int = 0):
        self._lib = L.load()
        h = C.c_void_p()
        check(self._lib.ssw_ctx_create(device_id, C.byref(h)), "ssw_ctx_create")
        self.handle = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ssw_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self._lib.ssw_ctx_synchronize(self.handle), "ssw_ctx_synchronize")

    def set_stream(self, hip_stream=None):
        """Enqueue on the caller's hipStream_t (an int / c_void_p, e.g. torch.cuda.current_stream().cuda_stream);
        None returns to the context's private stream.  See the stream contract in include/ssw.h."""
        check(self._lib.ssw_ctx_set_stream(self.handle, C.c_void_p(hip_stream) if hip_stream else None), "ssw_ctx_set_stream")

    def wait_event(self, hip_event):
        check(self._lib.ssw_ctx_wait_event(self.handle, C.c_void_p(hip_event)), "ssw_ctx_wait_event")

    def record_event(self, hip_event):
        check(self._lib.ssw_ctx_record_event(self.handle, C.c_void_p(hip_event)), "ssw_ctx_record_event")

    def set_chunk_frames(self, n: int):
        check(self._lib.ssw_ctx_set_chunk_frames(self.handle, n), "ssw_ctx_set_chunk_frames")

    def pass_frames(self, n_frames: int, w: int, h: int) -> int:
        """Frames per internal pass a batch call over n_frames frames of w x h would use."""
        return int(self._lib.ssw_ctx_pass_frames(self.handle, n_frames, w, h))

    def set_dct_folding(self, level=True):
        """Basis-GEMM strategy (include/ssw.h): False / 0 dense; 1 (= 2) one folding level inside the GEMM
        kernel; 3 / 4 operand-ready GEMMs with one / two
        folding levels; 5 a third level on long forward row passes; 6 the same without the size
        threshold.  True selects the default (5)."""
        lvl = (L.DCT_FOLDING_DEFAULT if level else 0) if isinstance(level, bool) else int(level)
        check(self._lib.ssw_ctx_set_dct_folding(self.handle, lvl), "ssw_ctx_set_dct_folding")

    def enable_timing(self, on: bool = True):
        check(self._lib.ssw_ctx_enable_timing(self.handle, int(on)), "ssw_ctx_enable_timing")

    def reset_timing(self):
        check(self._lib.ssw_ctx_reset_timing(self.handle), "ssw_ctx_reset_timing")

    def timing(self) -> dict:
        """Per stage: milliseconds, timed regions, and the work done in them (flop for the GEMM stages,
        algorithmic bytes for the HBM-bound ones)."""
        ms = (C.c_double * len(L.STAGES))()
        n = (C.c_uint64 * len(L.STAGES))()
        work = (C.c_double * len(L.STAGES))()
        check(self._lib.ssw_ctx_get_timing(self.handle, ms, n), "ssw_ctx_get_timing")
        check(self._lib.ssw_ctx_get_work(self.handle, work), "ssw_ctx_get_work")
        traffic = (C.c_double * len(L.STAGES))()
        check(self._lib.ssw_ctx_get_traffic(self.handle, traffic), "ssw_ctx_get_traffic")
        return {s: {"ms": ms[i], "launches": int(n[i]), "work": work[i], "bytes": traffic[i]} for i, s in enumerate(L.STAGES)}

    def transform_plan(self, n_frames: int, w: int, h: int, dct_type: int = L.DCT2) -> dict:
        """Which strategy of the 2-D transform a batch of this shape takes (include/ssw.h: ssw_ctx_transform_plan)."""
        f = C.c_uint32()
        check(self._lib.ssw_ctx_transform_plan(self.handle, n_frames, w, h, dct_type, C.byref(f)), "ssw_ctx_transform_plan")
        return {name: bool(f.value & bit) for name, bit in L.PLAN_FLAGS.items()}

    def set_overlap(self, on: bool = True):
        """Two chunks in flight on two streams in the batch entry points (default) or one at a time."""
        check(self._lib.ssw_ctx_set_overlap(self.handle, int(on)), "ssw_ctx_set_overlap")

    def set_prune(self, on: bool = True):
        """Batch extract: derived frames transformed only where extract reads them (default) or fully."""
        check(self._lib.ssw_ctx_set_prune(self.handle, int(on)), "ssw_ctx_set_prune")

    def set_odd_split(self, on: bool = True):
        """f64 transforms: odd halves as rotated quarter-length cosine + sine pairs (default) or as exact folded sums."""
        check(self._lib.ssw_ctx_set_odd_split(self.handle, int(on)), "ssw_ctx_set_odd_split")

    def prune_stats(self) -> dict:
        st = (C.c_uint64 * 3)()
        check(self._lib.ssw_ctx_get_prune_stats(self.handle, st), "ssw_ctx_get_prune_stats")
        return {"pruned_chunks": int(st[0]), "redone_chunks": int(st[1]), "columns_needed": int(st[2])}

    def select_stats(self) -> dict:
        st = (C.c_uint64 * 2)()
        check(self._lib.ssw_ctx_get_select_stats(self.handle, st), "ssw_ctx_get_select_stats")
        return {"frames": int(st[0]), "exact_fallback_frames": int(st[1])}

    def set_copy_threads(self, n: int = 0):
        """Host threads that move pageable buffers through the pinned staging ring (0 = automatic)."""
        check(self._lib.ssw_ctx_set_copy_threads(self.handle, int(n)), "ssw_ctx_set_copy_threads")

    def transfer_stats(self, reset: bool = False) -> dict:
        """Bytes / host seconds of the host-buffer entry points' uploads and downloads (include/ssw.h)."""
        st = (C.c_double * len(L.TRANSFER_STATS))()
        check(self._lib.ssw_ctx_get_transfer_stats(self.handle, st, int(reset)), "ssw_ctx_get_transfer_stats")
        return {n: float(st[i]) for i, n in enumerate(L.TRANSFER_STATS)}

    def pinned_empty(self, shape, dtype) -> np.ndarray:
        """numpy array in pinned (page-locked) host memory: the handles DMA straight from / into it.
        The memory belongs to the context and is released when the array (and its views) are gone."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        check(self._lib.ssw_host_alloc(self.handle, n, C.byref(p)), "ssw_host_alloc")
        return np.asarray(_PinnedBlock(self, p, n))[:n].view(dt).reshape(shape)

    def mem_info(self):
        free, total = C.c_size_t(), C.c_size_t()
        check(self._lib.ssw_dev_mem_info(self.handle, C.byref(free), C.byref(total)), "ssw_dev_mem_info")
        return int(free.value), int(total.value)

    # device memory for hosts without their own allocator (tests; the bench uses torch tensors)
    def alloc(self, nbytes: int) -> "DeviceBuffer":
        return DeviceBuffer(self, nbytes)

    def to_device(self, array: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(array)
        buf = DeviceBuffer(self, a.nbytes)
        check(self._lib.ssw_copy_to_dev(self.handle, buf.ptr, a.ctypes.data, a.nbytes), "ssw_copy_to_dev")
        return buf


class _PinnedBlock:
    """Owner of one ssw_host_alloc block, exposed to numpy through __array_interface__: arrays made from it
    (and their views) keep it alive, the block is released with the last of them."""

    def __init__(self, ctx: Context, ptr, nbytes: int):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, nbytes
        self.__array_interface__ = {"data": (int(ptr.value), False), "shape": (max(nbytes, 1),), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                self.ctx._lib.ssw_host_free(self.ctx.handle, self.ptr)
            self.ptr = None
        except Exception:
            pass


class DeviceBuffer:
    def __init__(self, ctx: Context, nbytes: int):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        check(ctx._lib.ssw_dev_alloc(ctx.handle, nbytes, C.byref(p)), "ssw_dev_alloc")
        self.ptr = p

    def to_host(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(self.ctx._lib.ssw_copy_to_host(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes), "ssw_copy_to_host")
        return out

    def free(self):
        if self.ptr and self.ctx.handle:
            self.ctx._lib.ssw_dev_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


class tuning:
    """Process-wide strategy thresholds / A-B switches (include/ssw.h: ssw_tuning_set).  `with tuning(efold_min=256): ...`
    sets them for the block and restores the previous state; use a fresh Context inside (workspaces and cached plans of an
    existing context were built under the old values)."""

    def __init__(self, **values):
        self._values = values
        self._lib = L.load()

    @staticmethod
    def get(name: str) -> int:
        v = C.c_longlong()
        check(L.load().ssw_tuning_get(name.encode(), C.byref(v)), "ssw_tuning_get")
        return int(v.value)

    def __enter__(self):
        self._previous = {name: tuning.get(name) for name in self._values}      # raises on an unknown name before anything is set
        for name, v in self._values.items():
            check(self._lib.ssw_tuning_set(name.encode(), int(v)), f"ssw_tuning_set({name})")
        return self

    def __exit__(self, *exc):
        for name, v in self._previous.items():        # nested blocks restore the enclosing block's value, not the default
            self._lib.ssw_tuning_set(name.encode(), v)
        return False


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None or _default_ctx.handle is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---- configuration enums (algorithm.rs:68-77, :115-124, :143-152) ---------------------------
@dataclass(frozen=True)
class _Method:
    tag: int
    alpha: float = 0.0
    function: Optional[Callable] = None


class Insertion:
    """x' = x + a w | x (1 + a w) | x exp(a w) | custom closure (algorithm.rs:68-77)."""
    @staticmethod
    def Option1(alpha: float): return _Method(L.OPTION1, float(alpha))
    @staticmethod
    def Option2(alpha: float): return _Method(L.OPTION2, float(alpha))
    @staticmethod
    def Option3(alpha: float): return _Method(L.OPTION3, float(alpha))
    @staticmethod
    def Custom(function: Callable): return _Method(L.METHOD_CUSTOM, 0.0, function)


Extraction = Insertion      # same variants, inverse functions (algorithm.rs:115-124)


@dataclass(frozen=True)
class _Ordering:
    tag: int
    function: Optional[Callable] = None


class OrderingMethod:
    Energy = _Ordering(L.ORDER_ENERGY)
    EnergyOrthogonal = _Ordering(L.ORDER_ENERGY_ORTHOGONAL)
    Legacy = _Ordering(L.ORDER_LEGACY)
    @staticmethod
    def Custom(function: Callable): return _Ordering(L.ORDER_CUSTOM, function)


class Precision:
    F32 = L.PRECISION_F32      # v_mfma_f32_32x32x2_f32 basis GEMMs: ~1.8x faster, f32-chain accuracy
    F64 = L.PRECISION_F64      # default: v_mfma_f64_16x16x4_f64, correctly rounded ("canonical")


@dataclass
class WriteConfig:
    """algorithm.rs:99-112; default Option2(0.1) + Energy."""
    insertion: _Method = field(default_factory=lambda: Insertion.Option2(0.1))
    ordering: _Ordering = OrderingMethod.Energy
    precision: int = Precision.F64

    @staticmethod
    def default(): return WriteConfig()

    def _c(self) -> L.Config:
        return L.Config(self.ordering.tag, self.insertion.tag, self.insertion.alpha, self.precision)


@dataclass
class ReadConfig:
    """algorithm.rs:127-140; default Option2(0.1) + Energy."""
    extraction: _Method = field(default_factory=lambda: Extraction.Option2(0.1))
    ordering: _Ordering = OrderingMethod.Energy
    precision: int = Precision.F64

    @staticmethod
    def default(): return ReadConfig()

    def _c(self) -> L.Config:
        return L.Config(self.ordering.tag, self.extraction.tag, self.extraction.alpha, self.precision)


def _as_rgb(image) -> np.ndarray:
    """What crosses the C ABI for a `DynamicImage`: 8- and 16-bit images as they are (`into_rgb32f()`, algorithm.rs:308,
    :476, then runs on the device: v / 255, v / 65535), everything else as f32."""
    a = np.asarray(image)
    if a.ndim != 3 or a.shape[2] not in (3, 4):
        raise ValueError("image must be [H, W, 3] (or RGBA [H, W, 4])")
    a = a[:, :, :3]
    if a.dtype in (np.uint8, np.uint16):
        return np.ascontiguousarray(a)
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- marks (algorithm.rs:596-666) --------------------------------------------------------------
class MarkBuf:
    def __init__(self, data: Optional[Sequence[float]] = None):
        self._data = np.zeros(0, np.float32) if data is None else np.array(data, np.float32).ravel().copy()

    @staticmethod
    def new() -> "MarkBuf":
        return MarkBuf()

    @staticmethod
    def generate_normal(length: int) -> "MarkBuf":
        """algorithm.rs:619-626: StandardNormal from a thread-local, non-deterministic RNG (host side)."""
        return MarkBuf(np.random.default_rng().standard_normal(length).astype(np.float32))

    @staticmethod
    def from_(data) -> "MarkBuf":            # `from` is a Python keyword
        return MarkBuf(data)

    def data(self) -> np.ndarray:
        return self._data

    def set_data(self, data):
        self._data = np.array(data, np.float32).ravel().copy()

    def __len__(self):
        return self._data.size


def _mark_data(m) -> np.ndarray:
    """The `Mark` trait (algorithm.rs:597-600, :659-666): anything exposing a f32 slice."""
    if isinstance(m, MarkBuf):
        return m.data()
    return np.ascontiguousarray(m, dtype=np.float32).ravel()


def _marks_c(marks):
    arrs = [np.ascontiguousarray(_mark_data(m), dtype=np.float32) for m in marks]
    ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * max(len(arrs), 1))(*[a.size for a in arrs])
    return arrs, ptrs, lens


# ---- Writer (algorithm.rs:285-433) -------------------------------------------------------------
class Writer:
    def __init__(self, image, config: Optional[WriteConfig] = None, ctx: Optional[Context] = None):
        """Writer::new (algorithm.rs:295-316): rgb -> yiq, DCT-II of the Y plane on the GPU."""
        self._ctx = ctx or default_context()
        self._lib = self._ctx._lib
        config = config or WriteConfig.default()
        rgb = _as_rgb(image)
        self.height, self.width = rgb.shape[:2]
        h = C.c_void_p()
        cfg = config._c()
        create = {np.dtype(np.uint8): self._lib.ssw_writer_create_rgb8,
                  np.dtype(np.uint16): self._lib.ssw_writer_create_rgb16}.get(rgb.dtype, self._lib.ssw_writer_create)
        check(create(self._ctx.handle, rgb.ctypes.data, self.width, self.height, C.byref(cfg), C.byref(h)), "Writer::new")
        self._h = h

    new = classmethod(lambda cls, image, config=None, ctx=None: cls(image, config, ctx))

    def coefficient_image(self) -> np.ndarray:
        out = np.empty((self.height, self.width), np.float32)
        check(self._lib.ssw_writer_coefficients(self._h, out.ctypes.data), "Writer::coefficient_image")
        return out

    def embed(self, marks):
        arrs, ptrs, lens = _marks_c(marks)
        check(self._lib.ssw_writer_embed(self._h, ptrs, lens, len(arrs)), "Writer::embed")

    def _out(self, out, dtype):
        if out is None:
            return np.empty((self.height, self.width, 3), dtype)
        if out.dtype != dtype or out.shape != (self.height, self.width, 3) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a contiguous {np.dtype(dtype).name} array [H, W, 3]")
        return out

    def result(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Writer::result (algorithm.rs:361-379) -> f32 [H, W, 3]."""
        out = self._out(out, np.float32)
        check(self._lib.ssw_writer_result(self._h, out.ctypes.data), "Writer::result")
        return out

    def result_rgb8(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """`writer.result().into_rgb8()`: quantised on the device, u8 [H, W, 3]."""
        out = self._out(out, np.uint8)
        check(self._lib.ssw_writer_result_rgb8(self._h, out.ctypes.data), "Writer::result")
        return out

    def mark(self, marks, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Writer::mark (algorithm.rs:355-358) -> f32 [H, W, 3]."""
        arrs, ptrs, lens = _marks_c(marks)
        out = self._out(out, np.float32)
        check(self._lib.ssw_writer_mark(self._h, ptrs, lens, len(arrs), out.ctypes.data), "Writer::mark")
        return out

    def mark_rgb8(self, marks, out: Optional[np.ndarray] = None) -> np.ndarray:
        """`writer.mark(marks).into_rgb8()` (examples/main.rs:271-278): u8 [H, W, 3]."""
        arrs, ptrs, lens = _marks_c(marks)
        out = self._out(out, np.uint8)
        check(self._lib.ssw_writer_mark_rgb8(self._h, ptrs, lens, len(arrs), out.ctypes.data), "Writer::mark")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._ctx.handle:
                self._lib.ssw_writer_destroy(self._h)
            self._h = None
        except Exception:
            pass


# ---- Reader (algorithm.rs:435-594) -------------------------------------------------------------
class Reader:
    def __init__(self, image, is_base: bool, config: Optional[ReadConfig], ctx: Optional[Context] = None):
        self._ctx = ctx or default_context()
        self._lib = self._ctx._lib
        rgb = _as_rgb(image)
        self.height, self.width = rgb.shape[:2]
        self.is_base = is_base
        h = C.c_void_p()
        cfg = config._c() if config is not None else None
        create = {np.dtype(np.uint8): self._lib.ssw_reader_create_rgb8,
                  np.dtype(np.uint16): self._lib.ssw_reader_create_rgb16}.get(rgb.dtype, self._lib.ssw_reader_create)
        check(create(self._ctx.handle, rgb.ctypes.data, self.width, self.height, int(is_base),
                     C.byref(cfg) if cfg is not None else None, C.byref(h)), "Reader::new_impl")
        self._h = h

    @staticmethod
    def base(image, config: Optional[ReadConfig] = None, ctx: Optional[Context] = None) -> "Reader":
        """Reader::base (algorithm.rs:462-464)."""
        return Reader(image, True, config or ReadConfig.default(), ctx)

    @staticmethod
    def derived(image, ctx: Optional[Context] = None, precision: int = Precision.F64) -> "ReaderDerived":
        """Reader::derived (algorithm.rs:469-471)."""
        return ReaderDerived(image, ctx, precision)

    def coefficients(self) -> np.ndarray:
        out = np.empty(self.height * self.width, np.float32)
        check(self._lib.ssw_reader_coefficients(self._h, out.ctypes.data), "Reader::coefficients")
        return out

    def indices(self, k: Optional[int] = None) -> np.ndarray:
        """Reader::indices (algorithm.rs:506-508); `k` limits the list to its first k entries."""
        n = self.height * self.width - 1
        k = n if k is None else int(k)
        out = np.empty(max(k, 0), np.uint64)
        check(self._lib.ssw_reader_indices(self._h, k, out.ctypes.data), "Reader::indices")
        return out

    def extract(self, derived: "ReaderDerived", extracted) -> np.ndarray:
        """Reader::extract (algorithm.rs:529-539).  `extracted` is the output buffer (numpy f32,
        written in place) or an int length; the filled array is returned."""
        out = np.empty(int(extracted), np.float32) if isinstance(extracted, (int, np.integer)) else extracted
        if out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("extracted must be a contiguous float32 array")
        d = derived._reader if isinstance(derived, ReaderDerived) else derived
        check(self._lib.ssw_reader_extract(self._h, d._h, out.ctypes.data, out.size), "Reader::extract")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._ctx.handle:
                self._lib.ssw_reader_destroy(self._h)
            self._h = None
        except Exception:
            pass


class ReaderDerived:
    """algorithm.rs:448-456: a Reader that can only be read from."""

    def __init__(self, image, ctx: Optional[Context] = None, precision: int = Precision.F64):
        cfg = ReadConfig(precision=precision)
        self._reader = Reader(image, False, cfg, ctx)

    new = classmethod(lambda cls, image, ctx=None, precision=Precision.F64: cls(image, ctx, precision))

    def coefficients(self) -> np.ndarray:
        return self._reader.coefficients()


# ---- the callers' loops over host images as one streaming call (examples/main.rs:271-278, :383-415) ----------------
def _frame_ptrs(images, w=None, h=None):
    """n 8-bit [H, W, 3] host images (a list, or one [n, H, W, 3] array) -> (kept-alive arrays, void* array, w, h)."""
    arrs = [np.ascontiguousarray(np.asarray(im)[:, :, :3]) for im in images]
    if not arrs:
        raise ValueError("no images")
    for a in arrs:
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape != arrs[0].shape:
            raise ValueError("images must be 8-bit [H, W, 3] arrays of one size")
    hh, ww = arrs[0].shape[:2]
    if (w, h) != (None, None) and (ww, hh) != (w, h):
        raise ValueError("image sizes differ")
    return arrs, (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs]), ww, hh


def mark_many(images, marks, config: Optional[WriteConfig] = None, ctx: Optional[Context] = None, out=None):
    """`for (image, mark) in ..: Writer::new(image, config).mark(&[&mark]).into_rgb8()` (examples/main.rs:271-278) as ONE
    streaming call over n 8-bit host images: ssw_batch_embed_host_rgb8.  marks: [n][k].  Returns / fills `out`, a list
    of n u8 [H, W, 3] arrays (pinned arrays from Context.pinned_empty are the DMA source / target themselves)."""
    ctx = ctx or default_context()
    config = config or WriteConfig.default()
    arrs, ptrs, w, h = _frame_ptrs(images)
    m = np.ascontiguousarray(np.stack([_mark_data(x) for x in marks]), dtype=np.float32)
    if m.ndim != 2 or m.shape[0] != len(arrs):
        raise ValueError("one mark of equal length per image")
    if out is None:
        out = [np.empty((h, w, 3), np.uint8) for _ in arrs]
    out = list(out)
    if len(out) != len(arrs):                   # the C side reads one output pointer per input frame
        raise ValueError("out: one output array per image")
    for o in out:
        if o.dtype != np.uint8 or o.shape != (h, w, 3) or not o.flags.c_contiguous:
            raise ValueError("out: contiguous u8 [H, W, 3] arrays")
    optrs = (C.c_void_p * len(out))(*[o.ctypes.data for o in out])
    cfg = config._c()
    check(ctx._lib.ssw_batch_embed_host_rgb8(ctx.handle, C.byref(cfg), ptrs, len(arrs), w, h, m.ctypes.data, m.shape[1], optrs),
          "ssw_batch_embed_host_rgb8")
    return out


def extract_many(base_images, derived_images, k: int, marks=None, config: Optional[ReadConfig] = None, ctx: Optional[Context] = None):
    """`Reader::base(b, config).extract(&Reader::derived(d), ..)` (+ `Tester::similarity` against marks[i]) per image pair
    (examples/main.rs:383-415) as ONE streaming call: ssw_batch_extract_host_rgb8.  Returns (extracted [n][k], sims [n] or None)."""
    ctx = ctx or default_context()
    config = config or ReadConfig.default()
    ba, bp, w, h = _frame_ptrs(base_images)
    if len(derived_images) != len(ba):          # checked before any pointer array is handed to the C side
        raise ValueError("one derived image per base image")
    da, dp, _, _ = _frame_ptrs(derived_images, w, h)
    n = len(ba)
    ext = np.empty((n, k), np.float32)
    m = sims = None
    if marks is not None:
        m = np.ascontiguousarray(np.stack([_mark_data(x) for x in marks]), dtype=np.float32)
        if m.shape != (n, k):
            raise ValueError("marks must be [n][k]")
        sims = np.empty(n, np.float32)
    cfg = config._c()
    check(ctx._lib.ssw_batch_extract_host_rgb8(ctx.handle, C.byref(cfg), bp, dp, n, w, h, k, ext.ctypes.data,
                                               m.ctypes.data if m is not None else None, sims.ctypes.data if sims is not None else None),
          "ssw_batch_extract_host_rgb8")
    return ext, sims


# ---- Tester (algorithm.rs:668-715) -------------------------------------------------------------
@dataclass
class Similarity:
    similarity: float

    def exceeds_sigma(self, n_sigma: float) -> bool:
        return self.similarity > n_sigma


class Tester:
    def __init__(self, extracted_watermark, ctx: Optional[Context] = None):
        self._e = np.ascontiguousarray(extracted_watermark, dtype=np.float32).ravel()
        self._ctx = ctx or default_context()

    new = classmethod(lambda cls, e, ctx=None: cls(e, ctx))

    def similarity(self, comparison_watermark) -> Similarity:
        m = np.ascontiguousarray(_mark_data(comparison_watermark), dtype=np.float32)
        out = C.c_float()
        check(self._ctx._lib.ssw_similarity(self._ctx.handle, self._e.ctypes.data, self._e.size, m.ctypes.data,
                                            m.size, C.byref(out)), "Tester::similarity")
        return Similarity(float(out.value))
