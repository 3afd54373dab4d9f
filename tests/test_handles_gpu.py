"""The single-image handles behind wm::Writer::mark() / Reader::extract (src/algorithm.rs:295-316, :355-379,
:474-480, :529-539; examples/main.rs:271-278) as a host caller uses them: 8-bit or f32 host frames in, host frames
out, through the pinned staging ring or straight from pinned memory, planes pooled in the context.  Everything
goes through the C ABI; the oracle is the checker."""
import ctypes as C

import numpy as np
import pytest

import gpu_util as G
import spread_spectrum_watermarking_amd as wm
from oracle import oracle as O
from spread_spectrum_watermarking_amd import _lib as L

pytestmark = pytest.mark.gpu
F32, F64 = L.PRECISION_F32, L.PRECISION_F64
from conftest import ALL_STRATEGIES  # noqa: E402
PRECISIONS = [F32, F64] if ALL_STRATEGIES else [F64]      # f32: the diagnostic build's operand-ready twin (conftest.py)


def _frame8(seed, w, h):
    return O.f32_to_u8(O.synth_frame(seed, 0, w, h))


@pytest.mark.parametrize("shape", [(192, 108), (208, 80), (1040, 144), (100, 75), (37, 64)])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_rgb8_handles_equal_batch_entry_points_and_f32_handles(shape, precision):
    """ssw_writer_create_rgb8 / ssw_writer_mark_rgb8 / ssw_reader_create_rgb8 (u8 host frames straight through)
    are bit-identical to ssw_batch_*_rgb8 with one frame and to the f32 handles fed with the host-converted frame
    (`into_rgb32f()` = v / 255, `into_rgb8()` = round(clamp * 255)); shapes with and without the fused chain."""
    w, h = shape
    k = 150
    ctx = G.ctx()
    img8 = _frame8(11, w, h)
    img32 = O.u8_to_f32(img8)
    mark = np.random.default_rng(5).standard_normal(k).astype(np.float32)
    cfg_w, cfg_r = wm.WriteConfig(precision=precision), wm.ReadConfig(precision=precision)

    w8 = wm.Writer(img8, cfg_w, ctx)
    w32 = wm.Writer(img32, cfg_w, ctx)
    coef = w8.coefficient_image()
    assert np.array_equal(coef, w32.coefficient_image())
    # the unfused primitives (colour conversion, then ssw_dct2d) give the same coefficients as the fused chain
    y = G.rgb_to_yiq(img32, with_iq=False)[0][0]
    assert np.array_equal(coef, G.dct2d(y, L.DCT2, precision))
    marked8 = w8.mark_rgb8([mark])
    marked32 = w32.mark([mark])
    assert marked8.dtype == np.uint8 and np.array_equal(marked8, O.f32_to_u8(marked32))
    assert np.array_equal(marked8, G.batch_embed_rgb8(img8[None], mark[None], G.default_config(precision))[0])
    # f32 writer, 8-bit result; 8-bit writer, f32 result
    assert np.array_equal(wm.Writer(img32, cfg_w, ctx).mark_rgb8([mark]), marked8)
    assert np.array_equal(wm.Writer(img8, cfg_w, ctx).mark([mark]), marked32)
    wr = wm.Writer(img8, cfg_w, ctx)
    wr.embed([mark])
    assert np.array_equal(wr.result_rgb8(), marked8)
    with pytest.raises(wm.SswError) as e:
        wr.result_rgb8()
    assert e.value.status == L.SSW_ERR_CONSUMED

    base = wm.Reader.base(img8, cfg_r, ctx)
    derived = wm.Reader.derived(marked8, ctx, precision)
    assert np.array_equal(base.coefficients().reshape(h, w), coef)
    ext = base.extract(derived, k)
    sim = wm.Tester(ext, ctx).similarity(mark).similarity
    e_b, s_b = G.batch_extract_rgb8(img8[None], marked8[None], k, mark[None], G.default_config(precision))
    assert np.array_equal(ext, e_b[0]) and np.float32(sim) == s_b[0]
    e32 = wm.Reader.base(img32, cfg_r, ctx).extract(wm.Reader.derived(O.u8_to_f32(marked8), ctx, precision), k)
    assert np.array_equal(ext, e32)
    if precision == F64:
        o_marked8 = O.f32_to_u8(O.embed_frame(img32, mark))
        assert np.mean(marked8 == o_marked8) >= 0.9999
        o_ext, o_sim = O.extract_frame(img32, O.u8_to_f32(o_marked8), mark)
        if np.array_equal(marked8, o_marked8):
            assert np.abs(ext - o_ext).max() <= 1e-5 * max(1.0, float(np.abs(o_ext).max()))
            assert abs(sim - o_sim) < 1e-4 * max(1.0, abs(o_sim))


@pytest.mark.parametrize("shape", [(256, 144), (1024, 272), (208, 80), (100, 75), (37, 64)])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_rgb16_entry_points_equal_the_f32_ones_on_host_converted_frames(shape, precision):
    """SURVEY 8(f) rank 2, the 16-bit half: `into_rgb32f()` of an ImageRgb16 (src/algorithm.rs:308, :476: v / 65535) runs
    on the device -- ssw_writer_create_rgb16 / ssw_reader_create_rgb16 / ssw_batch_*_rgb16 are bit-identical to the f32
    entry points fed with the oracle-converted frame (deep, two-level and unfused shapes), and agree with the oracle's
    own pipeline on that frame."""
    w, h = shape
    k = 150
    ctx = G.ctx()
    img32 = O.synth_frame(13, 0, w, h)
    img16 = O.f32_to_u16(img32)
    conv = O.u16_to_f32(img16)
    assert np.array_equal(conv, (img16.astype(np.float32) / np.float32(65535)))
    assert np.array_equal(G.convert_rgb16(values_u16=img16), conv)                       # ssw_convert_rgb16_to_f32
    assert np.array_equal(G.convert_rgb16(values_f32=img32 * 1.5 - 0.2), O.f32_to_u16(img32 * 1.5 - 0.2))   # clamp + round
    mark = np.random.default_rng(6).standard_normal(k).astype(np.float32)
    cfg_w, cfg_r = wm.WriteConfig(precision=precision), wm.ReadConfig(precision=precision)

    w16, w32 = wm.Writer(img16, cfg_w, ctx), wm.Writer(conv, cfg_w, ctx)
    coef = w16.coefficient_image()
    assert np.array_equal(coef, w32.coefficient_image())
    marked = w16.mark([mark])
    assert np.array_equal(marked, w32.mark([mark]))
    cfg = G.default_config(precision)
    assert np.array_equal(G.batch_embed_rgb16(img16[None], mark[None], cfg)[0], marked)
    assert np.array_equal(G.batch_embed(conv[None], mark[None], cfg)["rgb"][0], marked)
    # the marked frame saved as a 16-bit image and read back: Reader::base / derived on u16 frames
    marked16 = G.convert_rgb16(values_f32=marked)
    assert np.array_equal(marked16, O.f32_to_u16(marked))
    base, derived = wm.Reader.base(img16, cfg_r, ctx), wm.Reader.derived(marked16, ctx, precision)
    assert np.array_equal(base.coefficients().reshape(h, w), coef)
    ext = base.extract(derived, k)
    sim = wm.Tester(ext, ctx).similarity(mark).similarity
    e_b, s_b = G.batch_extract_rgb16(img16[None], marked16[None], k, mark[None], cfg)
    assert np.array_equal(ext, e_b[0]) and np.float32(sim) == s_b[0]
    e32 = wm.Reader.base(conv, cfg_r, ctx).extract(wm.Reader.derived(O.u16_to_f32(marked16), ctx, precision), k)
    assert np.array_equal(ext, e32)
    if precision == F64:
        o_marked = O.embed_frame(conv, mark)
        assert np.abs(marked - o_marked).max() <= 2e-7
        o_marked16 = O.f32_to_u16(o_marked)
        assert np.mean(marked16 == o_marked16) >= 0.999
        if np.array_equal(marked16, o_marked16):
            o_ext, o_sim = O.extract_frame(conv, O.u16_to_f32(o_marked16), mark)
            assert np.abs(ext - o_ext).max() <= 1e-5 * max(1.0, float(np.abs(o_ext).max()))
            assert abs(sim - o_sim) < 1e-4 * max(1.0, abs(o_sim))


def test_rgb8_handles_4k_against_the_oracle():
    """configs[1]-sized frame through the 8-bit handles (25 MB each way: the staged path with its copy threads):
    `Writer::new(img).mark(&[&mark]).into_rgb8()` then `Reader::base / derived / extract` and `Tester::similarity`
    against the oracle's exact pipeline on the same bytes."""
    w, h, k = 3840, 2160, 1000
    ctx = G.ctx()
    img8 = O.f32_to_u8(G.synth(3, 5, 1, w, h)[0])
    mark = np.random.default_rng(21).standard_normal(k).astype(np.float32)
    ctx.transfer_stats(reset=True)
    marked8 = wm.Writer(img8, ctx=ctx).mark_rgb8([mark])
    base = wm.Reader.base(img8, ctx=ctx)
    ext = base.extract(wm.Reader.derived(marked8, ctx), k)
    sim = wm.Tester(ext, ctx).similarity(mark).similarity
    st = ctx.transfer_stats()
    assert st["h2d_bytes"] >= 3 * img8.nbytes and st["h2d_bytes"] < 3 * img8.nbytes + (1 << 20)    # 3 B/px in, three times
    assert img8.nbytes <= st["d2h_bytes"] < img8.nbytes + (1 << 20)
    assert st["staged_bytes"] >= 4 * img8.nbytes                                                    # pageable numpy buffers
    img32 = O.u8_to_f32(img8)
    o_marked8 = O.f32_to_u8(O.embed_frame(img32, mark))
    assert np.mean(marked8 == o_marked8) >= 0.9999
    assert np.array_equal(base.indices(k), O.indices(O.dct2d(O.rgb_to_yiq(img32)[0]), k=k))
    o_ext, o_sim = O.extract_frame(img32, O.u8_to_f32(marked8), mark)        # from the GPU's own 8-bit frame
    assert np.abs(ext - o_ext).max() <= 1e-5 * max(1.0, float(np.abs(o_ext).max()))
    assert abs(sim - o_sim) < 1e-4 * abs(o_sim)


def test_pinned_host_buffers_are_the_dma_source_and_target():
    """ssw_host_alloc'd (pinned) frames: no staging copy (transfer stats say "direct"), same results."""
    w, h, k = 1920, 1080, 400
    ctx = G.ctx()
    img8 = _frame8(4, w, h)
    mark = np.random.default_rng(2).standard_normal(k).astype(np.float32)
    ref = wm.Writer(img8, ctx=ctx).mark_rgb8([mark])
    pin_in = ctx.pinned_empty(img8.shape, np.uint8)
    pin_out = ctx.pinned_empty(img8.shape, np.uint8)
    pin_in[...] = img8
    ctx.transfer_stats(reset=True)
    out = wm.Writer(pin_in, ctx=ctx).mark_rgb8([mark], out=pin_out)
    st = ctx.transfer_stats()
    assert out is pin_out and np.array_equal(pin_out, ref)
    assert st["direct_bytes"] == 2 * img8.nbytes and st["staged_bytes"] == 0
    ext_pin = wm.Reader.base(pin_in, ctx=ctx).extract(wm.Reader.derived(pin_out, ctx), k)
    ext_ref = wm.Reader.base(img8, ctx=ctx).extract(wm.Reader.derived(ref, ctx), k)
    assert np.array_equal(ext_pin, ext_ref)
    del pin_in, pin_out, out


@pytest.mark.parametrize("threads", [1, 2, 5])
def test_staged_transfers_are_exact_for_every_thread_count_and_size(threads):
    """ssw_copy_to_dev / ssw_copy_to_host through the staging ring: sizes around the 1 MiB piece, the 4 MiB slice,
    the 64 MiB staging buffer and the three-buffer round; the bytes must come back unchanged."""
    ctx = wm.Context(0)
    try:
        ctx.set_copy_threads(threads)
        rng = np.random.default_rng(threads)
        for nbytes in (1, 255 << 10, (256 << 10) + 3, (1 << 20) - 1, (4 << 20) + 5, (64 << 20) - 7, (64 << 20) + 4097,
                       (200 << 20) + 12345):
            a = rng.integers(0, 256, nbytes, dtype=np.uint8)
            d = ctx.to_device(a)
            back = d.to_host(np.uint8, a.shape)
            d.free()
            assert np.array_equal(a, back), nbytes
    finally:
        ctx.close()


def test_handle_planes_are_pooled_in_the_context():
    """Destroyed handles hand their planes to the context; the next handle of that size takes them instead of
    hipMalloc: device memory in use does not grow over many create / destroy rounds, and ssw_ctx_destroy releases it."""
    probe = G.ctx()
    w, h, k = 512, 288, 64
    img8 = _frame8(9, w, h)
    mark = np.random.default_rng(0).standard_normal(k).astype(np.float32)
    # what the HIP runtime allocates on the first launches of a process (code objects, kernel-argument pools) is not the
    # context's: one round on a throw-away context first, whatever ran before this test
    warm = wm.Context(0)
    wb = wm.Reader.base(img8, ctx=warm)
    wb.extract(wm.Reader.derived(wm.Writer(img8, ctx=warm).mark_rgb8([mark]), warm), k)
    del wb
    warm.close()
    free_start, _ = probe.mem_info()
    ctx = wm.Context(0)
    first = wm.Writer(img8, ctx=ctx).mark_rgb8([mark])
    b = wm.Reader.base(img8, ctx=ctx)
    e0 = b.extract(wm.Reader.derived(first, ctx), k)
    del b
    free_after_first, _ = probe.mem_info()
    for _ in range(40):
        assert np.array_equal(wm.Writer(img8, ctx=ctx).mark_rgb8([mark]), first)
        b = wm.Reader.base(img8, ctx=ctx)
        assert np.array_equal(b.extract(wm.Reader.derived(first, ctx), k), e0)
        del b
    free_after_many, _ = probe.mem_info()
    assert free_after_first - free_after_many < (8 << 20)
    # two handles alive at once get distinct planes
    w1, w2 = wm.Writer(img8, ctx=ctx), wm.Writer(img8[::-1].copy(), ctx=ctx)
    c1, c2 = w1.coefficient_image(), w2.coefficient_image()
    assert not np.array_equal(c1, c2) and np.array_equal(c1, wm.Writer(img8, ctx=ctx).coefficient_image())
    del w1, w2
    ctx.close()
    free_end, _ = probe.mem_info()
    assert free_start - free_end < (64 << 20)


def test_handles_in_flight_do_not_disturb_each_other():
    """Constructors only enqueue: several readers and writers created back to back (their uploads on the copy
    stream, two alternating device staging buffers) must each see their own frame."""
    ctx = G.ctx()
    w, h, k = 640, 360, 100
    frames = [_frame8(s, w, h) for s in range(5)]
    mark = np.random.default_rng(1).standard_normal(k).astype(np.float32)
    want = [wm.Writer(f, ctx=ctx).coefficient_image() for f in frames]
    writers = [wm.Writer(f, ctx=ctx) for f in frames]
    readers = [wm.Reader.base(f, ctx=ctx) for f in frames]
    for wr, rd, c in zip(writers, readers, want):
        assert np.array_equal(rd.coefficients().reshape(h, w), c)
        assert np.array_equal(wr.coefficient_image(), c)
    marked = [wr.mark_rgb8([mark]) for wr in writers]
    for f, m, rd in zip(frames, marked, readers):
        e = rd.extract(wm.Reader.derived(m, ctx), k)
        assert wm.Tester(e, ctx).similarity(mark).similarity > 0.5 * np.linalg.norm(mark)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_derived_reader_is_transformed_on_use_and_only_where_it_is_read(precision):
    """Reader::derived (src/algorithm.rs:469-480) only uploads; Reader::extract (:529-562) then transforms the
    frequency columns the base reader's first k indices use (the batch path's pruned transform, n = 1).  Same values
    as the full transform (forced here through coefficients()), far fewer flop; a frame whose columns do not fit
    (white noise) falls back to the full transform; a second, longer extract and a second base reader still work."""
    ctx = G.ctx()
    w, h, k = 1040, 144, 120
    cfg = wm.ReadConfig(precision=precision)
    img = O.synth_frame(8, 2, w, h)
    mark = np.random.default_rng(3).standard_normal(k).astype(np.float32)
    marked = wm.Writer(img, wm.WriteConfig(precision=precision), ctx).mark([mark])
    base = wm.Reader.base(img, cfg, ctx)
    base.indices(k)                                        # ordering done: the timers below see the derived side only
    full = wm.Reader.derived(marked, ctx, precision)
    full.coefficients()                                    # forces Reader::derived's full transform
    ctx.enable_timing(True)
    try:
        ctx.reset_timing()
        want = base.extract(full, k)
        assert ctx.timing()["dct_row"]["launches"] == 0    # already transformed
        lazy = wm.Reader.derived(marked, ctx, precision)
        ctx.reset_timing()
        got = base.extract(lazy, k)
        pruned_flop = ctx.timing()["dct_row"]["work"]
        ctx.reset_timing()
        coef = lazy.coefficients()                         # the same handle can still produce all coefficients
        full_flop = ctx.timing()["dct_row"]["work"]
    finally:
        ctx.enable_timing(False)
    assert np.array_equal(got, want)
    assert 0 < pruned_flop < 0.5 * full_flop
    assert np.array_equal(coef, full.coefficients())
    assert np.array_equal(base.extract(lazy, k + 50), base.extract(full, k + 50))
    other = wm.Reader.base(O.synth_frame(8, 3, w, h), cfg, ctx)
    assert np.array_equal(other.extract(wm.Reader.derived(marked, ctx, precision), k), other.extract(full, k))
    # white noise: the index list touches nearly every column -> overflow -> full transform, same values
    noise = np.random.default_rng(9).random((h, w, 3), dtype=np.float32)
    nb = wm.Reader.base(noise, cfg, ctx)
    nd_full = wm.Reader.derived(noise[::-1].copy(), ctx, precision)
    nd_full.coefficients()
    assert np.array_equal(nb.extract(wm.Reader.derived(noise[::-1].copy(), ctx, precision), 1000), nb.extract(nd_full, 1000))
    # error behaviour is unchanged: a derived reader is not a base, shapes must match
    with pytest.raises(wm.SswError) as e:
        lazy._reader.indices(3)
    assert e.value.status == L.SSW_ERR_NOT_BASE
    with pytest.raises(wm.SswError) as e:
        base.extract(wm.Reader.derived(O.synth_frame(1, 1, w, h + 8), ctx, precision), k)
    assert e.value.status == L.SSW_ERR_LENGTH_MISMATCH


def test_second_embed_and_multi_mark_still_follow_the_reference():
    """Writer::embed twice (the staging vectors of the handle are reused) and several marks of different
    lengths, on an 8-bit writer: equal to the f32 writer fed with the host-converted frame."""
    ctx = G.ctx()
    w, h = 256, 144
    img8 = _frame8(6, w, h)
    rng = np.random.default_rng(4)
    m1, m2, m3 = (rng.standard_normal(n).astype(np.float32) for n in (50, 120, 80))
    a, b = wm.Writer(img8, ctx=ctx), wm.Writer(O.u8_to_f32(img8), ctx=ctx)
    for wr in (a, b):
        wr.embed([m1, m2])
        wr.embed([m3])
    assert np.array_equal(a.coefficient_image(), b.coefficient_image())
    assert np.array_equal(a.result_rgb8(), O.f32_to_u8(b.result()))


def test_context_creation_keeps_the_callers_device():
    """ssw_ctx_create restores the calling thread's current HIP device; a bad ordinal is an argument error."""
    L.load()
    # the HIP runtime instance libssw_hip.so is linked against (dlopen of an already loaded path returns that instance;
    # a bare "libamdhip64.so" could resolve to a second copy, e.g. torch's, with its own idea of the current device)
    paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l})
    hip = C.CDLL(paths[0] if paths else "libamdhip64.so")
    dev = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(dev)) == 0
    before = dev.value
    c = wm.Context(0)
    c.close()
    assert hip.hipGetDevice(C.byref(dev)) == 0 and dev.value == before
    n = C.c_int(0)
    assert hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value >= 1
    bad = C.c_void_p()
    assert L.load().ssw_ctx_create(n.value, C.byref(bad)) == L.SSW_ERR_BAD_ARG
