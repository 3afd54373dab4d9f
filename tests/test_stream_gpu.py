"""Host-image streaming entry points (ssw_batch_embed_host_rgb8 / ssw_batch_extract_host_rgb8, csrc/ssw_stream.hip): the loops
of the reference's callers over images on the host -- examples/main.rs:271-278 (`watermark`) and :383-415 (`test`) -- as one
call each, groups of frames crossing PCIe beside the kernels of their neighbours.  Bit-identical to the single-image handles
(src/algorithm.rs:295-316, :355-379, :462-500, :529-593, :696-714) for pageable and pinned buffers, any group size, ragged
last groups; the oracle checks one frame end to end."""
import numpy as np
import pytest

import gpu_util as G
import spread_spectrum_watermarking_amd as wm
from oracle import oracle as O
from spread_spectrum_watermarking_amd import _lib as L

pytestmark = pytest.mark.gpu


def _frames(n, w, h, seed=3):
    return [O.f32_to_u8(O.synth_frame(seed + i, 0, w, h)) for i in range(n)]


def _handles(ctx, frames, marks, k):
    marked, ext, sims = [], [], []
    for f, m in zip(frames, marks):
        marked.append(wm.Writer(f, ctx=ctx).mark_rgb8([m]))
    for f, d, m in zip(frames, marked, marks):
        e = wm.Reader.base(f, ctx=ctx).extract(wm.Reader.derived(d, ctx), k)
        ext.append(e)
        sims.append(np.float32(wm.Tester(e, ctx).similarity(m).similarity))
    return marked, np.stack(ext), np.array(sims, np.float32)


@pytest.mark.parametrize("shape,n,group", [((512, 288), 11, 4), ((1024, 272), 5, 2), ((208, 80), 9, 8), ((100, 75), 3, 1), ((640, 384), 19, 8)])
def test_streaming_calls_equal_a_loop_over_the_handles(shape, n, group, monkeypatch):
    w, h = shape
    k = 120
    ctx = wm.Context(0)
    frames = _frames(n, w, h)
    marks = np.random.default_rng(9).standard_normal((n, k)).astype(np.float32)
    ref_marked, ref_ext, ref_sims = _handles(ctx, frames, marks, k)
    monkeypatch.setenv("SSW_STREAM_GROUP", str(group))      # frames per group (default: >= 8)
    got = wm.mark_many(frames, marks, ctx=ctx)
    for a, b in zip(got, ref_marked):
        assert np.array_equal(a, b)
    ext, sims = wm.extract_many(frames, got, k, marks, ctx=ctx)
    assert np.array_equal(ext, ref_ext) and np.array_equal(sims, ref_sims)
    ext2, none = wm.extract_many(frames, got, k, None, ctx=ctx)
    assert none is None and np.array_equal(ext2, ref_ext)
    # the oracle on frame 0 (8-bit flow: embed -> into_rgb8 -> extract)
    f32 = O.u8_to_f32(frames[0])
    o_marked8 = O.f32_to_u8(O.embed_frame(f32, marks[0]))
    assert np.mean(got[0] == o_marked8) >= 0.9999
    if np.array_equal(got[0], o_marked8):
        o_ext, o_sim = O.extract_frame(f32, O.u8_to_f32(o_marked8), marks[0])
        assert np.abs(ext[0] - o_ext).max() <= 1e-5 * max(1.0, float(np.abs(o_ext).max()))
        assert abs(float(sims[0]) - o_sim) < 1e-4 * max(1.0, abs(o_sim))
    ctx.close()


def test_streaming_from_and_into_pinned_buffers():
    """Pinned buffers are the DMA source / target (nothing staged); mixed with pageable ones in one call."""
    w, h, n, k = 768, 432, 13, 200
    ctx = wm.Context(0)
    frames = _frames(n, w, h, seed=21)
    marks = np.random.default_rng(2).standard_normal((n, k)).astype(np.float32)
    ref = wm.mark_many(frames, marks, ctx=ctx)
    pin_in = [ctx.pinned_empty((h, w, 3), np.uint8) for _ in range(n)]
    for p, f in zip(pin_in, frames):
        p[...] = f
    pin_out = [ctx.pinned_empty((h, w, 3), np.uint8) if i % 3 else np.empty((h, w, 3), np.uint8) for i in range(n)]
    ctx.transfer_stats(reset=True)
    out = wm.mark_many(pin_in, marks, ctx=ctx, out=pin_out)
    st = ctx.transfer_stats()
    for a, b in zip(out, ref):
        assert np.array_equal(a, b)
    assert st["direct_bytes"] >= n * w * h * 3                       # every input frame was its own DMA source
    e1, s1 = wm.extract_many(pin_in, out, k, marks, ctx=ctx)
    e2, s2 = wm.extract_many(frames, ref, k, marks, ctx=ctx)
    assert np.array_equal(e1, e2) and np.array_equal(s1, s2)
    ctx.close()


def test_streaming_argument_errors():
    ctx = wm.Context(0)
    f = _frames(2, 64, 48)
    m = np.zeros((2, 10), np.float32)
    with pytest.raises(wm.SswError) as e:
        wm.extract_many(f, f, 64 * 48, None, ctx=ctx)                # k >= W * H: algorithm.rs:553-555
    assert e.value.status == L.SSW_ERR_K_TOO_LARGE
    with pytest.raises(ValueError):
        wm.mark_many(f, m[:1], ctx=ctx)
    with pytest.raises(ValueError):
        wm.mark_many([f[0], f[1][:40]], m, ctx=ctx)
    ctx.close()
