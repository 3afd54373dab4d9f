"""GPU tests of the host-side machinery around the kernels: context lifetime, the two-lane / two-stream
batch pipelines, the pruned transform of derived frames, stream hand-off, error mapping, and the
multi-rank wiring of bench.py.  Everything goes through the C ABI; the oracle is the checker."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gpu_util as G
from conftest import ROOT
from oracle import oracle as O
from spread_spectrum_watermarking_amd import _lib as L
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd.api import check

pytestmark = pytest.mark.gpu
F32, F64 = L.PRECISION_F32, L.PRECISION_F64
from conftest import ALL_STRATEGIES  # noqa: E402
PRECISIONS = [F32, F64] if ALL_STRATEGIES else [F64]      # f32: the diagnostic build's operand-ready twin (conftest.py)


# ---- context lifetime / error mapping ------------------------------------------------------------------
_LEAK_SCRIPT = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check
probe = wm.Context(0)
rgb = np.random.default_rng(0).random((6, 288, 512, 3)).astype(np.float32)
marks = np.random.default_rng(0).standard_normal((6, 100)).astype(np.float32)
cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64)

def cycle():
    ctx = wm.Context(0)
    ctx.set_chunk_frames(2)                      # three passes: both lanes, the pruned path, every workspace buffer
    lib = ctx._lib
    d, dm = ctx.to_device(rgb), ctx.to_device(marks)
    out, ext, sims = ctx.alloc(rgb.nbytes), ctx.alloc(6 * 100 * 4), ctx.alloc(6 * 4)
    check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg), d.ptr, 6, 512, 288, dm.ptr, 100, out.ptr, None, None), "embed")
    check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), d.ptr, out.ptr, 6, 512, 288, 100, ext.ptr, dm.ptr, sims.ptr), "extract")
    ctx.synchronize()
    for b in (d, dm, out, ext, sims):
        b.free()
    ctx.close()

cycle()                                          # first cycle: code objects, runtime pools
free0, _ = probe.mem_info()
for _ in range(3):
    cycle()
free1, _ = probe.mem_info()
for _ in range(3):
    cycle()
free2, _ = probe.mem_info()
# a leak repeats with every cycle; a one-time allocation of the runtime (its pools grow on first use) shows in one window only
print("leaked_bytes", min(free0 - free1, free1 - free2), "windows", free0 - free1, free1 - free2)
"""


def test_context_create_destroy_releases_device_memory(tmp_path):
    """ssw_ctx_destroy frees every workspace buffer (operand planes, lanes, selection, bases, compact planes): three
    create / batch embed + extract / destroy cycles leave the device's free memory where it was.  In a fresh
    process, so that other tests' contexts and the allocator's history do not blur the reading.  Two windows of three
    cycles each, the smaller difference counts: what leaks per cycle shows in both, what the runtime allocates once for
    itself (seen once on a fresh box in r5: the first window read >= 8 MiB, a repeat read 0) shows in one."""
    script = tmp_path / "leak_check.py"
    script.write_text(_LEAK_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    # no retry: a child that dies (rc != 0) is a failure of context teardown, whatever came before it
    assert r.returncode == 0 and "leaked_bytes" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    leaked = int(r.stdout.split("leaked_bytes")[1].split()[0])
    assert leaked < 8 << 20, f"device memory leaked across context cycles: {r.stdout[-300:]}"


def test_alloc_failure_is_out_of_memory_and_recoverable():
    ctx = G.ctx()
    p = C.c_void_p()
    _, total = ctx.mem_info()
    assert ctx._lib.ssw_dev_alloc(ctx.handle, total * 4, C.byref(p)) == L.SSW_ERR_OUT_OF_MEMORY
    assert b"hipMalloc" in ctx._lib.ssw_last_error()
    buf = ctx.alloc(1024)                     # the runtime's error state was cleared
    buf.free()


def test_automatic_pass_size_shrinks_when_device_memory_is_short():
    """the automatic pass size (2^30 pixels, 36 B/px of workspace per lane) is clamped to half of what the
    device can give -- free memory plus what the context already holds -- so a nearly full device gets smaller passes
    instead of SSW_ERR_OUT_OF_MEMORY, with bit-identical results."""
    w, h, n, k = 1920, 1080, 60, 200
    ctx = wm.Context(0)
    lib = ctx._lib
    try:
        rgb = G.synth(6, 0, n, w, h)
        marks = np.random.default_rng(6).standard_normal((n, k)).astype(np.float32)
        cfg = G.default_config()
        d, dm = ctx.to_device(rgb), ctx.to_device(marks)
        out = ctx.alloc(rgb.nbytes)

        def run():
            check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg), d.ptr, n, w, h, dm.ptr, k, out.ptr, None, None), "embed")
            return out.to_host(np.float32, rgb.shape)
        assert ctx.pass_frames(n, w, h) == n                      # plenty of memory: one pass
        want = run()
        free, _ = ctx.mem_info()
        hog = ctx.alloc(free - (3 << 30))                         # leave ~3 GB + the ~4.7 GB workspace the context holds
        try:
            short = ctx.pass_frames(n, w, h)
            assert 1 <= short < n                                 # (3 + 4.7) / 2 GB over 36 B/px of 1080p frames: ~50
            got = run()
        finally:
            hog.free()
        assert np.array_equal(got, want)
        for b in (d, dm, out):
            b.free()
    finally:
        ctx.close()


def test_second_embed_ranks_the_original_coefficients():
    """Writer::new fixes the ordering once (algorithm.rs:314); embed() twice must not re-rank the
    modified plane."""
    rgb = O.synth_frame(3, 0, 160, 96)
    rng = np.random.default_rng(7)
    m1 = (rng.standard_normal(50) * 4).astype(np.float32)          # strong marks: re-ranking would differ
    m2 = (rng.standard_normal(80) * 4).astype(np.float32)
    w = wm.Writer(rgb, wm.WriteConfig(insertion=wm.Insertion.Option2(0.5)))
    c0 = w.coefficient_image()
    idx = O.indices(c0, k=80)
    w.embed([m1])
    c1 = w.coefficient_image()
    assert np.array_equal(c1, O.embed(c0, idx[:50], [m1], O.OPTION2, 0.5))
    w.embed([m2])
    c2 = w.coefficient_image()
    assert np.array_equal(c2, O.embed(c1, idx, [m2], O.OPTION2, 0.5))
    assert not np.array_equal(O.indices(c1, k=80), idx)            # the test would notice a re-rank


def test_batch_embed_truncates_a_long_mark_like_zip():
    """algorithm.rs:396: a mark longer than w*h-1 is cut silently -- also in the batch entry points."""
    rgb = np.random.default_rng(1).random((2, 3, 4, 3)).astype(np.float32)       # 12 coefficients, 11 usable
    marks = np.random.default_rng(2).standard_normal((2, 40)).astype(np.float32)
    res = G.batch_embed(rgb, marks)
    for f in range(2):
        assert np.array_equal(res["rgb"][f], wm.Writer(rgb[f]).mark([marks[f]]))


_STREAM_SCRIPT = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
dev = torch.device("cuda", 0)
x = torch.rand((3, 72, 136), device=dev)          # torch initialises HIP first (its bundled runtime and the
torch.cuda.synchronize()                           # library's then share one ROCr, as in bench.py)
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check
ctx = wm.Context(0)
F64 = L.PRECISION_F64
ref = x.clone()
check(ctx._lib.ssw_dct2d(ctx.handle, L.DCT2, F64, 3, 136, 72, ref.data_ptr()), "ssw_dct2d")
ctx.synchronize()
ref = ref.cpu().numpy()
# (1) the library on the caller's stream: producer -> library -> consumer, no host synchronisation in between
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    y = x * 1.0
    ctx.set_stream(s.cuda_stream)
    check(ctx._lib.ssw_dct2d(ctx.handle, L.DCT2, F64, 3, 136, 72, y.data_ptr()), "ssw_dct2d")
    z = y + 0.0
s.synchronize()
assert np.array_equal(z.cpu().numpy(), ref)
ctx.set_stream(None)
# (2) private stream, chained through the caller's events
y2 = x.clone()
produced, done = torch.cuda.Event(), torch.cuda.Event()
produced.record()
ctx.wait_event(produced.cuda_event)
check(ctx._lib.ssw_dct2d(ctx.handle, L.DCT2, F64, 3, 136, 72, y2.data_ptr()), "ssw_dct2d")
done.record()                                      # creates the handle; re-recorded on the context's stream next
ctx.record_event(done.cuda_event)
done.synchronize()
assert np.array_equal(y2.cpu().numpy(), ref)
assert int(ctx._lib.ssw_ctx_stream(ctx.handle) or 0) != s.cuda_stream
ctx.close()
print("stream-ok")
"""


def test_caller_stream_and_events(tmp_path):
    """ssw_ctx_set_stream / wait_event / record_event: chaining with a torch stream without host
    synchronisation.  In a fresh interpreter, so that torch brings up HIP before the library does."""
    script = tmp_path / "stream_check.py"
    script.write_text(_STREAM_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "stream-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


_GRAPH_SCRIPT = r"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import spread_spectrum_watermarking_amd as wm
from spread_spectrum_watermarking_amd import _lib as L
from spread_spectrum_watermarking_amd.api import check
lib = L.load(); ctx = wm.Context(0)
n, w, h, k = 4, 512, 288, 100
dev = torch.device("cuda", 0)
rgb = torch.empty((n, h, w, 3), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
check(lib.ssw_synth_frames(ctx.handle, 3, 0, n, w, h, rgb.data_ptr()), "synth"); ctx.synchronize()
marks = torch.randn((n, k), device=dev); out = torch.empty_like(rgb)
ext = torch.zeros((n, k), device=dev); sims = torch.zeros((n,), device=dev)
cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64)
def step():
    check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg), rgb.data_ptr(), n, w, h, marks.data_ptr(), k, out.data_ptr(), None, None), "embed")
    check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), rgb.data_ptr(), out.data_ptr(), n, w, h, k, ext.data_ptr(), marks.data_ptr(), sims.data_ptr()), "extract")
torch.cuda.synchronize()
step(); ctx.synchronize()                      # eager: workspaces sized, bases cached
want = (out.clone(), ext.clone(), sims.clone())
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)
out.zero_(); ext.zero_(); sims.zero_(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
    step()
pruned_before = ctx.prune_stats()["pruned_chunks"]
g.replay(); torch.cuda.synchronize()
assert torch.equal(out, want[0]) and torch.equal(ext, want[1]) and torch.equal(sims, want[2])
assert pruned_before == 1          # the eager call took the pruned path; the captured one (no host round trip allowed) did not
print("graph-ok")
ctx.set_stream(None); ctx.close()
"""


def test_batch_calls_can_be_captured_into_a_graph(tmp_path):
    """The device-pointer entry points only enqueue: with the context on a caller's stream, ssw_batch_embed +
    ssw_batch_extract can be captured into a HIP graph (torch.cuda.CUDAGraph) once their workspaces exist; during
    capture the extract takes the full derived transform instead of the pruned one, whose overflow check needs a
    host round trip.  Replaying the graph reproduces the eager results bit for bit."""
    script = tmp_path / "graph_check.py"
    script.write_text(_GRAPH_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "graph-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# ---- two lanes / two streams ---------------------------------------------------------------------------------
def _run_batch(rgb, marks, cfg, overlap, prune, chunk, u8=False):
    ctx = G.ctx()
    ctx.set_overlap(overlap)
    ctx.set_prune(prune)
    ctx.set_chunk_frames(chunk)
    try:
        k = marks.shape[1]
        if u8:
            wm8 = G.batch_embed_rgb8(rgb, marks, cfg)
            ext, sims = G.batch_extract_rgb8(rgb, wm8, k, marks, cfg)
            return wm8, None, None, ext, sims
        res = G.batch_embed(rgb, marks, cfg, want_coef=True, want_idx=True)
        ext, sims = G.batch_extract(rgb, res["rgb"], k, marks, cfg)
        return res["rgb"], res["coef"], res["idx"], ext, sims
    finally:
        ctx.set_overlap(True)
        ctx.set_prune(True)
        ctx.set_chunk_frames(0)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("shape", [(108, 192), (144, 1040), (90, 160)])
def test_overlapped_pipeline_is_bit_identical_to_serial(precision, shape):
    """7 frames in chunks of 2 (ragged last chunk): two chunks in flight on two streams must give
    exactly what one chunk at a time on one stream gives -- frames, coefficients, indices, marks, sims."""
    h, w = shape
    n, k = 7, 150
    rgb = G.synth(6, 2, n, w, h)
    marks = np.random.default_rng(4).standard_normal((n, k)).astype(np.float32)
    cfg = G.default_config(precision)
    a = _run_batch(rgb, marks, cfg, True, True, 2)
    b = _run_batch(rgb, marks, cfg, False, True, 2)
    c = _run_batch(rgb, marks, cfg, False, False, 7)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert np.all(a[4] > 0.9 * np.linalg.norm(marks, axis=1))
    rgb8 = O.f32_to_u8(rgb)
    a8 = _run_batch(rgb8, marks, cfg, True, True, 3, u8=True)
    b8 = _run_batch(rgb8, marks, cfg, False, False, 7, u8=True)
    assert np.array_equal(a8[0], b8[0]) and np.array_equal(a8[3], b8[3]) and np.array_equal(a8[4], b8[4])


# ---- pruned derived transform -----------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", [((144, 1040), 5, 200), ((80, 1056), 6, 150), ((160, 1056), 6, 150), ((1080, 1920), 5, 1000)])
def test_pruned_derived_transform_is_bit_identical_to_full(precision, case):
    """Reader::extract reads k coefficients of the derived plane (algorithm.rs:556-561): transforming only
    the frequency columns the index lists use must give the same bits as the full transform -- with two
    folding levels on the rows (1040 / 1920 columns) and three (1056 at level 6), and against the handles,
    whose Reader::derived always transforms fully."""
    (h, w), level, k = case
    n = 5 if w < 1500 else 3
    rgb = G.synth(8, 4, n, w, h)
    marks = np.random.default_rng(9).standard_normal((n, k)).astype(np.float32)
    cfg = G.default_config(precision)
    ctx = G.ctx()
    ctx.set_dct_folding(level)
    wm.default_context().set_dct_folding(level)
    try:
        ctx.reset_timing()
        pruned = _run_batch(rgb, marks, cfg, True, True, 2)
        stats = ctx.prune_stats()
        full = _run_batch(rgb, marks, cfg, True, False, 2)
        assert stats["pruned_chunks"] == (n + 1) // 2 and stats["redone_chunks"] == 0
        assert 0 < stats["columns_needed"] < stats["pruned_chunks"] * w // 4
        assert np.array_equal(pruned[3], full[3]) and np.array_equal(pruned[4], full[4])
        rd = wm.Reader.base(rgb[0], wm.ReadConfig(precision=precision))
        assert np.array_equal(pruned[3][0], rd.extract(wm.Reader.derived(pruned[0][0], precision=precision), k))
    finally:
        ctx.set_dct_folding(True)
        wm.default_context().set_dct_folding(True)


@pytest.mark.parametrize("shape", [(272, 1024), (144, 1040), (1080, 1920)])
def test_batch_paths_with_the_odd_split_off_agree_with_the_default(shape):
    """ssw_ctx_set_odd_split(0) -- every GEMM operand an exact folded sum, the round-2 arithmetic -- through the batch
    entry points: same index lists, extraction and similarity equal to the default (split, deep, class-major where the
    shape allows: 1024 x 272 takes all of it) within the rounding of single coefficients, and the pruned derived
    transform stays bit-identical to the full one in that mode too."""
    h, w = shape
    n, k = 3, 300
    rgb = G.synth(13, 1, n, w, h)
    marks = np.random.default_rng(13).standard_normal((n, k)).astype(np.float32)
    cfg = G.default_config(F64)
    ctx = G.ctx()
    default = _run_batch(rgb, marks, cfg, True, True, 2)
    ctx.set_odd_split(False)
    try:
        exact = _run_batch(rgb, marks, cfg, True, True, 2)
        exact_full = _run_batch(rgb, marks, cfg, True, False, 2)
    finally:
        ctx.set_odd_split(True)
    assert np.array_equal(exact[3], exact_full[3]) and np.array_equal(exact[4], exact_full[4])
    assert np.array_equal(default[2], exact[2])                                   # index lists
    assert np.mean(default[1] == exact[1]) >= 0.9995                              # coefficient planes
    assert np.abs(default[0] - exact[0]).max() <= 2.4e-7                          # marked frames: single ulps
    assert np.abs(default[3] - exact[3]).max() <= 1e-5 * max(1.0, float(np.abs(exact[3]).max()))
    assert np.abs(default[4] - exact[4]).max() <= 1e-4 * float(np.abs(exact[4]).max())


def test_pruned_path_falls_back_when_the_columns_do_not_fit():
    """White-noise frames spread their largest coefficients over all frequency columns: the compact plane
    overflows, the chunk is redone with the full transform, and the result still equals the full path."""
    h, w, n, k = 144, 1040, 4, 200
    rng = np.random.default_rng(12)
    rgb = rng.random((n, h, w, 3)).astype(np.float32)
    rgb[2:] = G.synth(8, 0, 2, w, h)                                  # second chunk: natural spectrum, fits
    marks = rng.standard_normal((n, k)).astype(np.float32)
    cfg = G.default_config(F64)
    ctx = G.ctx()
    ctx.reset_timing()
    pruned = _run_batch(rgb, marks, cfg, True, True, 2)
    stats = ctx.prune_stats()
    full = _run_batch(rgb, marks, cfg, False, False, 2)
    assert stats["pruned_chunks"] == 2 and stats["redone_chunks"] == 1
    assert np.array_equal(pruned[3], full[3]) and np.array_equal(pruned[4], full[4])


def test_two_contexts_in_one_process_shard_like_two_ranks():
    """One context per shard, driven by two host threads at the same time (a context is used by one thread at a time;
    different contexts are independent): the contiguous split of bench.shard_frames must reproduce the one-context run
    frame for frame -- the single-process form of the multi-GPU path."""
    import threading
    import bench
    n, w, h, k = 6, 512, 288, 120
    rgb = G.synth(14, 0, n, w, h)
    marks = np.random.default_rng(15).standard_normal((n, k)).astype(np.float32)
    cfg = G.default_config()
    whole = _run_batch(rgb, marks, cfg, True, True, 2)
    out = [None, None]

    def shard(r):
        lo, hi = bench.shard_frames(n, 2, r)
        ctx = wm.Context(0)
        try:
            ctx.set_chunk_frames(2)
            lib = ctx._lib
            d, dm = ctx.to_device(rgb[lo:hi]), ctx.to_device(marks[lo:hi])
            m = hi - lo
            o, e, s_ = ctx.alloc(rgb[lo:hi].nbytes), ctx.alloc(m * k * 4), ctx.alloc(m * 4)
            check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg), d.ptr, m, w, h, dm.ptr, k, o.ptr, None, None), "embed")
            check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), d.ptr, o.ptr, m, w, h, k, e.ptr, dm.ptr, s_.ptr), "extract")
            out[r] = (o.to_host(np.float32, rgb[lo:hi].shape), e.to_host(np.float32, (m, k)), s_.to_host(np.float32, (m,)))
        finally:
            ctx.close()

    threads = [threading.Thread(target=shard, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert out[0] is not None and out[1] is not None
    assert np.array_equal(np.concatenate([out[0][0], out[1][0]]), whole[0])
    assert np.array_equal(np.concatenate([out[0][1], out[1][1]]), whole[3])
    assert np.array_equal(np.concatenate([out[0][2], out[1][2]]), whole[4])


# ---- bench.py: ranks and launcher -----------------------------------------------------------------------------
SMALL = ["--steps", "1", "--warmup", "1", "--width", "512", "--height", "288", "--k", "100", "--no-alt",
         "--no-cpu-baseline", "--no-serial-leg", "--no-timers-off-leg"]


def _bench(extra, env_extra=None, expect_ok=True):
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(v, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, env=env,
                       timeout=900)
    if expect_ok:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout                                # exactly ONE JSON line, from rank 0
        return json.loads(lines[0])
    return r


def test_bench_two_ranks_equal_one_process(tmp_path):
    """`bench.py --gpus 2` run plainly spawns two rank processes (env-based rank set-up, one context per
    rank, barrier, MAX-reduce, gathers).  Both ranks share device 0 here (the test box has one GPU), so
    the coordination backend is gloo -- RCCL refuses two ranks on one device; the data path is the same.
    The gathered per-frame results must equal a single-process run of the same 6 frames."""
    two, one = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    j2 = _bench(["--gpus", "2", "--batch", "3", "--dump", two] + SMALL,
                {"SSW_BENCH_SHARE_DEVICE": "1", "SSW_BENCH_DIST_BACKEND": "gloo"})
    j1 = _bench(["--gpus", "1", "--batch", "6", "--dump", one] + SMALL)
    assert j2["n_gpus"] == 2 and j2["ranks"]["ranks_seen"] == [0, 1] and len(j2["ranks"]["mpix_per_s_per_rank"]) == 2
    assert j1["n_gpus"] == 1 and j1["ranks"]["ranks_seen"] == [0]
    a, b = np.load(two), np.load(one)
    assert a["sims"].shape == (6,) and np.array_equal(a["sims"], b["sims"]) and np.array_equal(a["extracted"], b["extracted"])
    assert j2["value"] > 0 and j2["scaling"] == "weak"


def test_bench_strong_scaling_splits_a_fixed_batch(tmp_path):
    """`--scaling strong --total-batch 7 --gpus 2`: the contiguous block split of SURVEY 8(e) (frame b -> rank
    floor(b * N / B): 3 + 4 frames), whole-job pixels in `value`, per-frame results equal to one process on 7 frames."""
    two, one = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    j2 = _bench(["--gpus", "2", "--scaling", "strong", "--total-batch", "7", "--dump", two, "--no-handle-leg",
                 "--no-full-transform-leg"] + SMALL, {"SSW_BENCH_SHARE_DEVICE": "1", "SSW_BENCH_DIST_BACKEND": "gloo"})
    j1 = _bench(["--gpus", "1", "--batch", "7", "--dump", one, "--no-handle-leg", "--no-full-transform-leg"] + SMALL)
    assert j2["scaling"] == "strong" and j2["config"]["total_frames"] == 7 and j2["config"]["frames_per_gpu"] == 3
    assert j1["config"]["total_frames"] == 7
    a, b = np.load(two), np.load(one)
    assert a["sims"].shape == (7,) and np.array_equal(a["sims"], b["sims"]) and np.array_equal(a["extracted"], b["extracted"])


def test_bench_line_carries_every_leg():
    """The default line's extra legs on a small shape: handle API (PCIe-inclusive, bit-identical to the batch entry
    points), the reference's four full transforms (prune off, bit-identical to the headline), CPU baseline and parity."""
    j = _bench(["--gpus", "1", "--batch", "4", "--steps", "1", "--warmup", "1", "--width", "512", "--height", "288", "--k", "100",
                "--no-alt", "--no-serial-leg", "--no-timers-off-leg"])
    h = j["handle_api"]
    assert h["bit_identical_to_batch_rgb8"] is True
    assert h["rgb8_pinned_stream"]["frame0_bit_identical_to_handles"] is True and h["rgb8_pinned_stream"]["staged_fraction"] == 0.0
    assert h["rgb8_pageable_stream"]["embed_extract_mpix_s"] > 0
    for leg in ("rgb8_pageable", "rgb8_pinned", "f32_pageable"):
        assert h[leg]["embed_extract_mpix_s"] > 0 and h[leg]["pcie_bytes_per_frame"] > 0
    assert h["rgb8_pinned"]["staged_fraction"] == 0.0 and h["rgb8_pageable"]["pcie_bytes_per_frame"] * 3 < h["f32_pageable"]["pcie_bytes_per_frame"]
    f = j["full_transform"]
    assert f["bit_identical_to_headline"] is True and f["pruned_derived_transform"]["pruned_chunks"] == 0
    assert f["executed_fraction_of_dense"] > j["kernels"]["dct_all"]["executed_fraction_of_dense"]
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["value"] > 0
    assert j["parity"]["frames"][0]["sim_delta_vs_cpu_exact"] < 1e-4
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] <= 1.0
    # r5: what bounds the step -- algorithmic bytes of every stage (GEMM launches included) and executed flop over the wall time
    rs = j["roofline_step"]
    assert rs["bound"] in ("hbm", "mfma") and 0 < rs["frac"] <= 1.0 and rs["frac"] <= rs["frac_if_not_overlapped"]
    by = rs["bytes_per_step_by_stage"]
    assert by["dct_row"] > 0 and by["dct_col"] > 0 and by["rgb_to_yiq"] > 0 and abs(sum(by.values()) - rs["algorithmic_bytes_per_step"]) < 16
    px = 4 * 512 * 288                                           # the step's pixels: RGB pre-passes move 28 + 20 + 20 B/px (writer, base, derived)
    assert abs(by["rgb_to_yiq"] - 68 * px) <= 0.01 * 68 * px
    assert set(j["config"]["transform_plan"]) >= {"pair_f64", "fused_cols"}
    # the two blocks describe the step: the GEMM kernel family and the pre-pass family (verdict r3 #2)
    assert j["roofline"]["name_match"].startswith("ssw::pair_gemm_") and 0 < j["roofline"]["share_of_step"] <= 1.0
    assert j["roofline_hbm"]["bound"] == "hbm" and j["roofline_hbm"]["name_match"] == "prep16" and 0 < j["roofline_hbm"]["frac"] <= 1.0
    assert 0 < j["roofline"]["best_launch"]["frac"] <= 1.0


def test_bench_attack_line_checks_itself_against_the_oracle():
    """`--config 4` flow (embed -> into_rgb8 -> CatmullRom 1/8 down + up -> extract, tests/attack_resize.rs:17-66) on a
    small shape: the line carries a CPU baseline of the same flow and a parity block against the oracle."""
    j = _bench(["--config", "4", "--gpus", "1", "--batch", "3", "--steps", "1", "--warmup", "1", "--width", "1024", "--height", "576",
                "--k", "400"])
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    p = j["parity"]
    assert p["resize_down_bit_exact"] is True and p["resize_up_bit_exact"] is True
    assert p["marked_rgb8_identical_fraction_vs_cpu_exact"] >= 0.9999
    assert p["extracted_max_abs_diff_vs_cpu_exact"] <= 1e-5 * 10 and p["sim_delta_vs_cpu_exact"] < 1e-4 * max(1.0, abs(p["sim_cpu_exact"]))
    assert j["roofline"]["best_launch"]["flop_per_launch"] > 0 and j["roofline"]["executed_flop_per_step"] > 0
    assert j["roofline_hbm"]["bound"] == "hbm" and 0 < j["roofline_hbm"]["frac"] <= 1.0
    assert j["config"]["chunk_frames"] == 3


def test_bench_under_torchrun_two_ranks(tmp_path):
    """The driver's multi-GPU command line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- bench.py is then one rank (RANK / LOCAL_RANK / WORLD_SIZE come
    from torchrun).  Two ranks on the test box's one GPU (gloo coordination, see above); ONE JSON line, from rank 0."""
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, SSW_BENCH_SHARE_DEVICE="1", SSW_BENCH_DIST_BACKEND="gloo")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(v, None)
    dumpf = str(tmp_path / "tr.npz")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "3", "--dump", dumpf] + SMALL,
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks"]["ranks_seen"] == [0, 1] and j["ranks"]["dist_backend"] == "gloo"
    assert np.load(dumpf)["sims"].shape == (6,)


def test_custom_closure_hybrid_through_the_low_level_entry_points():
    """`OrderingMethod::Custom` / `Insertion::Custom` closures (src/algorithm.rs:24-64) run on the host, the
    transforms around them on the device through ssw_rgb_to_yiq / ssw_dct2d / ssw_yiq_to_rgb.  With closures that restate the
    built-in Energy ordering (:214-221) and Option2 insertion (:414-432) the hybrid must reproduce Writer::mark bit for bit."""
    w, h, k = 1280, 720, 500
    rgb = G.synth(17, 0, 1, w, h)[0]
    mark = np.random.default_rng(5).standard_normal(k).astype(np.float32)
    y, i, q = G.rgb_to_yiq(rgb)
    coef = G.dct2d(y[0], L.DCT2, F64)

    def ordering(c):                         # the caller's closure: stable descending sort of c * c, DC skipped (:200-221)
        e = (c.reshape(-1) * c.reshape(-1))[1:]
        return 1 + np.argsort(-e.astype(np.float64), kind="stable")

    def insert(c, m):                        # Option2 with alpha 0.1, f32 arithmetic, un-fused (:420-424)
        return c * (np.float32(1.0) + np.float32(0.1) * m)

    order = ordering(coef)[:k]
    flat = coef.reshape(-1).copy()
    flat[order] = insert(flat[order], mark)
    back = G.dct2d(flat.reshape(h, w), L.DCT3, F64)
    marked = G.yiq_to_rgb(back, i[0], q[0])[0]
    ref = wm.Writer(rgb).mark([mark])
    assert np.array_equal(order.astype(np.uint32), wm.Reader.base(rgb).indices(k).astype(np.uint32))
    assert np.array_equal(marked, ref)


def test_bench_eight_ranks_dry_run_on_one_gpu(tmp_path):
    """SURVEY 8(e) / BASELINE configs[3]: the 8-rank launch of the driver's scaling sweep, rehearsed on this box's single
    GPU (every rank on device 0, gloo coordination; NO throughput is claimed from it): eight processes, eight contexts,
    eight streaming rings for the handle leg, the contiguous block split, rank 0's host legs (CPU baseline, parity,
    handles) while seven ranks wait in the final barrier, one JSON line -- so that the first real 8-GPU run can only
    fail for hardware reasons.  The gathered per-frame results equal a one-process run of the same 16 frames."""
    eight, one = str(tmp_path / "eight.npz"), str(tmp_path / "one.npz")
    small = ["--steps", "1", "--warmup", "1", "--width", "512", "--height", "256", "--k", "100", "--no-alt", "--no-serial-leg",
             "--no-timers-off-leg"]
    j8 = _bench(["--gpus", "8", "--batch", "2", "--dump", eight] + small, {"SSW_BENCH_SHARE_DEVICE": "1", "SSW_BENCH_DIST_BACKEND": "gloo"})
    assert j8["n_gpus"] == 8 and j8["ranks"]["ranks_seen"] == list(range(8)) and len(j8["ranks"]["mpix_per_s_per_rank"]) == 8
    assert j8["scaling"] == "weak" and j8["config"]["total_frames"] == 16 and j8["config"]["frames_per_gpu"] == 2
    assert j8["cpu_baseline"]["value"] > 0 and j8["parity"]["frames"][0]["sim_delta_vs_cpu_exact"] < 1e-4
    assert j8["handle_api"]["bit_identical_to_batch_rgb8"] is True
    j1 = _bench(["--gpus", "1", "--batch", "16", "--dump", one, "--no-cpu-baseline", "--no-handle-leg", "--no-full-transform-leg"] + small)
    a, b = np.load(eight), np.load(one)
    assert a["sims"].shape == (16,) and np.array_equal(a["sims"], b["sims"]) and np.array_equal(a["extracted"], b["extracted"])


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    r = _bench(["--gpus", str(n + 1), "--batch", "2"] + SMALL, expect_ok=False)
    assert r.returncode != 0 and f"has {n} GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no result line from a refused run
    # torchrun-style environment that does not match --gpus is refused as well
    r = _bench(["--gpus", "2", "--batch", "2"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, expect_ok=False)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_rccl_path_world_size_one():
    """The nccl (= RCCL) process group of the real multi-GPU run -- init with device_id, barrier, all_reduce,
    all_gather -- exercised with one rank on the one GPU of the test box."""
    j = _bench(["--gpus", "1", "--batch", "4"] + SMALL,
               {"SSW_FORCE_DIST": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": "29617"})
    assert j["ranks"]["dist_backend"] == "nccl" and j["ranks"]["ranks_seen"] == [0]


def test_graft_entry_smoke_runs():
    """__graft_entry__.smoke() -- what the driver runs on the GPU box before the bench -- inside the suite, so that a change of
    bars or of the default library cannot break it unnoticed (r5: it had, for the f32 leg)."""
    import importlib
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    g.smoke()
