#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of embed + extract + similarity on a batch of 4K frames.

One "step" = one pass of the whole hot path over the rank's batch of synthetic frames:
    ssw_batch_embed   (Writer::new + mark:            rgb->yiq, DCT2, top-k, embed, DCT3, yiq->rgb)
    ssw_batch_extract (Reader::base + derived + extract + Tester::similarity: 2x(rgb->y, DCT2), top-k, ...)
Inputs (frames, marks) are resident in HBM before the timed region starts.  Work shards by frame
across ranks (one process per GPU, no data-path collective: frames are independent) -> weak scaling.

Contract: python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F64_MFMA_TFLOPS = 78.6      # MI355X spec sheet: FP64 matrix
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured copy)


def shard_frames(total_frames: int, world: int, rank: int):
    """Contiguous block split of SURVEY 8(e): frame b -> rank floor(b * world / total)."""
    lo = (rank * total_frames) // world
    hi = ((rank + 1) * total_frames) // world
    return lo, hi


def run_attack_resize(args, lib, L, ctx, check, dist, dev, rank, world, rgb, rgb_out, marks, marks_host,
                      extracted, sims, B, W, H, K):
    """SURVEY 8(f) rank 1 / configs[4]: Writer::mark -> into_rgb8 -> resize to 1/8 (CatmullRom) and back
    -> Reader::extract + similarity, all on 8-bit device-resident frames (tests/attack_resize.rs)."""
    cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64 if args.precision == "f64" else L.PRECISION_F32)
    n_val = B * H * W * 3
    frames8 = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    marked8 = torch.empty_like(frames8)
    small8 = torch.empty((B, H // 8, W // 8, 3), dtype=torch.uint8, device=dev)
    back8 = torch.empty_like(frames8)
    torch.cuda.synchronize()
    check(lib.ssw_convert_f32_to_rgb8(ctx.handle, rgb.data_ptr(), n_val, frames8.data_ptr()), "to_rgb8")
    ctx.synchronize()

    def step():
        check(lib.ssw_batch_embed_rgb8(ctx.handle, C.byref(cfg), frames8.data_ptr(), B, W, H, marks.data_ptr(), K,
                                       marked8.data_ptr()), "ssw_batch_embed_rgb8")
        check(lib.ssw_resize_rgb8(ctx.handle, marked8.data_ptr(), B, W, H, W // 8, H // 8, small8.data_ptr()), "resize down")
        check(lib.ssw_resize_rgb8(ctx.handle, small8.data_ptr(), B, W // 8, H // 8, W, H, back8.data_ptr()), "resize up")
        check(lib.ssw_batch_extract_rgb8(ctx.handle, C.byref(cfg), frames8.data_ptr(), back8.data_ptr(), B, W, H, K,
                                         extracted.data_ptr(), marks.data_ptr(), sims.data_ptr()), "ssw_batch_extract_rgb8")

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.enable_timing(True)
    ctx.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    stage = ctx.timing()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sims_host = sims.cpu().numpy()
    if rank == 0:
        px = float(B) * W * H * args.steps
        rz_ms = stage["resize"]["ms"]
        # algorithmic bytes of the two resizes: 3 B/px read + 3/64 written (down), 3/64 read + 3 written (up)
        rz_gbs = px * (6.0 + 6.0 / 64.0) / (rz_ms * 1e-3) / 1e9 if rz_ms > 0 else 0.0
        print(json.dumps({
            "metric": "Mpixels/sec embed + resize attack (12.5 %) + extract", "value": round(world * px / 1e6 / elapsed, 2),
            "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU {W}x{H} 8-bit frames, {K}-coeff mark, embed -> into_rgb8 -> CatmullRom "
                                   f"resize to 1/8 and back -> extract + similarity (configs[4] flow)",
                       "frames_per_gpu": B, "width": W, "height": H, "k": K},
            "stage_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in stage.items()},
            "resize": {"gbs_algorithmic": round(rz_gbs, 1), "frac_hbm": round(rz_gbs / PEAK_HBM_GBS, 4)},
            "sim_mean": round(float(sims_host.mean()), 4), "sim_min": round(float(sims_host.min()), 4),
            "sim_sigma_threshold_6_passed": bool((sims_host > 6.0).all()),
        }))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU (configs[3]: 2048 frames / 8 GPUs)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--chunk", type=int, default=0,
                    help="frames per internal pass (0 = the library's automatic choice, ~2^28 pixels; bounds the workspace)")
    ap.add_argument("--precision", choices=["f32", "f64"], default="f64",
                    help="headline precision: f64 = canonical (bit-parity with the CPU path), f32 = fast")
    ap.add_argument("--no-alt", action="store_true", help="skip the second measurement in the other precision")
    ap.add_argument("--attack-resize", action="store_true",
                    help="configs[4] flow on 8-bit frames: embed -> into_rgb8 -> CatmullRom 1/8 down + up -> extract")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fold", action="store_true", help="dense basis GEMMs instead of the even/odd-folded ones")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SSW_FORCE_DIST"):      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import spread_spectrum_watermarking_amd as wm
    from spread_spectrum_watermarking_amd import _lib as L
    from spread_spectrum_watermarking_amd.api import check

    lib = L.load()
    ctx = wm.Context(local_rank)
    ctx.set_chunk_frames(args.chunk)
    fold_level = 0 if args.no_fold else int(os.environ.get("SSW_FOLD_LEVEL", str(L.DCT_FOLDING_DEFAULT)))
    ctx.set_dct_folding(fold_level)
    W, H, K, B = args.width, args.height, args.k, args.batch
    chunk_eff = min(args.chunk if args.chunk > 0 else max(1, (1 << 28) // (W * H)), B)   # mirrors effective_chunk() in the library

    # ---- inputs resident in HBM ------------------------------------------------------------------
    first_frame = rank * B                       # global frame index of this rank's shard (weak scaling)
    rgb = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
    rgb_out = torch.empty_like(rgb)
    gen = torch.Generator().manual_seed(args.seed * 1000003 + rank)
    marks_host = torch.randn((B, K), generator=gen, dtype=torch.float32)
    marks = marks_host.to(dev)
    extracted = torch.zeros((B, K), dtype=torch.float32, device=dev)
    sims = torch.zeros((B,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    check(lib.ssw_synth_frames(ctx.handle, args.seed, first_frame, B, W, H, rgb.data_ptr()), "ssw_synth_frames")
    ctx.synchronize()

    if args.attack_resize:
        run_attack_resize(args, lib, L, ctx, check, dist, dev, rank, world, rgb, rgb_out, marks, marks_host,
                          extracted, sims, B, W, H, K)
        if dist is not None:
            dist.destroy_process_group()
        ctx.close()
        return

    def measure(prec_name):
        """W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; returns
        (seconds [max over ranks], per-stage hipEvent timings, sims of the last step)."""
        cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64 if prec_name == "f64" else L.PRECISION_F32)

        def step():
            check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg), rgb.data_ptr(), B, W, H, marks.data_ptr(), K,
                                      rgb_out.data_ptr(), None, None), "ssw_batch_embed")
            check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), rgb.data_ptr(), rgb_out.data_ptr(), B, W, H, K,
                                        extracted.data_ptr(), marks.data_ptr(), sims.data_ptr()), "ssw_batch_extract")

        def barrier():
            ctx.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        for _ in range(args.warmup):
            step()
        ctx.enable_timing(True)                  # hipEvent pairs around every kernel on the ctx stream
        ctx.reset_timing()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        stage = ctx.timing()
        ctx.enable_timing(False)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        sims_host = sims.cpu().numpy()
        norms = marks_host.norm(dim=1).numpy()
        if not np.all(sims_host > 0.9 * norms):
            raise SystemExit(f"rank {rank}: similarity check failed ({prec_name}): "
                             f"min sim/|mark| = {(sims_host / norms).min():.3f}")
        return elapsed, stage, sims_host

    steps = args.steps
    px_total = float(B) * W * H * steps
    transforms_per_step = 4                       # DCT2, DCT3 (embed), DCT2, DCT2 (extract)
    fold_rows = (not args.no_fold) and W % 8 == 0 and W >= 16
    fold_cols = (not args.no_fold) and H % 8 == 0 and H >= 16 and W % 4 == 0

    def kernel_report(prec_name, stage):
        """Per-kernel achieved rates from the live event timers.  GEMMs: EXECUTED flop / time is the
        utilisation (folding executes 1/2 or 3/8 of the dense 2*lines*N*N); the dense figure / time
        is reported separately as "effective"."""
        peak = PEAK_F64_MFMA_TFLOPS if prec_name == "f64" else PEAK_F32_MFMA_TFLOPS
        passes = transforms_per_step * steps
        row_dense = 2.0 * B * H * W * W * passes
        col_dense = 2.0 * B * W * H * H * passes
        # which strategy each pass runs (mirrors dct2d_planes in csrc/ssw_lib.hip)
        operand = fold_level >= 3 and fold_rows and fold_cols
        two_rows = operand and fold_level >= 4 and W % 16 == 0 and W >= 64
        two_cols = operand and fold_level >= 4 and H % 8 == 0 and H >= 64       # the transposing pre-pass needs H/4 even only
        three_rows = two_rows and W % 32 == 0 and W >= 128 and (fold_level >= 6 or (fold_level == 5 and W >= 3072))
        row_frac = 0.375 if two_rows else (0.5 if fold_rows else 1.0)
        if three_rows:                               # the three forward transforms of a step: 11/32; the inverse: 3/8
            row_frac = (3 * 11.0 / 32.0 + 0.375) / 4.0
        col_frac = 0.375 if two_cols else (0.5 if fold_cols else 1.0)
        row_flops, col_flops = row_dense * row_frac, col_dense * col_frac
        esz = 8.0 if prec_name == "f64" else 4.0
        fused_rgb = bool(two_rows and W >= H)        # the three forward transforms of a step start from RGB
        rgb_bytes = (3 * 12.0 + 8.0 + 3 * esz) if fused_rgb else 56.0
        row_ms, row_n = stage["dct_row"]["ms"], max(stage["dct_row"]["launches"], 1)
        col_ms, col_n = stage["dct_col"]["ms"], max(stage["dct_col"]["launches"], 1)
        row_tf = row_flops / (row_ms * 1e-3) / 1e12 if row_ms > 0 else 0.0
        col_tf = col_flops / (col_ms * 1e-3) / 1e12 if col_ms > 0 else 0.0

        def gbs(bytes_per_px, passes_, ms):
            return (bytes_per_px * px_total * passes_) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        kernels = {
            "dct_rows": {"tflops": round(row_tf, 2), "frac_mfma": round(row_tf / peak, 4),
                         "pass_ms": round(row_ms / passes, 4),
                         "timed_passes": row_n, "executed_fraction_of_dense": row_frac,
                         "effective_dense_tflops": round(row_dense / (row_ms * 1e-3) / 1e12, 2) if row_ms > 0 else 0.0},
            "dct_cols": {"tflops": round(col_tf, 2), "frac_mfma": round(col_tf / peak, 4),
                         "pass_ms": round(col_ms / passes, 4),
                         "timed_passes": col_n, "executed_fraction_of_dense": col_frac,
                         "effective_dense_tflops": round(col_dense / (col_ms * 1e-3) / 1e12, 2) if col_ms > 0 else 0.0},
            # algorithmic bytes: writer rgb->yiq 24 B/px + two reader rgb->y 16 B/px = 56 B/px;
            # fused with the first operand pre-pass (below) the stage reads 12 B/px three times and writes
            # I, Q once and the operand planes (esz B/px) three times
            "rgb_to_yiq": {"gbs": round(gbs(rgb_bytes, 1, stage["rgb_to_yiq"]["ms"]), 1),
                           "frac_hbm": round(gbs(rgb_bytes, 1, stage["rgb_to_yiq"]["ms"]) / PEAK_HBM_GBS, 4),
                           "fused_with_operand_prepass": fused_rgb},
            "yiq_to_rgb": {"gbs": round(gbs(24.0, 1, stage["yiq_to_rgb"]["ms"]), 1),
                           "frac_hbm": round(gbs(24.0, 1, stage["yiq_to_rgb"]["ms"]) / PEAK_HBM_GBS, 4)},
            # top-k: 4 B/px algorithmic, two selections per step (writer + base reader)
            "select": {"gbs": round(gbs(4.0, 2, stage["select"]["ms"]), 1),
                       "frac_hbm": round(gbs(4.0, 2, stage["select"]["ms"]) / PEAK_HBM_GBS, 4)},
        }
        if operand:
            # operand pre-passes: f32 plane in (4 B/px), operand planes out (8 B/px in f64, 4 in f32), once per pass
            per_pass = 4.0 + esz
            prep_passes = transforms_per_step * 2 - (3 if fused_rgb else 0)
            prep_gbs = gbs(per_pass, prep_passes, stage["dct_prep"]["ms"])
            kernels["dct_prep"] = {"gbs": round(prep_gbs, 1), "frac_hbm": round(prep_gbs / PEAK_HBM_GBS, 4),
                                   "ms_per_step": round(stage["dct_prep"]["ms"] / steps, 3)}
        lines_per_launch = chunk_eff * H
        if operand:
            # dominant launch: the row GEMM over the odd frequencies (all W/2 of them, K = W/2)
            main_ms, main_n = stage["dct_row_main"]["ms"], max(stage["dct_row_main"]["launches"], 1)
            main_flop = 2.0 * lines_per_launch * (W / 2.0) * (W / 2.0)
            main_tf = main_flop * main_n / (main_ms * 1e-3) / 1e12 if main_ms > 0 else 0.0
            kernels["dct_rows"]["main_launch"] = {"avg_ms": round(main_ms / main_n, 4), "launches": main_n,
                                                  "flop_per_launch": main_flop, "tflops": round(main_tf, 2)}
            cm_ms, cm_n = stage["dct_col_main"]["ms"], max(stage["dct_col_main"]["launches"], 1)
            cm_flop = 2.0 * chunk_eff * W * (H / 2.0) * (H / 2.0)
            kernels["dct_cols"]["main_launch"] = {"avg_ms": round(cm_ms / cm_n, 4), "launches": cm_n,
                                                  "flop_per_launch": cm_flop,
                                                  "tflops": round(cm_flop * cm_n / (cm_ms * 1e-3) / 1e12, 2) if cm_ms > 0 else 0.0}
            roofline = {"bound": "mfma",
                        "kernel": ("pair_gemm_%s_kernel<rows, odd half>" if two_rows else "pair_gemm_%s_kernel<rows>") % prec_name,
                        "achieved": round(main_tf, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(main_tf / peak, 4), "traffic": None,
                        "instance": "ssw::pair_gemm_%s_kernel<false, %d, true, 0>" % (prec_name, 0),
                        "note": ("executed flop of one launch (2 * lines * (W/2) outputs * (W/2) sums: the odd-frequency "
                                 "half of the even/odd-folded basis GEMM) / its average duration (forward and inverse "
                                 "launches; `instance` is the forward one's name in the rocprofv3 kernel stats); "
                                 "whole-pass rates incl. the even-half launches in kernels.dct_rows")}
        else:
            roofline = {"bound": "mfma",
                        "kernel": ("dct_rows_folded_%s_kernel" if fold_rows else "dct_rows_%s_kernel") % prec_name,
                        "achieved": round(row_tf, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(row_tf / peak, 4), "traffic": None,
                        "note": ("executed flop per launch (even/odd-folded basis: half the dense 2*rows*W*W) / average "
                                 "launch time; dense-effective rate in kernels.dct_rows.effective_dense_tflops")
                        if fold_rows else "dense flop per launch / average launch time"}
        # HBM-side traffic of the dominant kernel: PMC counters collected offline exactly as
        # MI355X_MICROARCH.md prescribes (separate --pmc passes, gfx950 FETCH_SIZE x2 correction) and
        # committed in profiles/r1_pmc_traffic.json; only quoted when this run matches that workload.
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            wl = pmc["workload"]
            if (wl["width"], wl["height"], wl["chunk_frames"]) == (W, H, chunk_eff) and roofline["kernel"] in pmc["kernels"]:
                roofline["traffic"] = pmc["kernels"][roofline["kernel"]]["hbm_bytes_per_launch"]
                roofline["traffic_unit"] = "bytes/launch (L2<->fabric, incl. Infinity-Cache hits)"
                if operand:   # D operand plane in (W/2 elements per line), odd half basis, f32 odd outputs
                    esz = 8 if prec_name == "f64" else 4
                    roofline["algorithmic_bytes_per_launch"] = int(lines_per_launch * (W // 2) * esz + (W // 2) ** 2 * esz +
                                                                   lines_per_launch * (W // 2) * 4)
                else:
                    roofline["algorithmic_bytes_per_launch"] = int(chunk_eff * H * W * 8 +
                                                                   2 * (W // 2) ** 2 * (8 if prec_name == "f64" else 4))
        except (OSError, KeyError, ValueError):
            pass
        return kernels, roofline, {k: round(v["ms"] / steps, 3) for k, v in stage.items()}

    elapsed, stage, sims_host = measure(args.precision)
    kernels, roofline, stage_ms = kernel_report(args.precision, stage)
    alt = None
    if not args.no_alt:
        alt_name = "f32" if args.precision == "f64" else "f64"
        alt_elapsed, alt_stage, alt_sims = measure(alt_name)
        alt_kernels, alt_roofline, alt_stage_ms = kernel_report(alt_name, alt_stage)
        alt = {"dtype": alt_name, "value": round(world * px_total / 1e6 / alt_elapsed, 2), "unit": "Mpix/s",
               "ms_per_step": round(alt_elapsed / steps * 1e3, 3), "roofline": alt_roofline,
               "kernels": {k: alt_kernels[k] for k in ("dct_rows", "dct_cols")},
               "sim_mean": round(float(alt_sims.mean()), 4),
               "max_abs_sim_diff_vs_headline": float(np.abs(alt_sims - sims_host).max())}
        # leave the headline precision's outputs in rgb_out / sims for the parity leg below
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            measure_cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1,
                                   L.PRECISION_F64 if args.precision == "f64" else L.PRECISION_F32)
            check(lib.ssw_batch_embed(ctx.handle, C.byref(measure_cfg), rgb.data_ptr(), 1, W, H, marks.data_ptr(), K,
                                      rgb_out.data_ptr(), None, None), "ssw_batch_embed")
            check(lib.ssw_batch_extract(ctx.handle, C.byref(measure_cfg), rgb.data_ptr(), rgb_out.data_ptr(), 1, W, H, K,
                                        extracted.data_ptr(), marks.data_ptr(), sims.data_ptr()), "ssw_batch_extract")
            ctx.synchronize()

    result = None
    if rank == 0:
        mpix = world * px_total / 1e6
        result = {
            "metric": "Mpixels/sec embed+extract (4K batch)",
            "value": round(mpix / elapsed, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU {W}x{H} f32 frames, {K}-coeff mark, embed+extract+similarity "
                                   f"(per-GPU shard of configs[3]: batch=2048 3840x2160 across 8 GPUs)",
                       "frames_per_gpu": B, "width": W, "height": H, "k": K, "alpha": 0.1,
                       "method": "Option2", "ordering": "Energy", "chunk_frames": chunk_eff,
                       "dct_folding_level": fold_level,
                       "parallelism": f"frame-sharded x{world}, no collectives"},
            "roofline": roofline,
            "kernels": kernels,
            "stage_ms_per_step": stage_ms,
            "sim_mean": round(float(sims_host.mean()), 4),
        }
        if alt is not None:
            result["alt_precision"] = alt

    # ---- CPU baseline: the oracle (faithful mode) on a bounded sample, rank 0 at N=1 only ----------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        frame0 = rgb[0].cpu().numpy()
        mark0 = marks_host[0].numpy()
        gpu_marked0 = rgb_out[0].cpu().numpy()
        # timed: what the reference does -- f32 FFT-class DCT + full stable sort of all W*H-1 keys
        t0 = time.perf_counter()
        cpu_marked = O.embed_frame(frame0, mark0, backend=O.BACKEND_F32, full_sort=True)
        cpu_ext, cpu_sim = O.extract_frame(frame0, cpu_marked, mark0, backend=O.BACKEND_F32, full_sort=True)
        cpu_s = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": round(W * H / 1e6 / cpu_s, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": f"1 frame {W}x{H} (frame 0 of the batch), embed+extract+similarity, oracle C restatement: "
                      f"f32 FFT DCT + full stable sort like the reference, single thread, {cpu_s:.1f} s",
        }
        # the same pipeline with a partial selection of the first k entries instead of the full sort (what
        # the GPU path computes; SURVEY 8(d): "so the comparison is not only against the sort")
        t0 = time.perf_counter()
        sel_marked = O.embed_frame(frame0, mark0, backend=O.BACKEND_F32, full_sort=False)
        O.extract_frame(frame0, sel_marked, mark0, backend=O.BACKEND_F32, full_sort=False)
        sel_s = time.perf_counter() - t0
        result["cpu_baseline_select"] = {
            "value": round(W * H / 1e6 / sel_s, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": f"same frame, f32 FFT DCT + top-k selection instead of the full sort, single thread, {sel_s:.1f} s",
        }
        # same work, one frame per thread on the host's cores (the reference itself is single-threaded;
        # this is the frame-parallel upper bound of SURVEY 8(d)).  ctypes releases the GIL in the C calls.
        import concurrent.futures as cf
        n_thr = max(1, min(os.cpu_count() or 1, 16))

        def one(_):
            m = O.embed_frame(frame0, mark0, backend=O.BACKEND_F32, full_sort=True)
            O.extract_frame(frame0, m, mark0, backend=O.BACKEND_F32, full_sort=True)
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(n_thr) as ex:
            list(ex.map(one, range(n_thr)))
        par_s = time.perf_counter() - t0
        result["cpu_baseline_parallel"] = {
            "value": round(n_thr * W * H / 1e6 / par_s, 4), "unit": "Mpix/s", "cores": n_thr, "kind": "port",
            "sample": f"{n_thr} frames {W}x{H}, one per thread ({os.cpu_count()} logical cores on the host), "
                      f"same faithful pipeline, {par_s:.1f} s",
        }
        # untimed: the oracle's correctly rounded (f64-backend) pipeline = what the canonical precision must equal
        ref_marked = O.embed_frame(frame0, mark0, backend=O.BACKEND_F64, full_sort=False)
        ref_ext, ref_sim = O.extract_frame(frame0, ref_marked, mark0, backend=O.BACKEND_F64, full_sort=False)
        gpu_ext0 = extracted[0].cpu().numpy()
        result["parity"] = {
            "precision": args.precision,
            "sim_gpu": float(sims_host[0]),
            "sim_cpu_f32fft": float(cpu_sim), "sim_delta_vs_cpu_f32fft": abs(float(sims_host[0]) - float(cpu_sim)),
            "sim_cpu_exact": float(ref_sim), "sim_delta_vs_cpu_exact": abs(float(sims_host[0]) - float(ref_sim)),
            "marked_frame_max_abs_diff_vs_cpu_exact": float(np.abs(gpu_marked0 - ref_marked).max()),
            "marked_frame_bit_identical_fraction": float(np.mean(gpu_marked0 == ref_marked)),
        }
        result["parity"]["extracted_max_abs_diff_vs_cpu_exact"] = float(np.abs(gpu_ext0 - ref_ext).max())

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
