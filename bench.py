#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of embed + extract + similarity on a batch of 4K frames.

One "step" = one pass of the whole hot path over the rank's batch of synthetic frames:
    ssw_batch_embed   (Writer::new + mark:            rgb->yiq, DCT2, top-k, embed, DCT3, yiq->rgb)
    ssw_batch_extract (Reader::base + derived + extract + Tester::similarity: 2x(rgb->y, DCT2), top-k, ...)
Inputs (frames, marks) are resident in HBM before the timed region starts.  Work shards by frame
across ranks (one process per GPU, no data-path collective: frames are independent) -> weak scaling.

Contract: python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
  * under torchrun (WORLD_SIZE / RANK / LOCAL_RANK in the environment) this process is one rank;
  * run plainly with --gpus N > 1 it is the LAUNCHER: it starts N rank processes of itself (before it
    has made any GPU call -- it never touches the GPU) and exits with their status.
--config {1,2,3,4} selects the BASELINE.json configuration of that index (default: the per-GPU shard of
configs[3], the one the metric is quoted on).
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
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

# BASELINE.json "configs", quoted verbatim, and what bench.py runs for each (per GPU)
BASELINE_CONFIGS = {
    1: {"quote": "single 3840\u00d72160 f32 frame, 1000-coeff embed + IDCT round-trip on 1 MI355X",
        "batch": 1, "width": 3840, "height": 2160, "k": 1000, "flow": "embed"},
    2: {"quote": "batch=256 1920\u00d71080 frames, embed+extract+similarity, 1 MI355X (HBM-roofline report)",
        "batch": 256, "width": 1920, "height": 1080, "k": 1000, "flow": "embed+extract"},
    3: {"quote": "batch=2048 3840\u00d72160 frames, 1000-coeff, batch-sharded across 8\u00d7MI355X (no collectives)",
        "batch": 256, "width": 3840, "height": 2160, "k": 1000, "flow": "embed+extract"},
    4: {"quote": "batch=512 7680\u00d74320 frames, 10000-coeff mark, + attack_resize 12.5% re-extract, 8\u00d7MI355X",
        "batch": 64, "width": 7680, "height": 4320, "k": 10000, "flow": "attack"},
}


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def count_gpus_sysfs(nodes: str = "/sys/class/kfd/kfd/topology/nodes", dri: str = "/dev/dri") -> int:
    """GPUs of this node as the kernel driver lists them -- no HIP / HSA call, so the launcher process never opens a device
    (a parent that has initialised the GPU must not fork ranks on this pool).  KFD topology nodes with simd_count > 0 are GPUs
    (CPUs have 0); HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow the count like they narrow the ranks' view."""
    n = 0
    for props in glob.glob(os.path.join(nodes, "*", "properties")):
        try:
            with open(props) as f:
                for line in f:
                    key, _, val = line.partition(" ")
                    if key == "simd_count" and int(val) > 0:
                        n += 1
                        break
        except (OSError, ValueError):
            pass
    render = len(glob.glob(os.path.join(dri, "renderD*")))        # what the container's device cgroup lets through
    n = render if n == 0 else (min(n, render) if render else n)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n: int) -> int:
    """Parent of a plain `python bench.py --gpus N` run: one child process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, the same command line.  The parent makes no GPU call
    (the GPUs are counted from sysfs, count_gpus_sysfs) and never re-execs."""
    shared = bool(os.environ.get("SSW_BENCH_SHARE_DEVICE"))           # tests: every rank on device 0
    n_dev = count_gpus_sysfs()
    if n_dev == 0:
        print("bench.py needs a GPU: the product path has no CPU fallback", file=sys.stderr)
        return 2
    if n > n_dev and not shared:
        print(f"bench.py: --gpus {n} but this node has {n_dev} GPU(s)", file=sys.stderr)
        return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n))
        # rank 0 owns stdout (the one JSON line); the others' chatter goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in pending:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def shard_frames(total_frames: int, world: int, rank: int):
    """Contiguous block split of SURVEY 8(e): frame b -> rank floor(b * world / total)."""
    lo = (rank * total_frames) // world
    hi = ((rank + 1) * total_frames) // world
    return lo, hi


def timed_region(args, ctx, dist, step):
    """W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize on both sides.
    Returns (own seconds, MAX over ranks, per-stage event timings of the timed steps)."""
    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.enable_timing(not args.no_stage_timers)      # hipEvent pairs around every stage, on the stream it runs on
    ctx.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    barrier()
    stage = ctx.timing()
    prune = ctx.prune_stats()
    ctx.enable_timing(False)
    return own, (dist.max(own) if dist is not None else own), stage, prune


def rate(work, ms):
    return work / (ms * 1e-3) if ms > 0 else 0.0


def hbm_report(stage, names):
    """Achieved GB/s of algorithmic bytes (counted by the library per launch) per HBM-bound stage."""
    out = {}
    for n in names:
        if stage[n]["launches"]:
            g = rate(stage[n]["work"], stage[n]["ms"]) / 1e9
            out[n] = {"gbs": round(g, 1), "frac_hbm": round(g / PEAK_HBM_GBS, 4),
                      "avg_ms": round(stage[n]["ms"] / stage[n]["launches"], 4), "launches": stage[n]["launches"]}
    return out


def step_roofline(prec_name, stage, steps, wall_ms_per_step):
    """What bounds the STEP (verdict r4 #3): algorithmic HBM bytes of every stage (the HBM-bound stages' own bytes, and for
    the GEMM launches the operand planes in, the results out and the inverse's A1 / T2 / E exchange: ssw_ctx_get_traffic)
    and the executed flop of every GEMM launch, each priced at its peak; a perfectly overlapped step would take the
    larger of the two times, one with no overlap at all their sum.  `frac` = that larger time / the measured wall time."""
    peak = PEAK_F64_MFMA_TFLOPS if prec_name == "f64" else PEAK_F32_MFMA_TFLOPS
    names = [k for k in stage if not k.endswith("_main")]
    by_stage = {k: stage[k]["bytes"] / steps for k in names if stage[k]["bytes"]}
    total_bytes = sum(by_stage.values())
    flop = (stage["dct_row"]["work"] + stage["dct_col"]["work"]) / steps
    t_hbm = total_bytes / (PEAK_HBM_GBS * 1e9) * 1e3
    t_mfma = flop / (peak * 1e12) * 1e3
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    return {"bound": bound, "algorithmic_bytes_per_step": total_bytes, "executed_flop_per_step": flop,
            "ms_at_hbm_peak": round(t_hbm, 3), "ms_at_mfma_peak": round(t_mfma, 3), "ms_per_step": round(wall_ms_per_step, 3),
            "frac": round(max(t_hbm, t_mfma) / wall_ms_per_step, 4) if wall_ms_per_step else None,
            "frac_if_not_overlapped": round((t_hbm + t_mfma) / wall_ms_per_step, 4) if wall_ms_per_step else None,
            "peak_hbm_gbs": PEAK_HBM_GBS, "peak_mfma_tflops": peak,
            "bytes_per_step_by_stage": {k: round(v) for k, v in by_stage.items()},
            "how": "max(algorithmic bytes / 8 TB/s, executed flop / MFMA peak) / measured ms per step; bytes counted per launch by "
                   "the library (ssw_ctx_get_traffic), each byte once"}


# The launch `roofline.best_launch` is about: ONE GEMM launch of a forward row pass, under its own template instance.
#   f64 (default): rows of 3072 columns or more run at level 2 (csrc/ssw_pipeline.hip build_pass): eight launches that all
#       sum W/16 terms over W/16 output pairs; the timed one is kind 7 -- class O of the split odd half rotated once more,
#       its "+" launch: (a, b) of AD plus (a, b) of the reversed BS against the cosine / sine rows of the W/2 bases
#       (csrc/dct_pair_prep.hip pair_prep16_rows_kernel); its own template instance (SUB = 4).  Shorter rows (level 1): class O
#       of the split odd half itself, W/8 pairs x W/8 terms.
#   f32: the unsplit odd half (the f32 twin keeps exact-operand folding)
FUSED_FORWARD = False      # set by main() from ssw_ctx_transform_plan: the row launches of this workload write the column operands (r5)


def main_kernel_label(prec_name):
    if prec_name != "f64":
        return "pair_gemm_f32_kernel<rows, odd half>"
    if FUSED_FORWARD:
        return "pair_gemm_f64_kernel<rows, fused column operands, class O rotated '+' launch>"
    return "pair_gemm_f64_kernel<rows, split odd half, class O (level 2: rotated, '+' launch)>"


def main_kernel_instance(prec_name):
    if prec_name != "f64":
        return "ssw::pair_gemm_f32_kernel<false, 0, true, 0>"
    return "ssw::pair_gemm_f64_kernel<false, 7, false, 4>" if FUSED_FORWARD else "ssw::pair_gemm_f64_kernel<false, 0, false, 4>"


MAIN_KERNEL_NOTE = {
    "f64": ("executed flop of one launch (two products of lines x P pair slots x P sums, P = W/16 at level 2 (rows of 3072 "
            "columns or more), W/8 below: the cosine and the sine part of class O of the split odd half -- at level 2 of its "
            "rotated '+' launch, P + 1 output pairs, the first and the last sharing a slot -- counted by the library per "
            "launch) / its average duration from a hipEvent pair on the stream it runs on, inside the timed region"),
    "f32": ("executed flop of one launch (2 * lines * (W/2) outputs * (W/2) sums: the odd-frequency half of the even/odd-"
            "folded basis GEMM, counted by the library per launch) / its average duration from a hipEvent pair on the "
            "stream it runs on, inside the timed region"),
}


def load_pmc(W, H, chunk_eff):
    """The newest committed PMC summary for this workload (profiles/r*_pmc_traffic*.json): HBM-side traffic collected
    offline exactly as MI355X_MICROARCH.md prescribes (separate --pmc passes, gfx950 FETCH_SIZE x2 correction), stamped
    with the commit it was collected at; quoted only for the workload it was collected on."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            wl = pmc["workload"]
            if (wl["width"], wl["height"], wl["chunk_frames"]) == (W, H, chunk_eff):
                pmc["_source"] = f"profiles/{os.path.basename(path)} (collected at commit {pmc.get('commit', '?')})"
                return pmc
        except (OSError, KeyError, ValueError):
            pass
    return None


def attach_pmc_traffic(roofline, prec_name, W, H, chunk_eff):
    """Per-launch traffic of one named launch (the `best_launch` sub-block)."""
    pmc = load_pmc(W, H, chunk_eff)
    if pmc and roofline["kernel"] in pmc.get("kernels", {}):
        k = pmc["kernels"][roofline["kernel"]]
        roofline["traffic"] = k["hbm_bytes_per_launch"]
        roofline["traffic_unit"] = "bytes/launch (L2<->fabric, incl. Infinity-Cache hits)"
        roofline["traffic_source"] = pmc["_source"]
        roofline["algorithmic_bytes_per_launch"] = k.get("algorithmic_bytes_per_launch")
        if k.get("mfma_busy_over_active_cycles") is not None:     # cycles, not seconds: what the kernel does with the clock it gets
            roofline["mfma_busy_over_active_cycles"] = k["mfma_busy_over_active_cycles"]


def family_rooflines(prec_name, stage, steps, W, H, chunk_eff, frames_per_step):
    """The two blocks that describe the STEP (verdict r3 #2): `roofline` = every GEMM launch of the step (the
    pair_gemm kernel family: row and column passes of the forward, inverse and pruned transforms), executed flop /
    summed duration; `roofline_hbm` = every deep pre-pass launch (rocprofv3 names containing "prep16": the RGB -> operand
    pre-passes timed as stage rgb_to_yiq and the transposing / inverse ones timed as stage dct_prep), algorithmic bytes /
    summed duration.  Flop and bytes are counted by the library per launch; durations are hipEvent pairs on the stream
    each stage runs on.  `share_of_step` = the family's share of the summed kernel time of all stages (two lanes overlap in
    wall time).  `best_launch` keeps r3's single-launch block (one launch of a forward row pass: main_kernel_label)."""
    peak = PEAK_F64_MFMA_TFLOPS if prec_name == "f64" else PEAK_F32_MFMA_TFLOPS
    all_ms = sum(v["ms"] for k, v in stage.items() if not k.endswith("_main"))
    gemm_ms = stage["dct_row"]["ms"] + stage["dct_col"]["ms"]
    executed = stage["dct_row"]["work"] + stage["dct_col"]["work"]
    tf = rate(executed, gemm_ms) / 1e12
    fam = f"pair_gemm_{prec_name}_kernel"
    main_ms, main_n = stage["dct_row_main"]["ms"], max(stage["dct_row_main"]["launches"], 1)
    main_tf = rate(stage["dct_row_main"]["work"], main_ms) / 1e12
    best = {"kernel": main_kernel_label(prec_name), "instance": main_kernel_instance(prec_name),
            "achieved": round(main_tf, 2), "frac": round(main_tf / peak, 4), "traffic": None,
            "avg_ms": round(main_ms / main_n, 4), "launches": main_n, "flop_per_launch": stage["dct_row_main"]["work"] / main_n,
            "note": MAIN_KERNEL_NOTE[prec_name]}
    attach_pmc_traffic(best, prec_name, W, H, chunk_eff)
    roofline = {"bound": "mfma", "kernel": f"{fam} (family: every GEMM launch of the step)", "name_match": f"ssw::{fam}<",
                "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
                "share_of_step": round(gemm_ms / all_ms, 4) if all_ms else None,
                "executed_flop_per_step": executed / steps, "kernel_ms_per_step": round(gemm_ms / steps, 3),
                "by_pass": {"rows": {"tflops": round(rate(stage["dct_row"]["work"], stage["dct_row"]["ms"]) / 1e12, 2),
                                     "ms_per_step": round(stage["dct_row"]["ms"] / steps, 3)},
                            "cols": {"tflops": round(rate(stage["dct_col"]["work"], stage["dct_col"]["ms"]) / 1e12, 2),
                                     "ms_per_step": round(stage["dct_col"]["ms"] / steps, 3)}},
                "how": "sum of executed flop of all launches of the family / sum of their durations; recompute from a rocprofv3 "
                       "kernel-trace summary as executed_flop_per_step x steps / (sum of TotalDurationNs over the rows whose Name "
                       "starts with name_match)",
                "best_launch": best}
    pre_ms = stage["rgb_to_yiq"]["ms"] + stage["dct_prep"]["ms"]
    pre_bytes = stage["rgb_to_yiq"]["work"] + stage["dct_prep"]["work"]
    gbs = rate(pre_bytes, pre_ms) / 1e9
    members = hbm_report(stage, ["rgb_to_yiq", "dct_prep"])
    roofline_hbm = {"bound": "hbm", "kernel": "deep operand pre-passes (family: pair_prep16_rows_light_kernel<rgb>, prep16_derived_fused_kernel (the derived "
                                               "frame's pre-pass + pruned row pass in one), prep16_cols_l2_kernel, prep16_inv_rows_l2_kernel, "
                                               "prep16_inv_cols_l2_kernel; below the level-2 sizes the prep16_*_staged_kernel forms)", "name_match": "prep16",
                    "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                    "share_of_step": round(pre_ms / all_ms, 4) if all_ms else None,
                    "algorithmic_bytes_per_step": pre_bytes / steps, "kernel_ms_per_step": round(pre_ms / steps, 3),
                    "members": members,
                    "how": "algorithmic bytes (SURVEY 8(d): what each pre-pass must read and write once, counted per launch) of all "
                           "launches of the family / sum of their durations"}
    pmc = load_pmc(W, H, chunk_eff)
    if pmc and "families" in pmc:
        scale = frames_per_step / float(chunk_eff)                 # the PMC run profiles steps of ONE pass of chunk_eff frames
        for block, key in ((roofline, "gemm"), (roofline_hbm, "prepass")):
            fm = pmc["families"].get(key)
            if fm:
                block["traffic"] = int(fm["hbm_bytes_per_step"] * scale)
                block["traffic_unit"] = "bytes/step over all launches of the family (L2<->fabric, incl. Infinity-Cache hits)"
                block["traffic_source"] = pmc["_source"]
                if "traffic_over_algorithmic" in fm:
                    block["traffic_over_algorithmic"] = fm["traffic_over_algorithmic"]
    return roofline, roofline_hbm


def run_attack_resize(args, lib, L, ctx, check, dist, dev, rank, world, rgb, rgb_out, marks, marks_host,
                      extracted, sims, B, total_frames, W, H, K, workload_tag, rank_report, dump):
    """SURVEY 8(f) rank 1 / configs[4]: Writer::mark -> into_rgb8 -> resize to 1/8 (CatmullRom) and back
    -> Reader::extract + similarity, all on 8-bit device-resident frames (tests/attack_resize.rs)."""
    cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64 if args.precision == "f64" else L.PRECISION_F32)
    n_val = B * H * W * 3
    frames8 = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    check(lib.ssw_convert_f32_to_rgb8(ctx.handle, rgb.data_ptr(), n_val, frames8.data_ptr()), "to_rgb8")
    ctx.synchronize()
    del rgb, rgb_out                                   # the 8-bit flow does not need the f32 frames
    torch.cuda.empty_cache()
    marked8 = torch.empty_like(frames8)
    small8 = torch.empty((B, H // 8, W // 8, 3), dtype=torch.uint8, device=dev)
    back8 = torch.empty_like(frames8)
    torch.cuda.synchronize()

    def step():
        check(lib.ssw_batch_embed_rgb8(ctx.handle, C.byref(cfg), frames8.data_ptr(), B, W, H, marks.data_ptr(), K,
                                       marked8.data_ptr()), "ssw_batch_embed_rgb8")
        check(lib.ssw_resize_rgb8(ctx.handle, marked8.data_ptr(), B, W, H, W // 8, H // 8, small8.data_ptr()), "resize down")
        check(lib.ssw_resize_rgb8(ctx.handle, small8.data_ptr(), B, W // 8, H // 8, W, H, back8.data_ptr()), "resize up")
        check(lib.ssw_batch_extract_rgb8(ctx.handle, C.byref(cfg), frames8.data_ptr(), back8.data_ptr(), B, W, H, K,
                                         extracted.data_ptr(), marks.data_ptr(), sims.data_ptr()), "ssw_batch_extract_rgb8")

    own, elapsed, stage, prune = timed_region(args, ctx, dist, step)
    sims_host = sims.cpu().numpy()
    ext_host = extracted.cpu().numpy()
    ranks = rank_report(own)
    dump(sims_host, ext_host)
    chunk_eff = ctx.pass_frames(B, W, H)
    # the same step with one pass at a time on one stream: what each kernel does alone (verdict r4 #8)
    serial = None
    if not args.no_serial_leg:
        keep = args.steps
        args.steps = max(1, min(args.steps, 3))
        ctx.set_overlap(False)
        _, s_elapsed, s_stage, _ = timed_region(args, ctx, dist, step)
        ctx.set_overlap(not args.no_overlap)
        if rank == 0:
            s_roofline, s_roofline_hbm = family_rooflines(args.precision, s_stage, args.steps, W, H, chunk_eff, B)
            serial = {"value": round(float(total_frames) * W * H * args.steps / 1e6 / s_elapsed, 2), "unit": "Mpix/s",
                      "ms_per_step": round(s_elapsed / args.steps * 1e3, 3), "steps": args.steps,
                      "roofline_step": step_roofline(args.precision, s_stage, args.steps, s_elapsed / args.steps * 1e3),
                      "roofline": s_roofline, "roofline_hbm": s_roofline_hbm,
                      "stage_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in s_stage.items()},
                      "hbm_kernels": hbm_report(s_stage, ["rgb_to_yiq", "dct_prep", "select", "yiq_to_rgb", "resize"]),
                      "bit_identical_to_overlapped": bool(np.array_equal(sims.cpu().numpy(), sims_host) and
                                                          np.array_equal(extracted.cpu().numpy(), ext_host))}
        args.steps = keep
    result = None
    if rank == 0:
        roofline, roofline_hbm = family_rooflines(args.precision, stage, args.steps, W, H, chunk_eff, B)
        result = {
            "metric": "Mpixels/sec embed + resize attack (12.5 %) + extract",
            "value": round(float(total_frames) * W * H * args.steps / 1e6 / elapsed, 2),
            "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU {W}x{H} 8-bit frames, {K}-coeff mark, embed -> into_rgb8 -> CatmullRom "
                                   f"resize to 1/8 and back -> extract + similarity; {workload_tag}",
                       "frames_per_gpu": B, "total_frames": total_frames, "width": W, "height": H, "k": K,
                       "chunk_frames": chunk_eff, "transform_plan": ctx.transform_plan(min(chunk_eff, B), W, H),
                       "parallelism": f"frame-sharded x{world}, no collectives"},
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "roofline_step": step_roofline(args.precision, stage, args.steps, elapsed / args.steps * 1e3),
            "stage_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in stage.items()},
            "hbm_kernels": hbm_report(stage, ["rgb_to_yiq", "dct_prep", "select", "yiq_to_rgb", "resize"]),
            "pruned_derived_transform": prune,
            "ranks": ranks,
            "sim_mean": round(float(sims_host.mean()), 4), "sim_min": round(float(sims_host.min()), 4),
            "sim_sigma_threshold_6_passed": bool((sims_host > 6.0).all()),
        }
        if serial is not None:
            result["serialized"] = serial
    if rank == 0 and not args.no_cpu_baseline:
        # tests/attack_resize.rs:17-66 on frame 0 with the oracle: timed in the reference's own arithmetic class
        # (f32 FFT DCT + full stable sort), then untimed with the exact (f64) backend as the parity checker
        from oracle import oracle as O
        frame8 = frames8[0].cpu().numpy()
        mark0 = marks_host[0].numpy()
        f32_frame = O.u8_to_f32(frame8)
        t0 = time.perf_counter()
        c_marked8 = O.f32_to_u8(O.embed_frame(f32_frame, mark0, backend=O.BACKEND_F32, full_sort=True))
        c_back = O.resize_rgb8(O.resize_rgb8(c_marked8, W // 8, H // 8), W, H)
        _, c_sim = O.extract_frame(f32_frame, O.u8_to_f32(c_back), mark0, backend=O.BACKEND_F32, full_sort=True)
        cpu_s = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": round(W * H / 1e6 / cpu_s, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": f"1 frame {W}x{H} (frame 0 of the batch): embed -> into_rgb8 -> CatmullRom 1/8 down + up -> extract + "
                      f"similarity, oracle C restatement (f32 FFT DCT + full stable sort like the reference), single "
                      f"thread, {cpu_s:.1f} s"}
        g_marked8 = marked8[0].cpu().numpy()
        g_small, g_back = small8[0].cpu().numpy(), back8[0].cpu().numpy()
        g_ext, g_sim = extracted[0].cpu().numpy(), float(sims_host[0])
        o_marked8 = O.f32_to_u8(O.embed_frame(f32_frame, mark0, backend=O.BACKEND_F64))
        o_small = O.resize_rgb8(g_marked8, W // 8, H // 8)             # the oracle's resize of the GPU's own frames
        o_back = O.resize_rgb8(g_small, W, H)
        o_ext, o_sim = O.extract_frame(f32_frame, O.u8_to_f32(g_back), mark0, backend=O.BACKEND_F64)
        result["parity"] = {
            "precision": args.precision, "frame": 0,
            "marked_rgb8_identical_fraction_vs_cpu_exact": float(np.mean(g_marked8 == o_marked8)),
            "resize_down_bit_exact": bool(np.array_equal(g_small, o_small)),
            "resize_up_bit_exact": bool(np.array_equal(g_back, o_back)),
            "extracted_max_abs_diff_vs_cpu_exact": float(np.abs(g_ext - o_ext).max()),
            "extracted_bit_identical_fraction": float(np.mean(g_ext == o_ext)),
            "sim_gpu": g_sim, "sim_cpu_exact": float(o_sim), "sim_delta_vs_cpu_exact": abs(g_sim - float(o_sim)),
            "sim_cpu_f32fft_own_flow": float(c_sim),
            "note": "extraction compared on the SAME attacked 8-bit frame (the GPU's); the marked frame against the "
                    "oracle's own embed; both resizes against the oracle's resize of the GPU's input frames"}
    if rank == 0:
        print(json.dumps(result))


def handle_api_leg(wm, L, lib, ctx, check, rgb0_dev, W, H, K, precision, reps=5):
    """The drop-in path a `wm::Writer::mark()` / `Reader::extract` caller takes (src/algorithm.rs:295-316, :355-379,
    :474-480, :529-539; examples/main.rs:271-278): ONE host image at a time through the single-image handles, host
    buffers in and out, PCIe included.  8-bit frames (what image files decode to) from pageable and from pinned host
    memory, and f32 frames (what `into_rgb32f()` yields) for comparison.  Never the headline."""
    prec = L.PRECISION_F64 if precision == "f64" else L.PRECISION_F32
    wcfg, rcfg = wm.WriteConfig(precision=prec), wm.ReadConfig(precision=prec)
    n_val = H * W * 3
    frame8_dev = torch.empty((H, W, 3), dtype=torch.uint8, device=rgb0_dev.device)
    torch.cuda.synchronize()
    check(lib.ssw_convert_f32_to_rgb8(ctx.handle, rgb0_dev.data_ptr(), n_val, frame8_dev.data_ptr()), "to_rgb8")
    ctx.synchronize()
    img8 = frame8_dev.cpu().numpy()
    img32 = img8.astype(np.float32) / np.float32(255)
    mark = np.random.default_rng(12345).standard_normal(K).astype(np.float32)
    px = W * H / 1e6

    def run(img, out_buf, u8_out):
        def embed():
            wr = wm.Writer(img, wcfg, ctx)
            return wr.mark_rgb8([mark], out=out_buf) if u8_out else wr.mark([mark], out=out_buf)

        def extract(marked):
            ext = wm.Reader.base(img, rcfg, ctx).extract(wm.Reader.derived(marked, ctx, prec), K)
            return ext, wm.Tester(ext, ctx).similarity(mark).similarity
        marked = embed()
        extract(marked)                                  # warm-up: bases, staging ring, plane pool
        ctx.transfer_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            marked = embed()
        t1 = time.perf_counter()
        for _ in range(reps):
            ext, sim = extract(marked)
        t2 = time.perf_counter()
        st = ctx.transfer_stats()
        return {"embed_ms": round((t1 - t0) / reps * 1e3, 3), "extract_ms": round((t2 - t1) / reps * 1e3, 3),
                "embed_mpix_s": round(px * reps / (t1 - t0), 1), "extract_mpix_s": round(px * reps / (t2 - t1), 1),
                "embed_extract_mpix_s": round(px * reps / (t2 - t0), 1),
                "pcie_bytes_per_frame": int((st["h2d_bytes"] + st["d2h_bytes"]) / reps),
                "staged_fraction": round(st["staged_bytes"] / max(st["h2d_bytes"] + st["d2h_bytes"], 1.0), 3),
                "sim": round(float(sim), 4)}, marked.copy(), ext.copy()

    out = {"frame": f"{W}x{H}", "k": K, "reps": reps, "dtype": precision,
           "flow": "Writer::new + mark [+ into_rgb8] | Reader::base + Reader::derived + extract + Tester::similarity, "
                   "one host image per call, host buffers in and out (PCIe included)"}
    # pageable legs: ordinary numpy arrays, the output array allocated once and reused like a caller's frame buffer
    out["rgb8_pageable"], marked8, ext8 = run(img8, np.empty_like(img8), True)
    pin_in, pin_out = ctx.pinned_empty(img8.shape, np.uint8), ctx.pinned_empty(img8.shape, np.uint8)
    pin_in[...] = img8
    out["rgb8_pinned"], marked8p, ext8p = run(pin_in, pin_out, True)
    out["f32_pageable"], _, _ = run(img32, np.empty_like(img32), False)
    # two host threads with a context each (ctypes releases the GIL inside the library): one thread's transfers run
    # beside the other's kernels -- what a caller with a queue of images gets from the same single-image API
    import threading
    n_thr, times = 2, []
    gate = threading.Barrier(n_thr)

    def worker():
        c = wm.Context(ctx.device_id)
        pi, po = c.pinned_empty(img8.shape, np.uint8), c.pinned_empty(img8.shape, np.uint8)
        pi[...] = img8
        def once():
            m = wm.Writer(pi, wcfg, c).mark_rgb8([mark], out=po)
            e = wm.Reader.base(pi, rcfg, c).extract(wm.Reader.derived(m, c, prec), K)
            return wm.Tester(e, c).similarity(mark).similarity
        once(); once()
        gate.wait()
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        times.append(time.perf_counter() - t0)
        del pi, po
        c.close()
    threads = [threading.Thread(target=worker) for _ in range(n_thr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if len(times) == n_thr:
        out["rgb8_pinned_two_threads"] = {"threads": n_thr, "embed_extract_mpix_s": round(px * reps * n_thr / max(times), 1),
                                          "note": "one context per host thread, same GPU; aggregate over both threads"}
    # the same caller with a QUEUE of images: the streaming entry points (ssw_batch_embed_host_rgb8 / _extract_host_rgb8,
    # csrc/ssw_stream.hip) -- n host images in one call, groups of >= 8 frames share the GEMM launches, uploads / kernels /
    # downloads of consecutive groups overlap.  Pinned and pageable frames; PCIe included; bit-identical to the handles.
    n_stream = 64
    smarks = np.random.default_rng(777).standard_normal((n_stream, K)).astype(np.float32)
    smarks[0] = mark

    def stream(frames, outs):
        n_s, mk = len(frames), smarks[:len(frames)]
        wm.mark_many(frames[:8], smarks[:8], wcfg, ctx, out=outs[:8])                    # warm-up (buffers, bases)
        wm.extract_many(frames[:8], outs[:8], K, smarks[:8], rcfg, ctx)
        best = None
        for _ in range(3):                               # best of three calls (the link and the host vary from call to call)
            ctx.transfer_stats(reset=True)
            ta = time.perf_counter()
            wm.mark_many(frames, mk, wcfg, ctx, out=outs)
            tb = time.perf_counter()
            ext, sims = wm.extract_many(frames, outs, K, mk, rcfg, ctx)
            tc = time.perf_counter()
            if best is None or tc - ta < best[2] - best[0]:
                best, st = (ta, tb, tc), ctx.transfer_stats()
        t0, t1, t2 = best
        return {"frames": n_s, "calls": 3, "embed_ms_per_frame": round((t1 - t0) / n_s * 1e3, 3), "extract_ms_per_frame": round((t2 - t1) / n_s * 1e3, 3),
                "embed_mpix_s": round(px * n_s / (t1 - t0), 1), "extract_mpix_s": round(px * n_s / (t2 - t1), 1),
                "embed_extract_mpix_s": round(px * n_s / (t2 - t0), 1),
                "pcie_bytes_per_frame": int((st["h2d_bytes"] + st["d2h_bytes"]) / n_s),
                "staged_fraction": round(st["staged_bytes"] / max(st["h2d_bytes"] + st["d2h_bytes"], 1.0), 3),
                "sim_mean": round(float(sims.mean()), 4)}, ext
    s_in = [ctx.pinned_empty(img8.shape, np.uint8) for _ in range(n_stream)]
    s_out = [ctx.pinned_empty(img8.shape, np.uint8) for _ in range(n_stream)]
    for i, b in enumerate(s_in):
        b[...] = np.roll(img8, 16 * i, axis=1)            # distinct frames, same statistics
    out["rgb8_pinned_stream"], s_ext = stream(s_in, s_out)
    out["rgb8_pinned_stream"]["frame0_bit_identical_to_handles"] = bool(np.array_equal(s_out[0], marked8p) and np.array_equal(s_ext[0], ext8p))
    pg_in = [np.array(b) for b in s_in[:16]]
    out["rgb8_pageable_stream"], _ = stream(pg_in, [np.empty_like(img8) for _ in range(16)])
    del s_in, s_out, pg_in
    # the handles against the batch entry points on the same bytes (n = 1): bit for bit
    cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, prec)
    marks_dev = torch.from_numpy(mark[None]).to(rgb0_dev.device)
    b_marked = torch.empty_like(frame8_dev)
    b_ext = torch.zeros((1, K), dtype=torch.float32, device=rgb0_dev.device)
    b_sim = torch.zeros((1,), dtype=torch.float32, device=rgb0_dev.device)
    torch.cuda.synchronize()
    check(lib.ssw_batch_embed_rgb8(ctx.handle, C.byref(cfg), frame8_dev.data_ptr(), 1, W, H, marks_dev.data_ptr(), K,
                                   b_marked.data_ptr()), "ssw_batch_embed_rgb8")
    check(lib.ssw_batch_extract_rgb8(ctx.handle, C.byref(cfg), frame8_dev.data_ptr(), b_marked.data_ptr(), 1, W, H, K,
                                     b_ext.data_ptr(), marks_dev.data_ptr(), b_sim.data_ptr()), "ssw_batch_extract_rgb8")
    ctx.synchronize()
    out["bit_identical_to_batch_rgb8"] = bool(np.array_equal(marked8, b_marked.cpu().numpy()) and
                                              np.array_equal(ext8, b_ext.cpu().numpy()[0]) and
                                              np.array_equal(marked8, marked8p) and np.array_equal(ext8, ext8p))
    # what the link itself gives: pinned host memory <-> device, 256 MiB each way
    nbytes = 256 << 20
    pin = ctx.pinned_empty((nbytes,), np.uint8)
    pin[...] = 1
    dbuf = ctx.alloc(nbytes)
    rates = {}
    for name, fn in (("h2d", lambda: check(lib.ssw_copy_to_dev(ctx.handle, dbuf.ptr, pin.ctypes.data, nbytes), "h2d")),
                     ("d2h", lambda: check(lib.ssw_copy_to_host(ctx.handle, pin.ctypes.data, dbuf.ptr, nbytes), "d2h"))):
        fn()
        t0 = time.perf_counter()
        for _ in range(4):
            fn()
        rates[name] = round(4 * nbytes / 1e9 / (time.perf_counter() - t0), 1)
    dbuf.free()
    out["pinned_link_gbs"] = rates
    del pin, pin_in, pin_out
    return out


class Dist:
    """torch.distributed used for what the contract asks of it -- the barrier around the timed region, the
    MAX over ranks of the elapsed time and the gather of per-rank results -- and nothing on the data path.
    Backend "nccl" (= RCCL) with device tensors; SSW_BENCH_DIST_BACKEND=gloo (CPU tensors) lets tests run
    several ranks on ONE GPU, which RCCL refuses."""

    def __init__(self, rank, world, dev):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world
        backend = os.environ.get("SSW_BENCH_DIST_BACKEND", "nccl")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
            self.tdev = dev
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            self.tdev = torch.device("cpu")
        self.backend = backend

    def barrier(self):
        self.dist.barrier()

    def max(self, x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, arr: np.ndarray):
        """all_gather of equally shaped arrays -> list indexed by rank."""
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.tdev)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.cpu().numpy() for o in out]

    def close(self):
        self.dist.destroy_process_group()


def marks_for_frames(seed: int, first_frame: int, n: int, k: int) -> torch.Tensor:
    """One N(0,1) mark per GLOBAL frame index, so that a sharded run and a single-process run of the same
    frames use the same marks (tests compare their similarities)."""
    g = torch.Generator()
    rows = []
    for f in range(first_frame, first_frame + n):
        g.manual_seed(seed * 1000003 + f)
        rows.append(torch.randn(k, generator=g, dtype=torch.float32))
    return torch.stack(rows) if rows else torch.zeros((0, k), dtype=torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4], default=None,
                    help="BASELINE.json configs[i] (per-GPU shard); default: configs[3], the metric's configuration")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU (configs[3]: 2048 frames / 8 GPUs = 256)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=0,
                    help="frames per internal pass (0 = the library's automatic choice, ~2^30 pixels; bounds the workspace)")
    ap.add_argument("--precision", choices=["f32", "f64"], default="f64",
                    help="headline precision: f64 = canonical (bit-parity with the CPU path), f32 = fast")
    ap.add_argument("--alt", action="store_true",
                    help="also time the OTHER precision (f32 MFMA chains for the default f64): a non-parity path -- its extracted "
                         "marks miss the 1e-5 bar (max <= 2e-3) and since r3 it is also the slower one; off by default")
    ap.add_argument("--no-alt", action="store_true", help="(default since r4; kept for old command lines)")
    ap.add_argument("--attack-resize", action="store_true",
                    help="configs[4] flow on 8-bit frames: embed -> into_rgb8 -> CatmullRom 1/8 down + up -> extract")
    ap.add_argument("--embed-only", action="store_true", help="configs[1] flow: Writer::new + mark only (DCT2 -> embed -> DCT3)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="frame-parallel CPU baseline also on all host cores (minutes)")
    ap.add_argument("--no-fold", action="store_true", help="dense basis GEMMs instead of the even/odd-folded ones")
    ap.add_argument("--no-timers-off-leg", action="store_true", help="skip the short re-measurement with the stage timers off")
    ap.add_argument("--no-serial-leg", action="store_true", help="skip the short re-measurement with one chunk at a time on one stream")
    ap.add_argument("--no-overlap", action="store_true", help="headline with one chunk at a time on one stream")
    ap.add_argument("--no-stage-timers", action="store_true", help="no hipEvent pairs in the timed region (no roofline numbers)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch frames on EVERY GPU (default); strong: --total-batch frames split over the GPUs")
    ap.add_argument("--total-batch", type=int, default=None,
                    help="strong scaling: frames of the whole job, contiguous block split frame b -> rank floor(b*N/B) "
                         "(default: the config's per-GPU batch, i.e. the N=1 workload)")
    ap.add_argument("--no-handle-leg", action="store_true", help="skip the single-image handle API leg (PCIe-inclusive)")
    ap.add_argument("--no-full-transform-leg", action="store_true",
                    help="skip the re-measurement with the pruned derived transform off (the reference's 4 full transforms)")
    ap.add_argument("--dump", default=None, help="rank 0 writes the gathered per-frame sims / extracted marks here (.npz)")
    args = ap.parse_args()

    preset = BASELINE_CONFIGS[args.config or 3]
    for name in ("batch", "width", "height", "k"):
        if getattr(args, name) is None:
            setattr(args, name, preset[name])
    if args.config is not None:
        args.attack_resize = args.attack_resize or preset["flow"] == "attack"
        args.embed_only = args.embed_only or preset["flow"] == "embed"

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))        # parent: no GPU call before or after this point

    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    dev_index = 0 if os.environ.get("SSW_BENCH_SHARE_DEVICE") else local_rank
    if n_dev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if dev_index >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but this node has {n_dev} GPU(s)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("SSW_FORCE_DIST"):      # the env switch exercises the RCCL path on one GPU
        dist = Dist(rank, world, dev)

    import spread_spectrum_watermarking_amd as wm
    from spread_spectrum_watermarking_amd import _lib as L
    from spread_spectrum_watermarking_amd.api import check

    lib = L.load()
    ctx = wm.Context(dev_index)
    ctx.set_chunk_frames(args.chunk)
    fold_level = 0 if args.no_fold else int(os.environ.get("SSW_FOLD_LEVEL", str(L.DCT_FOLDING_DEFAULT)))
    ctx.set_dct_folding(fold_level)
    ctx.set_overlap(not args.no_overlap)
    W, H, K = args.width, args.height, args.k
    if args.scaling == "strong":
        total_frames = args.total_batch or args.batch
        first_frame, hi = shard_frames(total_frames, world, rank)       # contiguous block split
        B = hi - first_frame
        if total_frames < world:
            raise SystemExit(f"--total-batch {total_frames} < {world} ranks")
    else:
        B = args.batch
        total_frames = B * world
        first_frame = rank * B                   # global frame index of this rank's shard (weak scaling)
    chunk_eff = ctx.pass_frames(B, W, H)          # frames per internal pass (automatic: ~2^30 pixels)
    global FUSED_FORWARD
    plan = ctx.transform_plan(min(chunk_eff, B), W, H)
    FUSED_FORWARD = plan["fused_cols"]
    workload_tag = (f"configs[{args.config or 3}] of BASELINE.json: \"{preset['quote']}\"" if (args.config is not None or
                    (B, W, H, K) == (preset["batch"], preset["width"], preset["height"], preset["k"])) else "custom shape")

    # ---- inputs resident in HBM ------------------------------------------------------------------
    rgb = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
    rgb_out = torch.empty_like(rgb)
    marks_host = marks_for_frames(args.seed, first_frame, B, K)
    marks = marks_host.to(dev)
    extracted = torch.zeros((B, K), dtype=torch.float32, device=dev)
    sims = torch.zeros((B,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    check(lib.ssw_synth_frames(ctx.handle, args.seed, first_frame, B, W, H, rgb.data_ptr()), "ssw_synth_frames")
    ctx.synchronize()

    def rank_report(elapsed_own):
        """ranks_seen / per-rank rates from an all-gather; exits non-zero when a rank is missing."""
        mine = np.array([rank, dev_index, B, float(B) * W * H * args.steps / 1e6 / elapsed_own], dtype=np.float64)
        rows = dist.gather(mine) if dist is not None else [mine]
        seen = sorted(int(r[0]) for r in rows)
        rep = {"ranks_seen": seen, "devices": [int(r[1]) for r in rows],
               "mpix_per_s_per_rank": [round(float(r[3]), 2) for r in rows],
               "dist_backend": dist.backend if dist is not None else None}
        if seen != list(range(args.gpus)):
            raise SystemExit(f"rank {rank}: expected ranks {list(range(args.gpus))}, saw {seen}")
        return rep

    def dump(sims_host, ext_host):
        if not args.dump:
            return
        if dist is None:
            all_s, all_e = [sims_host], [ext_host]
        else:                                        # strong scaling: shards may differ by one frame -> pad, gather, trim
            counts = [int(c[0]) for c in dist.gather(np.array([B], dtype=np.int64))]
            top = max(counts)
            pad_s = np.full((top,), np.nan, np.float32); pad_s[:B] = sims_host
            pad_e = np.full((top, K), np.nan, np.float32); pad_e[:B] = ext_host
            all_s = [a[:c] for a, c in zip(dist.gather(pad_s), counts)]
            all_e = [a[:c] for a, c in zip(dist.gather(pad_e), counts)]
        if rank == 0:
            np.savez(args.dump, sims=np.concatenate(all_s), extracted=np.concatenate(all_e))

    if args.attack_resize:
        run_attack_resize(args, lib, L, ctx, check, dist, dev, rank, world, rgb, rgb_out, marks, marks_host,
                          extracted, sims, B, total_frames, W, H, K, workload_tag, rank_report, dump)
        if dist is not None:
            dist.barrier()                   # rank 0 runs the CPU baseline / parity legs before it gets here
            dist.close()
        ctx.close()
        return

    embed_only = args.embed_only

    def measure(prec_name, overlap=not args.no_overlap):
        """One timed region in the given precision; returns (own s, max-over-ranks s, stage timings, prune stats, sims, extracted)."""
        cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64 if prec_name == "f64" else L.PRECISION_F32)
        ctx.set_overlap(overlap)

        def step():
            check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg), rgb.data_ptr(), B, W, H, marks.data_ptr(), K,
                                      rgb_out.data_ptr(), None, None), "ssw_batch_embed")
            if not embed_only:
                check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), rgb.data_ptr(), rgb_out.data_ptr(), B, W, H, K,
                                            extracted.data_ptr(), marks.data_ptr(), sims.data_ptr()), "ssw_batch_extract")

        own, elapsed, stage, prune = timed_region(args, ctx, dist, step)
        ctx.set_overlap(not args.no_overlap)
        if embed_only:                     # configs[1]: verify the round trip with one untimed extraction
            check(lib.ssw_batch_extract(ctx.handle, C.byref(cfg), rgb.data_ptr(), rgb_out.data_ptr(), B, W, H, K,
                                        extracted.data_ptr(), marks.data_ptr(), sims.data_ptr()), "ssw_batch_extract")
            ctx.synchronize()
        sims_host = sims.cpu().numpy()
        ext_host = extracted.cpu().numpy()
        # every frame of the batch: a wrong index or a wrong coefficient anywhere shows up as an O(1) error in
        # the extracted mark (Option2, alpha 0.1, no quantisation: the round trip returns the mark to ~1e-2)
        err = np.abs(ext_host - marks_host.numpy()).max(axis=1)
        norms = marks_host.norm(dim=1).numpy()
        if not (np.all(err < 0.1) and np.all(sims_host > 0.97 * norms)):
            bad = int(np.argmax(err))
            raise SystemExit(f"rank {rank}: round-trip check failed ({prec_name}): frame {bad} max|extracted - mark| = "
                             f"{err[bad]:.3g}, min sim/|mark| = {(sims_host / norms).min():.3f}")
        return own, elapsed, stage, prune, sims_host, ext_host

    steps = args.steps
    px_total = float(B) * W * H * steps                      # this rank's pixels
    job_px_total = float(total_frames) * W * H * steps       # the whole job's (weak: world x the rank's)
    transforms_per_step = 2 if embed_only else 4     # DCT2, DCT3 (embed) [, DCT2, DCT2 (extract)]
    dense_flop_per_step = transforms_per_step * 2.0 * B * W * H * (W + H)      # F2D = 2 W H (W + H)

    def kernel_report(prec_name, stage, steps):
        """Per-kernel achieved rates from the live event timers and the work the library counted for the
        same launches.  GEMMs: EXECUTED flop / time is the utilisation (folding and the split odd halves execute about a tenth of
        the dense 2*lines*N*N, the pruned derived transform a few percent of it); the reference's dense
        flop / time is reported separately as "effective"."""
        peak = PEAK_F64_MFMA_TFLOPS if prec_name == "f64" else PEAK_F32_MFMA_TFLOPS
        kernels = {}
        for name, key in (("dct_rows", "dct_row"), ("dct_cols", "dct_col")):
            ms, n = stage[key]["ms"], max(stage[key]["launches"], 1)
            tf = rate(stage[key]["work"], ms) / 1e12
            mk = key + "_main"
            mms, mn = stage[mk]["ms"], max(stage[mk]["launches"], 1)
            kernels[name] = {"tflops": round(tf, 2), "frac_mfma": round(tf / peak, 4), "ms_per_step": round(ms / steps, 3),
                             "timed_stages": n, "executed_flop_per_step": stage[key]["work"] / steps,
                             "main_launch": {"avg_ms": round(mms / mn, 4), "launches": mn,
                                             "flop_per_launch": stage[mk]["work"] / mn,
                                             "tflops": round(rate(stage[mk]["work"], mms) / 1e12, 2)}}
        gemm_ms = stage["dct_row"]["ms"] + stage["dct_col"]["ms"]
        executed = stage["dct_row"]["work"] + stage["dct_col"]["work"]
        kernels["dct_all"] = {"executed_fraction_of_dense": round(executed / (dense_flop_per_step * steps), 4) if steps else 0.0,
                              "effective_dense_tflops": round(rate(dense_flop_per_step * steps, gemm_ms) / 1e12, 2)}
        kernels.update(hbm_report(stage, ["rgb_to_yiq", "dct_prep", "select", "yiq_to_rgb"]))
        roofline, roofline_hbm = family_rooflines(prec_name, stage, steps, W, H, chunk_eff, B)
        return kernels, (roofline, roofline_hbm), {k: round(v["ms"] / steps, 3) for k, v in stage.items()}

    own, elapsed, stage, prune, sims_host, ext_host = measure(args.precision)
    kernels, (roofline, roofline_hbm), stage_ms = kernel_report(args.precision, stage, steps)
    ranks = rank_report(own)
    dump(sims_host, ext_host)

    # the same workload with one chunk at a time on one stream: per-kernel timings without co-running kernels
    serial = None
    if not args.no_serial_leg:
        keep = args.steps
        args.steps = max(1, min(args.steps, 5))
        s_own, s_elapsed, s_stage, _, s_sims, s_ext = measure(args.precision, overlap=False)
        s_kernels, (s_roofline, s_roofline_hbm), s_stage_ms = kernel_report(args.precision, s_stage, args.steps)
        serial = {"value": round(float(total_frames) * W * H * args.steps / 1e6 / s_elapsed, 2), "unit": "Mpix/s",
                  "ms_per_step": round(s_elapsed / args.steps * 1e3, 3), "steps": args.steps,
                  "roofline_step": step_roofline(args.precision, s_stage, args.steps, s_elapsed / args.steps * 1e3),
                  "roofline": s_roofline, "roofline_hbm": s_roofline_hbm, "kernels": s_kernels, "stage_ms_per_step": s_stage_ms,
                  "bit_identical_to_overlapped": bool(np.array_equal(s_sims, sims_host) and np.array_equal(s_ext, ext_host))}
        args.steps = keep

    timers_off = None
    if not args.no_timers_off_leg and not args.no_stage_timers:
        keep = (args.steps, args.no_stage_timers)
        args.steps, args.no_stage_timers = max(1, min(args.steps, 5)), True
        _, o_elapsed, _, _, _, _ = measure(args.precision)
        timers_off = {"value": round(float(total_frames) * W * H * args.steps / 1e6 / o_elapsed, 2), "unit": "Mpix/s",
                      "ms_per_step": round(o_elapsed / args.steps * 1e3, 3), "steps": args.steps,
                      "note": "same workload with the stage timers disabled: what the event pairs in the timed region cost"}
        args.steps, args.no_stage_timers = keep

    # the reference's literal unit of work: Reader::derived transforms the whole frame, 4 full
    # 2-D transforms per embed + extract -- the same step with the pruned derived transform switched off
    full = None
    if not args.no_full_transform_leg and not embed_only:
        keep = args.steps
        args.steps = max(1, min(args.steps, 5))
        ctx.set_prune(False)
        _, f_elapsed, f_stage, f_prune, f_sims, f_ext = measure(args.precision)
        ctx.set_prune(True)
        f_kernels, _, f_stage_ms = kernel_report(args.precision, f_stage, args.steps)
        full = {"value": round(float(total_frames) * W * H * args.steps / 1e6 / f_elapsed, 2), "unit": "Mpix/s",
                "ms_per_step": round(f_elapsed / args.steps * 1e3, 3), "steps": args.steps,
                "transforms_per_frame": 4,
                "executed_fraction_of_dense": f_kernels["dct_all"]["executed_fraction_of_dense"],
                "effective_dense_tflops": f_kernels["dct_all"]["effective_dense_tflops"],
                "dct_rows_frac_mfma": f_kernels["dct_rows"]["frac_mfma"], "dct_cols_frac_mfma": f_kernels["dct_cols"]["frac_mfma"],
                "stage_ms_per_step": f_stage_ms, "pruned_derived_transform": f_prune,
                "bit_identical_to_headline": bool(np.array_equal(f_sims, sims_host) and np.array_equal(f_ext, ext_host)),
                "note": "ssw_ctx_set_prune(0): every derived frame fully transformed like Reader::derived "
                        "(src/algorithm.rs:469-480); the headline computes only the frequency columns extract reads"}
        args.steps = keep

    # configs[1] (one frame, launch-bound): the same step captured ONCE into a HIP graph by the caller (the device-pointer entry
    # points only enqueue, include/ssw.h stream contract) and replayed: what a caller with a fixed frame buffer gets once the
    # ~25 launches of a single-frame embed cost one graph launch.  Reported beside the eager headline, never instead of it.
    graph = None
    if embed_only and B <= 2 and not args.no_timers_off_leg:
        try:
            cfg_g = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1, L.PRECISION_F64 if args.precision == "f64" else L.PRECISION_F32)
            gs = torch.cuda.Stream()
            ctx.synchronize(); torch.cuda.synchronize()
            eager_out = rgb_out.clone()
            ctx.set_stream(gs.cuda_stream)
            rgb_out.zero_(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=gs, capture_error_mode="relaxed"):
                check(lib.ssw_batch_embed(ctx.handle, C.byref(cfg_g), rgb.data_ptr(), B, W, H, marks.data_ptr(), K,
                                          rgb_out.data_ptr(), None, None), "ssw_batch_embed (capture)")
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            n_rep = max(20, args.steps)
            t0 = time.perf_counter()
            for _ in range(n_rep):
                g.replay()
            torch.cuda.synchronize()
            g_s = time.perf_counter() - t0
            graph = {"value": round(float(total_frames) * W * H * n_rep / 1e6 / g_s, 2), "unit": "Mpix/s", "ms_per_step": round(g_s / n_rep * 1e3, 4),
                     "replays": n_rep, "bit_identical_to_eager": bool(torch.equal(rgb_out, eager_out)),
                     "note": "torch.cuda.CUDAGraph capture of ssw_batch_embed on the caller's stream (ssw_ctx_set_stream), replayed; no stage timers inside"}
            del g
        except Exception as e:                       # a capture failure must not cost the headline
            graph = {"error": str(e)[:200]}
        finally:
            ctx.set_stream(None)
            ctx.synchronize(); torch.cuda.synchronize()

    alt = None
    if args.alt and not args.no_alt and not L.all_strategies():
        print("bench.py: --alt needs the diagnostic build (make ALL_STRATEGIES=1; SSW_LIB_PATH=.../libssw_hip_all.so): the default "
              "library runs SSW_PRECISION_F32 on the dense kernels; leg skipped", file=sys.stderr)
    elif args.alt and not args.no_alt:
        alt_name = "f32" if args.precision == "f64" else "f64"
        _, alt_elapsed, alt_stage, _, alt_sims, _ = measure(alt_name)
        alt_kernels, (alt_roofline, _), _ = kernel_report(alt_name, alt_stage, steps)
        alt = {"dtype": alt_name, "value": round(job_px_total / 1e6 / alt_elapsed, 2), "unit": "Mpix/s",
               "ms_per_step": round(alt_elapsed / steps * 1e3, 3), "roofline": alt_roofline,
               "kernels": {k: alt_kernels[k] for k in ("dct_rows", "dct_cols")},
               "sim_mean": round(float(alt_sims.mean()), 4),
               "max_abs_sim_diff_vs_headline": float(np.abs(alt_sims - sims_host).max()),
               "parity": "NOT a parity path: extracted marks within median 1e-5 / max 2e-3 of the oracle only (DESIGN.md 5)"}

    result = None
    if rank == 0:
        mpix = job_px_total / 1e6
        flow = "embed (Writer::new + mark: DCT2 -> embed -> DCT3 round trip)" if embed_only else "embed+extract+similarity"
        result = {
            "metric": ("Mpixels/sec embed + IDCT round trip" if embed_only else
                       "Mpixels/sec embed+extract (4K batch)" if (W, H) == (3840, 2160) else f"Mpixels/sec embed+extract ({W}x{H} batch)"),
            "value": round(mpix / elapsed, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": (f"batch={B}/GPU" if args.scaling == "weak" else f"batch={total_frames} over {world} GPU(s)") +
                                   f" {W}x{H} f32 frames, {K}-coeff mark, {flow}; {workload_tag}",
                       "frames_per_gpu": B, "total_frames": total_frames, "width": W, "height": H, "k": K, "alpha": 0.1,
                       "method": "Option2", "ordering": "Energy", "chunk_frames": chunk_eff,
                       "transform_plan": {k: v for k, v in plan.items()},
                       "dct_folding_level": fold_level, "dct_odd_split": "on (f64: odd halves as rotated quarter-length cosine + sine pairs, deep pre-passes; DESIGN.md 4.1)" if args.precision == "f64" else "off (f32 twin: exact-operand folding)",
                       "overlap": "one chunk at a time on one stream" if args.no_overlap else "two chunks in flight on two streams",
                       "parallelism": f"frame-sharded x{world}, no collectives"},
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "roofline_step": step_roofline(args.precision, stage, steps, elapsed / steps * 1e3),
            "kernels": kernels,
            "stage_ms_per_step": stage_ms,
            "pruned_derived_transform": prune,
            "ranks": ranks,
            "sim_mean": round(float(sims_host.mean()), 4),
        }
        if serial is not None:
            result["serialized"] = serial
        if timers_off is not None:
            result["timers_off"] = timers_off
        if graph is not None:
            result["graph_replay"] = graph
        if full is not None:
            result["full_transform"] = full
        if alt is not None:
            result["alt_precision"] = alt

    # ---- the drop-in handle API, one host image per call (PCIe-inclusive; rank 0) -------------------
    if rank == 0 and not args.no_handle_leg:
        result["handle_api"] = handle_api_leg(wm, L, lib, ctx, check, rgb[0], W, H, K, args.precision)

    # ---- CPU baseline: the oracle (faithful mode) on a bounded sample, on rank 0 (at N > 1 after the timed
    # region and its closing barrier: the other ranks wait in the final barrier below) -----------------
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as O
        if alt is not None or serial is not None or timers_off is not None or full is not None:   # the headline precision's outputs again
            measure_cfg = L.Config(L.ORDER_ENERGY, L.OPTION2, 0.1,
                                   L.PRECISION_F64 if args.precision == "f64" else L.PRECISION_F32)
            check(lib.ssw_batch_embed(ctx.handle, C.byref(measure_cfg), rgb.data_ptr(), B, W, H, marks.data_ptr(), K,
                                      rgb_out.data_ptr(), None, None), "ssw_batch_embed")
            check(lib.ssw_batch_extract(ctx.handle, C.byref(measure_cfg), rgb.data_ptr(), rgb_out.data_ptr(), B, W, H, K,
                                        extracted.data_ptr(), marks.data_ptr(), sims.data_ptr()), "ssw_batch_extract")
            ctx.synchronize()
        frame0 = rgb[0].cpu().numpy()
        mark0 = marks_host[0].numpy()
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
        # same work, one frame per thread (the reference itself is single-threaded; this is the frame-parallel
        # upper bound of SURVEY 8(d)).  ctypes releases the GIL in the C calls.  Thread counts: 16 and a quarter of
        # the cores by default; --cpu-all-cores adds ALL the host's logical cores (capped only by host memory:
        # ~0.7 GB of planes and sort keys per 4K frame in flight).  The full sort is memory-bound: on the 256-thread
        # host of the GPU boxes all cores take 145 s for 14.6 Mpix/s where 16 threads reach 25.7 in 5 s
        # (profiles/r2_bench_config3_n1.json holds that sweep), so the long leg is opt-in.  Every count measured is
        # reported, the best one is the figure.
        import concurrent.futures as cf
        n_cores = os.cpu_count() or 1
        try:
            avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable:")][0]
        except (OSError, IndexError, ValueError):
            avail = 16 << 30
        per_thread = 80 * W * H                        # bytes: rgb out + planes + sort keys (16 B/coefficient)
        cap = max(1, int(0.5 * avail / per_thread))

        def one(_):
            m = O.embed_frame(frame0, mark0, backend=O.BACKEND_F32, full_sort=True)
            O.extract_frame(frame0, m, mark0, backend=O.BACKEND_F32, full_sort=True)
        sweep = []
        counts = {min(16, n_cores, cap), min(max(n_cores // 4, 1), cap)}
        if args.cpu_all_cores:
            counts.add(min(n_cores, cap))
        if world > 1:                                  # the other ranks are waiting: the single-thread figure only
            counts = set()
        for n_thr in sorted(counts):
            t0 = time.perf_counter()
            with cf.ThreadPoolExecutor(n_thr) as ex:
                list(ex.map(one, range(n_thr)))
            par_s = time.perf_counter() - t0
            sweep.append({"threads": n_thr, "value": round(n_thr * W * H / 1e6 / par_s, 4), "seconds": round(par_s, 1)})
        if sweep:
            best = max(sweep, key=lambda e: e["value"])
            result["cpu_baseline_parallel"] = {
                "value": best["value"], "unit": "Mpix/s", "cores": best["threads"], "kind": "port",
                "sample": f"{best['threads']} frames {W}x{H}, one per thread, same faithful pipeline, {best['seconds']} s; "
                          f"host has {n_cores} logical cores", "sweep": sweep,
            }
        # untimed: the oracle's correctly rounded (f64-backend) pipeline = what the canonical precision must equal,
        # on the first and the last frame of the batch (the last one sits in the last chunk of the pipeline)
        gpu_ext = extracted.cpu().numpy()
        sims_now = sims.cpu().numpy()
        checks = []
        for f in sorted({0, B - 1}):
            frame = rgb[f].cpu().numpy()
            mk = marks_host[f].numpy()
            marked_f = rgb_out[f].cpu().numpy()
            ref_marked = O.embed_frame(frame, mk, backend=O.BACKEND_F64, full_sort=False)
            ref_ext, ref_sim = O.extract_frame(frame, ref_marked, mk, backend=O.BACKEND_F64, full_sort=False)
            checks.append({"frame": f, "sim_gpu": float(sims_now[f]), "sim_cpu_exact": float(ref_sim),
                           "sim_delta_vs_cpu_exact": abs(float(sims_now[f]) - float(ref_sim)),
                           "marked_frame_max_abs_diff_vs_cpu_exact": float(np.abs(marked_f - ref_marked).max()),
                           "marked_frame_bit_identical_fraction": float(np.mean(marked_f == ref_marked)),
                           "extracted_max_abs_diff_vs_cpu_exact": float(np.abs(gpu_ext[f] - ref_ext).max())})
        result["parity"] = {"precision": args.precision, "sim_cpu_f32fft": float(cpu_sim),
                            "sim_delta_vs_cpu_f32fft": abs(float(sims_now[0]) - float(cpu_sim)),
                            # frame 0, each side extracting from its own marked frame: the f32-FFT backend (the stand-in for
                            # rustdct) is itself 3e-5 .. 1.6e-4 away from the exact transform in the worst element
                            "extracted_max_abs_diff_vs_cpu_f32fft": float(np.abs(gpu_ext[0] - cpu_ext).max()),
                            "extracted_median_abs_diff_vs_cpu_f32fft": float(np.median(np.abs(gpu_ext[0] - cpu_ext))),
                            "frames": checks}

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()                       # rank 0 arrives after its host-side legs
        dist.close()
    ctx.close()


if __name__ == "__main__":
    main()
