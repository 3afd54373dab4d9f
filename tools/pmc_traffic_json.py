#!/usr/bin/env python3
"""Builds profiles/<tag>_pmc_traffic.json from the per-kernel PMC summaries of tools/collect_profiles.sh
(<dir>/<tag>_pmc_fetch.txt and <tag>_pmc_write.txt, written by tools/pmc_summary.py).

HBM-side bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024: on gfx950 FETCH_SIZE reports half
the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are
included (memory-side L2 requests), so this is an upper bound on DRAM traffic.
Besides the per-instance entries the file carries two FAMILY sums per 128-frame (chunk_frames) step -- what bench.py's
`roofline` (every pair_gemm_f64_kernel launch) and `roofline_hbm` (every *prep16* launch) blocks quote: bytes summed over
all launches of the family in one step of the profiled run (`steps_profiled` steps of one pass each).
usage: pmc_traffic_json.py <dir> <tag> <width> <height> <chunk_frames> <commit> [steps_profiled = 2]"""
import ast
import json
import sys

d, tag, W, H, chunk, commit = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
steps_profiled = int(sys.argv[7]) if len(sys.argv) > 7 else 2          # collect_profiles.sh: --steps 1 --warmup 1


def load(path):
    out = {}
    for line in open(path):
        if "{" not in line:
            continue
        name, rest = line.split(" {", 1)
        body, n = rest.rsplit("} n=", 1)
        out[name.strip()] = (ast.literal_eval("{" + body + "}"), int(n))
    return out


def strip_tile(table):      # instance names carry the block-tile template argument: "<..., 0, 128>" -> "<..., 0>"
    out = {}
    for name, v in table.items():
        if name.startswith("pair_gemm_") and name.endswith(", 128>"):
            name = name[:-len(", 128>")] + ">"
        out[name] = v
    return out


fetch, write = strip_tile(load(f"{d}/{tag}_pmc_fetch.txt")), strip_tile(load(f"{d}/{tag}_pmc_write.txt"))
try:        # MFMA pipe occupancy in cycles (independent of the clock the chip holds): SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over
            # GRBM_GUI_ACTIVE / 8 XCDs (rocprofv3 reports sums)
    mfma = strip_tile(load(f"{d}/{tag}_pmc_mfma.txt"))
except OSError:
    mfma = {}
lines_r, lines_c = chunk * H, chunk * W
lines_p = chunk * 16 * (-(-(H // 16) // 8) * 8) if H % 16 == 0 else lines_r      # unit-ordered, padded lines of a fused row pass
esz = 8
K8 = lambda n: -(-(n // 8) // 16) * 16        # padded sum length of the n/8-wide operand planes
K16 = lambda n: -(-(n // 16) // 16) * 16      # ... of the n/16-wide ones
# level 2 (csrc/dct_pair_prep.hip dct_pair_efold / dct_pair_efold_cols): rows of 1280 columns or more, columns of 720 rows or more
# (a multiple of 16): every launch sums n/16 terms over n/16 pairs; below, the full-length classes sum n/8 terms over n/8 pairs
l2r, l2c = W >= 1280 and W % 64 == 0, H >= 720 and H % 16 == 0
PR, KR = (W // 16, K16(W)) if l2r else (W // 8, K8(W))
PC, KC = (H // 16, K16(H)) if l2c else (H // 8, K8(H))
names = {   # instance in the rocprofv3 output -> (label used by bench.py / DESIGN.md, algorithmic bytes per launch, note)
    "pair_gemm_f64_kernel<false, 0, false, 4>": ("pair_gemm_f64_kernel<rows, split odd half, class O (level 2: rotated, '+' launch)>",
        2 * lines_r * KR * esz + 2 * (PR + 1) * KR * esz + lines_r * (2 * PR) * 4,
        "two operand planes (k-blocked f64) x cosine / sine rows -> two frequencies per pair (f32); level 2: W/16 + 1 pairs in W/16 slots"),
    # r5: the fused forward transform -- the row launches' epilogue writes the column operands (EPI_FWD_COLOP = 7): operand lines
    # are the frame's rows in unit order, padded to whole k-blocks of units (H/16 -> a multiple of 8)
    "pair_gemm_f64_kernel<false, 7, false, 4>": ("pair_gemm_f64_kernel<rows, fused column operands, class O rotated '+' launch>",
        2 * lines_p * KR * esz + 2 * (PR + 1) * KR * esz + lines_r * (2 * PR) * esz,
        "r5: two row operand planes in, the launch's two frequencies of every line out as entries of the sixteen f64 column-operand planes (8 B per output instead of 4)"),
    "pair_gemm_f64_kernel<false, 7, false, 3>": ("pair_gemm_f64_kernel<rows, fused column operands, other classes>",
        2 * lines_p * KR * esz + 2 * PR * KR * esz + lines_r * (2 * PR) * esz,
        "r5: the other seven launches of a fused forward row pass: mean over launches"),
    "pair_gemm_f64_kernel<false, 0, false, 3>": ("pair_gemm_f64_kernel<rows, split odd halves, other classes>",
        2 * lines_r * KR * esz + 2 * PR * KR * esz + lines_r * (2 * PR) * 4,
        "the other split classes of the forward row pass (level 2: six launches of the same size) and the gathered launches of the pruned transform: mean over launches"),
    "pair_gemm_f64_kernel<false, 0, false, 1>": ("pair_gemm_f64_kernel<rows, (SSS, SS-) launch>",
        2 * lines_r * KR * esz + 2 * PR * KR * esz + lines_r * (2 * PR) * 4, "frequencies 0 and 4 mod 8 (level 2: R1 folded, 0 and 8 mod 16)"),
    "pair_gemm_f64_kernel<true, 0, false, 3>": ("pair_gemm_f64_kernel<cols, split odd halves>",
        2 * lines_c * KC * esz + 2 * PC * KC * esz + lines_c * (2 * PC) * 4, "forward column pass, the split classes (level 2: seven launches of the same size): mean over launches"),
    "pair_gemm_f64_kernel<true, 0, false, 1>": ("pair_gemm_f64_kernel<cols, (SSS, SS-) launch>",
        2 * lines_c * KC * esz + 2 * PC * KC * esz + lines_c * (2 * PC) * 4, "forward column pass, frequencies 0 and 4 mod 8 (level 2: 0 and 8 mod 16)"),
    "pair_gemm_f64_kernel<false, 4, false, 0>": ("pair_gemm_f64_kernel<rows, inverse split odd half>",
        2 * lines_r * KR * esz + 2 * PR * KR * esz + lines_r * (2 * PR) * esz + lines_r * (4 * PR) * 4,
        "inverse row pass: one launch of the split odd part + the unrounded even half E -> four outputs per pair (f32); level 2: four such launches"),
    "pair_gemm_f64_kernel<true, 5, false, 0>": ("pair_gemm_f64_kernel<cols, inverse split odd half + yiq->rgb>",
        2 * lines_c * KC * esz + 2 * PC * KC * esz + lines_c * (2 * PC) * esz + lines_c * (4 * PC) * (8 + 12),
        "last pass of Writer::result: split odd part + unrounded even half in, I and Q in, RGB f32 out (level 2: four launches, a quarter of the rows each)"),
    "pair_prep16_rows_kernel<double, 1, false>": ("pair_prep16_rows_kernel<double, rgb>", lines_r * W * (12 + esz),
        "reader: RGB f32 in, the ten f64 operand planes of the deep row pass out"),
    "pair_prep16_rows_kernel<double, 1, true>": ("pair_prep16_rows_kernel<double, rgb, with I/Q>", lines_r * W * (12 + 8 + esz),
        "writer: RGB f32 in, operand planes + I, Q planes out"),
    "pair_prep16_rows_light_kernel<1, false, 8>": ("pair_prep16_rows_light_kernel<rgb>", lines_r * W * (12 + esz),
        "r5, reader: RGB f32 in, the sixteen f64 operand planes of the level-2 row pass out (the < 64-VGPR form, csrc/dct_pair_prep_light.hip)"),
    "pair_prep16_rows_light_kernel<1, true, 8>": ("pair_prep16_rows_light_kernel<rgb, with I/Q>", lines_r * W * (12 + 8 + esz),
        "r5, writer: RGB f32 in, operand planes + I, Q planes out"),
    "prep16_derived_fused_kernel<1>": ("prep16_derived_fused_kernel<rgb>", lines_r * W * 12 + lines_r * 256 * 4,
        "r5: the derived frame's pruned row pass in one kernel (csrc/dct_pair_derived.hip): RGB f32 in, the compact plane (256 columns for k = 1000) out"),
    "pair_prep16_cols_kernel<double>": ("pair_prep16_cols_kernel<double>", lines_r * W * (4 + esz),
        "r3 kernel (SSW_PREP_STAGED=0): f32 plane in, transposed deep f64 operand planes out (mean over launches incl. the narrow pruned ones)"),
    "pair_prep16_inv_rows_kernel<double>": ("pair_prep16_inv_rows_kernel<double>", lines_r * W * (4 + esz), "r3 kernel: coefficient plane in, deep inverse operand planes out"),
    "pair_prep16_inv_cols_kernel<double>": ("pair_prep16_inv_cols_kernel<double>", lines_r * W * (4 + esz), "r3 kernel: the same, transposed"),
    # r4: the LDS-staged forms (csrc/dct_pair_prep_staged.hip); <1 | 2, true> = class-major tiles, deep
    "prep16_cols_staged_kernel<1, true>": ("prep16_cols_staged_kernel<class-major tile, deep>", lines_r * W * (4 + esz),
        "forward column pre-pass: f32 plane (class-major tiles of 128 columns) in, the ten transposed f64 operand planes out"),
    "prep16_cols_staged_kernel<0, true>": ("prep16_cols_staged_kernel<natural, deep>", lines_r * W * (4 + esz), "the same from a natural-order plane (the compact planes of the pruned transform: mean over launches)"),
    "prep16_cols_staged_kernel<0, false>": ("prep16_cols_staged_kernel<natural, semi-deep>", lines_r * W * (4 + esz), "H % 16 != 0 (1080 rows)"),
    "prep16_inv_rows_staged_kernel": ("prep16_inv_rows_staged_kernel", lines_r * W * (4 + esz), "inverse row pre-pass: coefficient plane in, deep inverse operand planes out"),
    "prep16_inv_cols_staged_kernel<2, true>": ("prep16_inv_cols_staged_kernel<class-major tile, deep>", lines_r * W * (4 + esz), "inverse column pre-pass (transposing)"),
    "prep16_inv_cols_staged_kernel<0, true>": ("prep16_inv_cols_staged_kernel<natural, deep>", lines_r * W * (4 + esz), "the same from a natural-order plane"),
    "prep16_inv_cols_staged_kernel<0, false>": ("prep16_inv_cols_staged_kernel<natural, semi-deep>", lines_r * W * (4 + esz), "H % 16 != 0 (1080 rows)"),
    # r4c: the level-2 forms
    "prep16_cols_l2_kernel<1>": ("prep16_cols_l2_kernel<class-major tile>", lines_r * W * (4 + esz),
        "forward column pre-pass at level 2: f32 plane (class-major tiles) in, sixteen transposed f64 operand planes H/16 wide out"),
    "prep16_cols_l2_kernel<0>": ("prep16_cols_l2_kernel<natural>", lines_r * W * (4 + esz), "the same from a natural-order plane (the compact planes of the pruned transform: mean over launches)"),
    "prep16_inv_rows_l2_kernel": ("prep16_inv_rows_l2_kernel", lines_r * W * (4 + esz), "inverse row pre-pass at level 2: coefficient plane in, sixteen operand planes W/16 wide out"),
    "prep16_inv_cols_l2_kernel<2>": ("prep16_inv_cols_l2_kernel<class-major tile>", lines_r * W * (4 + esz), "inverse column pre-pass at level 2 (transposing)"),
    "prep16_inv_cols_l2_kernel<0>": ("prep16_inv_cols_l2_kernel<natural>", lines_r * W * (4 + esz), "the same from a natural-order plane"),
    "select_compact_kernel<true>": ("select_compact_kernel<energy>", lines_r * W * 4, "the one full pass of the top-k selection"),
}
out = {"_how": __doc__.strip().split("usage:")[0].strip(), "commit": commit,
       "workload": {"width": W, "height": H, "chunk_frames": chunk}, "kernels": {}}
for inst, (label, alg, note) in names.items():
    if inst not in fetch or inst not in write:
        continue
    f, w = fetch[inst][0]["FETCH_SIZE"], write[inst][0]
    hbm = int(2 * f * 1024 + w["WRITE_SIZE"] * 1024)
    hit, miss = w.get("TCC_HIT_sum", 0.0), w.get("TCC_MISS_sum", 0.0)
    out["kernels"][label] = {"instance": inst, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w["WRITE_SIZE"], "TCC_HIT": hit, "TCC_MISS": miss,
                             "launches_averaged": fetch[inst][1], "hbm_bytes_per_launch": hbm,
                             "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
                             "algorithmic_bytes_per_launch": int(alg), "traffic_over_algorithmic": round(hbm / alg, 2), "note": note}
    m = mfma.get(inst, ({}, 0))[0]
    if m.get("GRBM_GUI_ACTIVE") and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        out["kernels"][label]["mfma_busy_over_active_cycles"] = round((m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (m["GRBM_GUI_ACTIVE"] / 8.0), 4)
# family sums per step: every launch of every instance whose name matches, bytes = mean x launches (pmc_summary.py prints both)
def family(match):
    tot_f = tot_w = 0.0
    n = 0
    members = {}
    for inst in fetch:
        if match not in inst or inst not in write:
            continue
        f, nf = fetch[inst][0]["FETCH_SIZE"], fetch[inst][1]
        w, nw = write[inst][0]["WRITE_SIZE"], write[inst][1]
        tot_f += f * nf
        tot_w += w * nw
        n += nf
        members[inst] = nf // steps_profiled if steps_profiled else nf
    return {"match": match, "launches_per_step": n // steps_profiled if steps_profiled else n, "members_launches_per_step": members,
            "hbm_bytes_per_step": int((2 * tot_f + tot_w) * 1024 / max(steps_profiled, 1)),
            "fetch_bytes_per_step_x2": int(2 * tot_f * 1024 / max(steps_profiled, 1)), "write_bytes_per_step": int(tot_w * 1024 / max(steps_profiled, 1))}


px_step = chunk * W * H
out["families"] = {"_unit": f"one step = embed + extract of one {chunk}-frame pass ({steps_profiled} steps profiled, sums divided by that)",
                   "gemm": family("pair_gemm_f64_kernel"), "prepass": family("prep16")}
# algorithmic bytes of the pre-pass family per step: writer frame RGB f32 -> operands + I, Q (28 B/px),
# base and derived frames (20 B/px each), forward columns x 2, inverse rows, inverse columns (12 B/px each); the
# column pre-passes of the pruned derived transform work on compact planes (a few percent of a frame, not counted)
# r5: a fused forward transform has no column pre-pass (its launches are absent from the profile): count what ran
n_fwd_cols = out["families"]["prepass"]["members_launches_per_step"].get("prep16_cols_l2_kernel<1>", 0) + \
             out["families"]["prepass"]["members_launches_per_step"].get("prep16_cols_staged_kernel<1, true>", 0)
# r5: the derived frame's row pass in one kernel (csrc/dct_pair_derived.hip) has no pre-pass and no operand planes: its frames
# (12 B/px in, the compact plane out) count for the pre-pass family, whose name match the kernel carries
derived_fused = any("prep16_derived_fused_kernel" in k for k in fetch)
pre_alg = px_step * (28 + 20 + (12 if derived_fused else 20) + (n_fwd_cols + 2) * 12) + (chunk * H * 256 * 4 if derived_fused else 0)
out["families"]["prepass"]["algorithmic_bytes_per_step"] = int(pre_alg)
out["families"]["prepass"]["traffic_over_algorithmic"] = round(out["families"]["prepass"]["hbm_bytes_per_step"] / pre_alg, 3)
# the GEMM family's algorithmic bytes per step, as ssw_ctx_get_traffic counts them (csrc/ssw_pipeline.hip build_pass): per
# pixel of a pass -- operand planes in (8) + result out (f32: 4; fused forward rows: the column operands, 8) + what the
# dependent launches of an inverse pass exchange (A1 1 + 1, T2 2 + 2, E 4 + 4 = 14); the last pass of Writer::result reads
# I, Q (8) and writes RGB f32 (12) instead of 4; the pruned derived row pass reads its operands (8) and writes a compact plane
fused = any(k.startswith("pair_gemm_f64_kernel<false, 7") for k in out["families"]["gemm"]["members_launches_per_step"])
fwd_rows = 16 if fused else 12
gemm_alg = px_step * (2 * (fwd_rows + 12)            # writer and base reader: forward rows + columns
                      + (0 if derived_fused else 8)   # derived frame: operands of the gathered row launches (the compact planes are a few percent)
                      + (8 + 14 + 4) + (8 + 14 + 8 + 12))      # inverse rows; inverse columns with the RGB epilogue
out["families"]["gemm"]["algorithmic_bytes_per_step"] = int(gemm_alg)
out["families"]["gemm"]["traffic_over_algorithmic"] = round(out["families"]["gemm"]["hbm_bytes_per_step"] / gemm_alg, 3)
json.dump(out, sys.stdout, indent=1)
print()
