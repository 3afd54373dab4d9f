"""A bounded, fixed-seed leg of the builder's fuzzers (tools/fuzz_dct.py, tools/fuzz_batch.py) inside `pytest -m gpu`, so
that the driver's GPU test run covers every strategy branch of the transform (deep / semi-deep / first-level split /
exact-operand folding / in-kernel folding / dense; rows or columns first; class-major tiles or natural planes; staged or
r3 pre-passes by shape) and of the batch pipelines (pruned + two lanes against full + one lane) -- not only the
hand-picked shapes of test_gpu_parity.py.  Reference: src/dct2d.rs:83-219 (the transform the oracle restates),
src/algorithm.rs:295-316, :355-379, :529-593 (the batch flows).  Everything goes through the C ABI; the oracle checks."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_batch  # noqa: E402
import fuzz_dct  # noqa: E402

pytestmark = pytest.mark.gpu

# 20 random shapes + two that pin layouts the draw missed: class-major in ONE tile (W % 64 == 0, not % 128: the r3 column
# pre-passes) and a semi-deep column pass behind a deep row pass
DCT_CASES = fuzz_dct.shapes(20, 20261002) + [(272, 320, 2, 77), (1048, 2048, 1, 78)]
BATCH_CASES = fuzz_batch.shapes(6, 20261002)


@pytest.mark.parametrize("kind", ["fwd", "ortho", "inv"])
@pytest.mark.parametrize("case", DCT_CASES, ids=[f"{h}x{w}x{n}" for (h, w, n, _) in DCT_CASES])
def test_random_shapes_through_ssw_dct2d_match_the_oracle(case, kind):
    h, w, n, data_seed = case
    same, err = fuzz_dct.check(h, w, n, data_seed, kind)
    assert same >= fuzz_dct.BAR_IDENTICAL and err <= fuzz_dct.BAR_ERR, (h, w, n, kind, same, err)


@pytest.mark.parametrize("case", BATCH_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}k{c[3]}" for c in BATCH_CASES])
def test_random_shapes_through_the_batch_pipelines(case):
    r = fuzz_batch.check(*case)
    assert r["same"], "pruned + two lanes differs from full transforms + one lane"
    assert fuzz_batch.passes(r), r


@pytest.mark.parametrize("case", fuzz_batch.shapes(8, 77), ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}k{c[3]}")
def test_random_shapes_through_the_level2_batch_pipelines(case):
    """The same on the kernels the 4K / 8K batches run -- level-2 passes, the fused forward transform where the shape allows,
    unmerged launches, the derived frame's row pass in one kernel (r5, csrc/dct_pair_derived.hip) -- reached at fuzz sizes
    through lowered thresholds (ssw_tuning_set; a fresh context): pruned + two lanes against full transforms + one lane bit
    for bit, frame 0 against the oracle."""
    import gpu_util as G
    from spread_spectrum_watermarking_amd import tuning
    with tuning(merge_max_lines=64, efold_min=256, efold_inv_min=256, efold_cols_min=64), G.fresh_ctx():
        r = fuzz_batch.check(*case)
    assert r["same"], "pruned + two lanes differs from full transforms + one lane"
    assert fuzz_batch.passes(r), r


def test_level2_row_passes_on_small_shapes():
    """Rows of 1280 columns or more (and columns of 720 rows or more) take the level-2 passes (csrc/ssw_pipeline.hip build_pass; 4K and 8K frames in
    test_gpu_parity.py / test_pipeline_gpu.py run them at full size, where only size-independent properties and committed
    vectors can check).  Here the thresholds are lowered in-process (ssw_tuning_set, a fresh context) so that small shapes
    the oracle transforms in seconds take the same kernels: tools/level2_check.py."""
    import level2_check
    lines = []
    bad = level2_check.run(lines.append)
    assert bad == 0, "\n".join(l for l in lines if "FAIL" in l)
    assert lines[-1] == "level-2 checks: all good"


def test_fused_forward_transform_equals_the_unfused_one():
    """r5 (csrc/dct_pair_f64_kernel.hpp, EPI_FWD_COLOP): the forward transform whose row launches write the column operands
    themselves -- no f32 plane between the passes, src/dct2d.rs:152-168 stays the rounding point -- against the unfused path
    (ssw_tuning_set fuse_cols = 0), coefficient planes bit for bit on six shapes from 144 x 256 to 8K (every tail mode of the
    row launches), first / last frames against the oracle, two batch pipelines end to end: tools/fuse_check.py."""
    import fuse_check
    lines = []
    bad = fuse_check.run(lines.append)
    assert bad == 0, "\n".join(l for l in lines if "FAIL" in l)
    assert lines[-1] == "fused forward transform: all good"


def test_light_row_prepass_is_bit_identical_to_the_register_resident_one():
    """r5: the level-2 row pre-pass in the form that fits beside the GEMMs (csrc/dct_pair_prep_light.hip: runs of pixels
    through LDS, one lane per unit, < 64 VGPRs) against pair_prep16_rows_kernel (prep_light = 0) -- whole batch pipelines from
    f32 / 8-bit / 16-bit frames (I and Q written for the writer), natural and unit line order, partial tiles of units, plain
    planes; and the inverse row pre-pass in the same style (inv_prep_light, off by default) against prep16_inv_rows_l2_kernel:
    tools/prep_light_check.py."""
    import prep_light_check
    lines = []
    bad = prep_light_check.run(lines.append, batch=[(2160, 3840, 8, 1000), (2160, 3840, 3, 500), (1080, 1920, 5, 500), (272, 512, 40, 100),
                                                   (144, 1088, 9, 64)], planes=[(272, 576, 30), (1088, 2048, 9)], derived=[])
    assert bad == 0, "\n".join(l for l in lines if "FAIL" in l)
    assert lines[-1] == "light row pre-pass: all good"


def test_fused_derived_pass_is_bit_identical_to_prepass_and_launches():
    """r5 (verdict r4 #6): the derived frame's pruned row pass in one kernel (csrc/dct_pair_derived.hip: pixels -> Y -> level-2
    fold -> MFMA against the gathered bases, no operand planes) against the pre-pass + nine gathered launches
    (derived_fused = 0): extracted marks and similarities of ssw_batch_extract{,_rgb8,_rgb16} bit for bit, marks of 64 .. 1024
    entries (class tiles of 4 .. 32 gathered columns), one to forty frames, 512 .. 7680 columns: tools/prep_light_check.py.
    Reader::extract reads the derived plane only at the base's first k indices, /root/reference/src/algorithm.rs:556-561."""
    import prep_light_check
    lines = []
    bad = prep_light_check.run(lines.append, batch=[], planes=[], inverse=[])
    assert bad == 0, "\n".join(l for l in lines if "FAIL" in l)
    assert sum("one kernel ==" in l for l in lines) == 3 * len(prep_light_check.DERIVED)


def test_gemm_ring_reproduces_the_register_staged_kernels_planes_bit_for_bit():
    """r6: the GEMM's operand tiles arrive by LDS-DMA in a ring (csrc/dct_pair_f64_kernel.hpp, SSW_GEMM_DMA) instead of through
    registers + ds_write -- same products, same k order per accumulator, so every plane is the r5 kernel's bit for bit.
    tests/golden/gemm_digests.json holds the digests of the register-staged build (-DSSW_GEMM_DMA=0) on ssw_synth_frames
    inputs: plane transforms (forward, orthonormal, inverse: src/dct2d.rs:83-219) of eight shapes covering every strategy --
    4K / 8K batches and single frames, 1080p, 720p, the 444-row dense path -- and three batch embed + extract pipelines
    (marked frames, extracted marks, similarities); tools/lib_ab_check.py computes them for the loaded library in a child process."""
    import lib_ab_check
    with open(os.path.join(ROOT, "tests", "golden", "gemm_digests.json")) as f:
        want = json.load(f)["digests"]
    got = lib_ab_check.digests()
    assert set(got) == set(want)
    diff = [k for k in want if got[k] != want[k]]
    assert not diff, f"planes differ from the register-staged kernel's: {diff}"


def test_gemm_tile_order_and_stagger_knobs_choose_between_equal_results():
    """ssw_tuning_set names of r6 (gemm_group_m, gemm_group_m_rows: tile rows per group of the block -> tile map; gemm_stagger:
    a launch-time offset for the CUs' second resident blocks; merge_batch: a batch pass's launches as one, class after class; tile48: 135 pairs as 48 + 48 + 39 or 64 + 64 + 7)
    are A/B switches of the schedule: the digests stay the same."""
    import lib_ab_check
    base = lib_ab_check.digests()
    for env in ({"SSW_GEMM_GROUP_M_ROWS": "1", "SSW_GEMM_GROUP_M": "16"}, {"SSW_GEMM_STAGGER": "2", "SSW_GEMM_GROUP_M_ROWS": "16"},
                {"SSW_MERGE_BATCH": "1", "SSW_TILE48": "0"}):
        assert lib_ab_check.digests(env=env) == base, env


def test_transform_plan_reports_the_default_path():
    """ssw_ctx_transform_plan (include/ssw.h): what a batch of a given shape runs -- the bench configuration takes the fused
    level-2 path, full HD has level-2 rows over semi-deep columns (no fusion), the reference's 640 x 444 photograph
    (tests/single_simple.rs:13) the dense kernels; fuse_cols = 0 switches the fusion off and nothing else."""
    import gpu_util as G
    from spread_spectrum_watermarking_amd import _lib as L, tuning
    c = G.ctx()
    p = c.transform_plan(128, 3840, 2160)
    assert all(p[k] for k in ("pair_f64", "rows_deep", "cols_deep", "rows_level2", "cols_level2", "class_major", "fused_cols")), p
    # one or two 4K frames run a pass's eight classes as one launch (both passes stay below merge_max_lines: 128-line tiles
    # fill the chip); three to six frames launch the classes one by one on 64-line tiles and keep the f32 plane
    assert c.transform_plan(1, 3840, 2160)["fused_cols"] and c.transform_plan(2, 3840, 2160)["fused_cols"]
    assert not c.transform_plan(3, 3840, 2160)["fused_cols"] and not c.transform_plan(5, 3840, 2160)["fused_cols"]
    assert c.transform_plan(7, 3840, 2160)["fused_cols"]
    assert not c.transform_plan(128, 3840, 2160, L.DCT3)["fused_cols"]                  # the inverse's fusion is opt-in
    p = c.transform_plan(256, 1920, 1080)
    assert p["pair_f64"] and p["rows_level2"] and p["class_major"] and not p["cols_deep"] and not p["fused_cols"], p
    assert not c.transform_plan(1, 640, 444)["pair_f64"]
    with tuning(fuse_cols=0):
        q = c.transform_plan(128, 3840, 2160)
        assert not q["fused_cols"] and q["cols_level2"] and q["rows_level2"]
    with tuning(fuse_inv_cols=1):
        assert c.transform_plan(128, 3840, 2160, L.DCT3)["fused_cols"]


def test_diagnostic_build_runs_the_strategy_matrix():
    """`make ALL_STRATEGIES=1` (lib/libssw_hip_all.so: the default library plus the r1 in-kernel folding and the f32 operand-
    ready twin).  The tests that need those strategies skip themselves under the default library; here they run in a
    child process that loads the diagnostic build through SSW_LIB_PATH -- the F32 parametrisations, folded against dense,
    every folding level in f32, pruned / fused paths of the f32 twin."""
    import subprocess
    lib = os.path.join(ROOT, "spread_spectrum_watermarking_amd", "lib", "libssw_hip_all.so")
    if not os.path.exists(lib):
        pytest.skip("diagnostic library not built (make -C spread_spectrum_watermarking_amd/csrc ALL_STRATEGIES=1)")
    env = dict(os.environ, SSW_LIB_PATH=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_pipeline_gpu.py"),
                        "-m", "gpu", "-q", "-x", "-k", "f32 or folded or strategy or pruned or fused_colour or batch_path"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


def test_tuning_table_round_trip():
    """ssw_tuning_set / get / reset (include/ssw.h): defaults, a set value, reset, unknown names."""
    from spread_spectrum_watermarking_amd import _lib as L, tuning
    lib = L.load()
    assert tuning.get("efold_min") == int(os.environ.get("SSW_EFOLD_MIN", 1280))
    with tuning(efold_min=256):
        assert tuning.get("efold_min") == 256
        with tuning(efold_min=512, fuse_cols=0):
            assert tuning.get("efold_min") == 512 and tuning.get("fuse_cols") == 0
        assert tuning.get("efold_min") == 256 and tuning.get("fuse_cols") == 1
    assert tuning.get("efold_min") == int(os.environ.get("SSW_EFOLD_MIN", 1280))
    assert lib.ssw_tuning_set(b"no_such_switch", 1) == L.SSW_ERR_BAD_ARG
    assert lib.ssw_tuning_reset(None) == L.SSW_OK
