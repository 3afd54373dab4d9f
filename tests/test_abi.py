"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/ssw.h declares, and refuses to run without a GPU (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from spread_spectrum_watermarking_amd import _lib as L
import spread_spectrum_watermarking_amd as wm


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ssw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssw_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 35
    lib = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ssw.h but not exported by libssw_hip.so"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(L.SIGNATURES) == names


def test_header_cites_the_reference_for_every_path_entry_point():
    text = open(os.path.join(ROOT, "include", "ssw.h")).read()
    for name in ("ssw_writer_create", "ssw_writer_mark", "ssw_reader_create", "ssw_reader_extract",
                 "ssw_similarity", "ssw_dct2d", "ssw_rgb_to_yiq", "ssw_topk_indices"):
        decl = text.index(name + "(")
        comment = text.rfind("/*", 0, decl)
        assert re.search(r"(algorithm|dct2d|yiq)\.rs:\d+", text[comment:decl]), name


def test_status_strings_and_default_config():
    lib = L.load()
    assert lib.ssw_status_string(0) == b"ok"
    assert b"exceeds available coefficients" in lib.ssw_status_string(L.SSW_ERR_K_TOO_LARGE)
    cfg = L.Config()
    lib.ssw_config_default(C.byref(cfg))
    assert (cfg.ordering, cfg.method, cfg.precision) == (L.ORDER_ENERGY, L.OPTION2, L.PRECISION_F64)
    assert abs(cfg.alpha - 0.1) < 1e-7
    d = wm.WriteConfig.default()._c()
    assert (d.ordering, d.method, d.precision) == (cfg.ordering, cfg.method, cfg.precision)


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(wm.SswError) as e:
        wm.Context(0)
    assert e.value.status == L.SSW_ERR_NO_DEVICE
    import numpy as np
    with pytest.raises(wm.SswError):
        wm.Writer(np.zeros((4, 4, 3), np.float32))


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "spread_spectrum_watermarking_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libssw_oracle" not in src and "sswo_" not in src and "import oracle" not in src, f
                assert "from oracle" not in src, f


def test_tuning_table_needs_no_gpu():
    """ssw_tuning_set / get / reset (include/ssw.h; csrc/tuning.hip): the process-wide table of strategy thresholds is host
    state -- defaults, nested overrides through the Python context manager, unknown names."""
    from spread_spectrum_watermarking_amd import _lib as L, tuning
    lib = L.load()
    base = tuning.get("efold_cols_min")
    with tuning(efold_cols_min=64):
        assert tuning.get("efold_cols_min") == 64
        with tuning(efold_cols_min=128, fuse_inv_cols=1):
            assert tuning.get("efold_cols_min") == 128 and tuning.get("fuse_inv_cols") == 1
        assert tuning.get("efold_cols_min") == 64 and tuning.get("fuse_inv_cols") == 0
    assert tuning.get("efold_cols_min") == base
    assert lib.ssw_tuning_set(b"no_such_switch", 1) == L.SSW_ERR_BAD_ARG
    assert lib.ssw_tuning_reset(None) == L.SSW_OK
    assert lib.ssw_build_all_strategies() == 0          # the default library is what build() puts at lib/libssw_hip.so
