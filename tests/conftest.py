import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _all_strategies():
    """Is the loaded library the diagnostic build (make ALL_STRATEGIES=1, SSW_LIB_PATH=.../libssw_hip_all.so)?  The default
    library carries the f64 pair path and the dense kernels only; the strategy-matrix tests (in-kernel folding, the f32
    operand-ready twin) parametrise over what is loaded."""
    try:
        from spread_spectrum_watermarking_amd import _lib as L
        return L.all_strategies()
    except Exception:
        return False


ALL_STRATEGIES = _all_strategies()
needs_all_strategies = pytest.mark.skipif(not ALL_STRATEGIES, reason="needs the diagnostic build (make ALL_STRATEGIES=1; SSW_LIB_PATH)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def marks():
    return dict(np.load(os.path.join(GOLDEN, "marks.npz")))


@pytest.fixture(scope="session")
def cat_images():
    g = np.load(os.path.join(GOLDEN, "cat_decoded_u8.npz"))
    return {"cat": g["cat"], "watermarked_with_1": g["watermarked_with_1"]}


def u8_to_f32(img_u8):
    """image 0.24 `into_rgb32f` for 8-bit input: v / 255 in f32 (unpinned)."""
    return img_u8.astype(np.float32) / np.float32(255)


def f32_to_u8(img_f32):
    """image 0.24 `into_rgb8` from Rgb32F: round(clamp(v,0,1)*255), half away from zero (unpinned)."""
    v = np.clip(img_f32.astype(np.float32), 0, 1) * np.float32(255)
    return np.floor(v + np.float32(0.5)).astype(np.uint8)
