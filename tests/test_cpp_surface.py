"""Builds and runs the C++ host-side mirror (include/ssw.hpp) against libssw_hip.so.

CPU part: the wrapper header and the test program compile and link (no GPU needed).
GPU part: the doc-test flow of the reference (lib.rs:24-66) through the C++ types, checked
against the CPU oracle on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "crate_surface_test.cpp")
LIBDIR = os.path.join(ROOT, "spread_spectrum_watermarking_amd", "lib")


def build(tmpdir):
    exe = os.path.join(str(tmpdir), "crate_surface_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                    "-L", LIBDIR, "-lssw_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_class_major_layout_maps_on_the_host(tmp_path):
    """csrc/dct_pair_common.hpp's column orders of the intermediate planes (deep transforms): bijections, inverses of each
    other, and consistent with the output maps of the GEMM launch classes -- compiled for the host, no device code runs."""
    exe = os.path.join(str(tmp_path), "class_layout_test")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "class_layout_test.cpp"),
                    "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip() == "ok"


def test_cpp_wrappers_compile_and_link(tmp_path):
    exe = build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_doc_test_flow_matches_oracle(tmp_path):
    from oracle import oracle as O
    exe = build(tmp_path)
    w, h, k = 320, 180, 200
    rgb = O.synth_frame(5, 0, w, h)
    mark = np.random.default_rng(5).standard_normal(k).astype(np.float32)
    p = lambda n: os.path.join(str(tmp_path), n)
    rgb.tofile(p("rgb.f32")); mark.tofile(p("mark.f32"))
    out = subprocess.run([exe, p("rgb.f32"), str(w), str(h), p("mark.f32"), str(k), p("marked.f32"), p("ext.f32"), p("marked8.u8")],
                         check=True, capture_output=True, text=True).stdout.split()
    vals = dict(zip(out[::2], out[1::2]))
    marked = np.fromfile(p("marked.f32"), np.float32).reshape(h, w, 3)
    ext = np.fromfile(p("ext.f32"), np.float32)
    ref_marked = O.embed_frame(rgb, mark)
    ref_ext, ref_sim = O.extract_frame(rgb, ref_marked, mark)
    assert np.abs(marked - ref_marked).max() <= 2e-7       # default = canonical precision
    assert np.abs(ext - ref_ext).max() <= 1e-5 * np.maximum(1.0, np.abs(ref_ext)).max()
    assert abs(float(vals["similarity"]) - ref_sim) < 1e-4
    assert vals["exceeds6"] == "1" and vals["consumed_ok"] == "1" and vals["too_large_ok"] == "1"
    assert abs(float(vals["random"])) < 5.0
    coef = O.dct2d(O.rgb_to_yiq(rgb)[0])
    assert int(vals["first_index"]) == int(O.indices(coef, k=1)[0])
    # the 8-bit leg (ImageRgb8 in, mark_rgb8 out): same bytes through the oracle
    img8 = O.f32_to_u8(rgb)
    marked8 = np.fromfile(p("marked8.u8"), np.uint8).reshape(h, w, 3)
    o_marked8 = O.f32_to_u8(O.embed_frame(O.u8_to_f32(img8), mark))
    assert np.mean(marked8 == o_marked8) >= 0.9999
    _, o_sim8 = O.extract_frame(O.u8_to_f32(img8), O.u8_to_f32(marked8), mark)
    assert abs(float(vals["similarity8"]) - o_sim8) < 1e-4 * max(1.0, abs(o_sim8))
