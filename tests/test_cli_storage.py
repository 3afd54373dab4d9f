"""the example CLI's storage format and flows (examples/main.rs) -- host plumbing."""
import io
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN
from spread_spectrum_watermarking_amd import cli
from spread_spectrum_watermarking_amd.storage import Configuration, DescribedWatermark, Version1Storage


def test_version1_json_layout_and_round_trip():
    st = Version1Storage(Configuration(0.1, "Option2", "Energy"),
                         [DescribedWatermark(np.array([0.5, -1.25, 3.0], np.float32), 'say "hi"')])
    doc = json.loads(st.to_json())
    # serde: externally tagged enum, unit variants by name (main.rs:110-131)
    assert list(doc) == ["Version1"]
    assert doc["Version1"]["config"] == {"insert_extract": {"alpha": 0.1, "method": "Option2"},
                                         "ordering": "Energy"}
    assert doc["Version1"]["watermarks"] == [{"values": [0.5, -1.25, 3.0], "description": 'say "hi"'}]
    back = Version1Storage.from_json(st.to_json())
    assert back.config == st.config and np.array_equal(back.watermarks[0].values, st.watermarks[0].values)
    assert back.watermarks[0].description == 'say "hi"'
    assert hash(back.config) == hash(st.config)                   # cache key of the test command


def test_legacy_wm_import(tmp_path):
    legacy = {"alpha": 0.05, "length": 3, "version": "1", "wm": [1.0, -2.0, 0.5]}
    p = tmp_path / "old.wm"
    p.write_text(json.dumps(legacy))
    st = Version1Storage.load(str(p))                             # main.rs:321-344
    assert st.config == Configuration(0.05, "Option2", "Legacy")
    assert np.array_equal(st.watermarks[0].values, np.array([1.0, -2.0, 0.5], np.float32))
    with pytest.raises(ValueError):
        Version1Storage.from_json(json.dumps({"Version2": {}}))
    with pytest.raises(ValueError):
        Configuration(0.1, "Option9", "Energy")


def test_argument_surface_and_output_names():
    p = cli.build_parser()
    a = p.parse_args(["watermark", "/tmp/foo.jpg"])
    assert (a.length, a.ordering, a.alpha, a.method, a.description, a.print_similarity) == (1000, "energy", 0.1, "option2", None, False)
    a = p.parse_args(["watermark", "x.png", "--length", "500", "--ordering", "energy-orthogonal", "--alpha", "0.2",
                      "--method", "option3", "-d", "mine", "-p"])
    assert (a.length, a.ordering, a.alpha, a.method, a.description, a.print_similarity) == (500, "energy-orthogonal", 0.2, "option3", "mine", True)
    t = p.parse_args(["test", "a.jpg", "b.png", "w1.json", "w2.wm"])
    assert t.similarity_exceed == 6.0 and t.watermark_files == ["w1.json", "w2.wm"]
    assert cli.out_paths("/tmp/foo.jpg") == ("/tmp/foo_wm.png", "/tmp/foo_wm.json")   # main.rs:245-251
    assert cli._rust_f32(6.0) == "6" and cli._rust_f32(31.886204) == "31.886204"


@pytest.mark.gpu
def test_watermark_then_test_commands(tmp_path):
    """`watermark <file>` then `test <base> <wm> <json> <other json>`: the embedded mark matches, a foreign one does not."""
    src = str(tmp_path / "cat.jpg")
    shutil.copy(os.path.join(GOLDEN, "porcelain_cat_grey_background.jpg"), src)
    assert cli.main(["watermark", src, "-d", 'the "cat"']) == 0
    png, js = cli.out_paths(src)
    assert os.path.exists(png) and os.path.exists(js)
    with pytest.raises(SystemExit):                               # refuses to overwrite (main.rs:253-265)
        cli.main(["watermark", src])
    st = Version1Storage.load(js)
    assert st.config == Configuration(0.1, "Option2", "Energy") and len(st.watermarks[0].values) == 1000
    other = tmp_path / "other.json"
    other.write_text(Version1Storage(st.config, [DescribedWatermark(
        np.random.default_rng(0).standard_normal(1000).astype(np.float32), "foreign")]).to_json())
    buf = io.StringIO()
    args = cli.build_parser().parse_args(["test", src, png, js, str(other)])
    assert cli.cmd_test(args, out=buf) == 0
    lines = buf.getvalue().splitlines()
    assert lines[0] == "-" and lines[1] == "  Matches: true" and lines[3] == "  MatchExceed: 6"
    assert lines[4] == '  Description: "the \\"cat\\""' and lines[5] == f'  File: "{js}"'
    assert float(lines[2].split(": ")[1]) > 25.0
    assert lines[7] == "  Matches: false" and abs(float(lines[8].split(": ")[1])) < 4.0
