"""Command line front-end mirroring the reference's example binary (examples/main.rs).

    python -m spread_spectrum_watermarking_amd.cli watermark <file> [--length 1000] [--ordering energy]
            [--alpha 0.1] [--method option2] [-d DESCRIPTION] [-p]
        -> <stem>_wm.png and <stem>_wm.json next to <file>            (main.rs:240-319)
    python -m spread_spectrum_watermarking_amd.cli test [--similarity-exceed 6.0] <base> <watermarked> <json|wm>...
        -> one YAML-ish record per stored watermark                     (main.rs:346-434)

Host plumbing only (argument parsing, PIL image I/O, JSON); all arithmetic goes through the GPU
library via the crate-surface mirror in api.py.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Dict, List, Optional, Tuple

import numpy as np

from .api import MarkBuf, Reader, Tester, Writer
from .storage import Configuration, DescribedWatermark, Version1Storage

_ORDERING_ARGS = {"energy": "Energy", "energy-orthogonal": "EnergyOrthogonal", "legacy": "Legacy"}
_METHOD_ARGS = {"option1": "Option1", "option2": "Option2", "option3": "Option3"}


def _rust_f32(x: float) -> str:
    """`{}` of an f32 in Rust: shortest round-trip digits, no trailing `.0`."""
    s = np.format_float_positional(np.float32(x), unique=True, trim="-")
    return s


def _open_image(path: str) -> np.ndarray:
    from PIL import Image
    try:
        return np.asarray(Image.open(path).convert("RGB"))
    except Exception as e:                       # the reference panics with this message
        raise SystemExit(f"Could not load image at {path!r}") from e


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="spread_spectrum_watermarking_amd.cli")
    sub = p.add_subparsers(dest="command")
    w = sub.add_parser("watermark", help="Embed a watermark into a file.")
    w.add_argument("file", help="The file to to watermark.")
    w.add_argument("--length", type=int, default=1000, help="Watermark length.")
    w.add_argument("--ordering", choices=sorted(_ORDERING_ARGS), default="energy", help="The ordering to be used.")
    w.add_argument("--alpha", type=float, default=0.1, help="Strength, alpha in the equations.")
    w.add_argument("--method", choices=sorted(_METHOD_ARGS), default="option2", help="Method to insert and extract with.")
    w.add_argument("-d", "--description", default=None, help="Description to associate with the watermark.")
    w.add_argument("-p", dest="print_similarity", action="store_true", help="Show embedded watermark similarity.")
    t = sub.add_parser("test", help="Test if any of the watermarks are present in the watermarked file.")
    t.add_argument("--similarity-exceed", type=float, default=6.0,
                   help="If the similarity exceeds this value it is considered to be matching.")
    t.add_argument("base", help="The original file.")
    t.add_argument("watermarked", help="The derived (watermarked) file.")
    t.add_argument("watermark_files", nargs="+", help="The watermark files to test from.")
    return p


def out_paths(image_path: str) -> Tuple[str, str]:
    """/tmp/foo.jpg -> /tmp/foo_wm.png, /tmp/foo_wm.json (main.rs:245-251)."""
    stem = os.path.splitext(image_path)[0] + "_wm"
    return stem + ".png", stem + ".json"


def cmd_watermark(args, out=sys.stdout) -> int:
    orig = _open_image(args.file)
    image_out, json_out = out_paths(args.file)
    for path in (image_out, json_out):                       # main.rs:253-265
        if os.path.exists(path):
            raise SystemExit(f"{path} file already exists")
    cfg = Configuration(alpha=args.alpha, method=_METHOD_ARGS[args.method], ordering=_ORDERING_ARGS[args.ordering])
    mark = MarkBuf.generate_normal(args.length)              # main.rs:269
    # main.rs:271-278: Writer::new(orig).mark(&[&mark]).into_rgb8() -- 8-bit in, 8-bit out, quantised on the device
    derived8 = Writer(orig, cfg.to_write_config()).mark_rgb8([mark])
    from PIL import Image
    Image.fromarray(derived8).save(image_out)
    storage = Version1Storage(cfg, [DescribedWatermark(mark.data(), args.description or "")])
    with open(json_out, "w") as f:
        f.write(storage.to_json())
    if args.print_similarity:                                # main.rs:306-316 (default ReadConfig, like the reference)
        from .api import ReadConfig
        ext = Reader.base(orig, ReadConfig.default()).extract(Reader.derived(derived8), args.length)
        sim = Tester(ext).similarity(mark)
        print(f"sim: Similarity {{ similarity: {_rust_f32(sim.similarity)} }}", file=out)
        print(f"exceeds 6 sigma: {'true' if sim.exceeds_sigma(6.0) else 'false'}", file=out)
    return 0


def cmd_test(args, out=sys.stdout) -> int:
    base = _open_image(args.base)
    watermarked = _open_image(args.watermarked)
    stored: List[Tuple[str, Version1Storage]] = [(p, Version1Storage.load(p)) for p in args.watermark_files]
    retrieved: Dict[Tuple[Configuration, int], np.ndarray] = {}          # main.rs:369-371
    for path, info in stored:
        for wmk in info.watermarks:
            key = (info.config, len(wmk.values))
            if key not in retrieved:                                     # main.rs:383-407
                reader = Reader.base(base, info.config.to_read_config())
                retrieved[key] = reader.extract(Reader.derived(watermarked), key[1])
            sim = Tester(retrieved[key]).similarity(wmk.values)
            desc = wmk.description.replace('"', '\\"')
            print("-", file=out)                                          # main.rs:418-429
            print(f"  Matches: {'true' if sim.exceeds_sigma(args.similarity_exceed) else 'false'}", file=out)
            print(f"  Similarity: {_rust_f32(sim.similarity)}", file=out)
            print(f"  MatchExceed: {_rust_f32(args.similarity_exceed)}", file=out)
            print(f"  Description: \"{desc}\"", file=out)
            print(f"  File: \"{path}\"", file=out)
    return 0


def main(argv: Optional[List[str]] = None) -> int:
    args = build_parser().parse_args(argv)
    if args.command == "watermark":
        return cmd_watermark(args)
    if args.command == "test":
        return cmd_test(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
