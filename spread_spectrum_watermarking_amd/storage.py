"""Watermark storage of the reference's example CLI (examples/main.rs:110-131, :321-344).

`WatermarkStorage::Version1 { config { insert_extract { alpha, method }, ordering }, watermarks [ {
values, description } ] }` as serde_json writes it (externally tagged enum, unit variants by name),
plus the legacy `.wm` import.  Pure host plumbing: no arithmetic.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import List

import numpy as np

from .api import Extraction, Insertion, OrderingMethod, ReadConfig, WriteConfig

ORDERINGS = {"Energy": OrderingMethod.Energy, "EnergyOrthogonal": OrderingMethod.EnergyOrthogonal,
             "Legacy": OrderingMethod.Legacy}
METHODS = ("Option1", "Option2", "Option3")


def _f32_shortest(x) -> float:
    """The f32 nearest to x as the Python float with the same shortest decimal (serde_json / ryu)."""
    return float(np.format_float_positional(np.float32(x), unique=True, trim="0"))


@dataclass(frozen=True)
class Configuration:
    """examples/main.rs:110-114 (hashable: the test command caches extractions per (config, length))."""
    alpha: float = 0.1
    method: str = "Option2"
    ordering: str = "Energy"

    def __post_init__(self):
        # alpha is an f32 in the reference; keep its shortest round-trip decimal (what serde_json prints)
        object.__setattr__(self, "alpha", _f32_shortest(self.alpha))
        if self.method not in METHODS:
            raise ValueError(f"unknown method {self.method!r}")
        if self.ordering not in ORDERINGS:
            raise ValueError(f"unknown ordering {self.ordering!r}")

    def _method(self):
        return getattr(Insertion, self.method)(float(np.float32(self.alpha)))

    def to_write_config(self, precision=None) -> WriteConfig:       # main.rs:97-99, :271-274
        c = WriteConfig(insertion=self._method(), ordering=ORDERINGS[self.ordering])
        if precision is not None:
            c.precision = precision
        return c

    def to_read_config(self, precision=None) -> ReadConfig:         # main.rs:101-107, :384-387
        c = ReadConfig(extraction=self._method(), ordering=ORDERINGS[self.ordering])
        if precision is not None:
            c.precision = precision
        return c


@dataclass
class DescribedWatermark:                                           # main.rs:116-120
    values: np.ndarray
    description: str = ""


@dataclass
class Version1Storage:                                              # main.rs:122-126
    config: Configuration = field(default_factory=Configuration)
    watermarks: List[DescribedWatermark] = field(default_factory=list)

    def to_json(self) -> str:
        """serde_json::to_string_pretty(&WatermarkStorage::Version1(..)) (main.rs:299-304)."""
        doc = {"Version1": {
            "config": {"insert_extract": {"alpha": self.config.alpha, "method": self.config.method},
                       "ordering": self.config.ordering},
            "watermarks": [{"values": [_f32_shortest(v) for v in np.asarray(w.values, np.float32)],
                            "description": w.description} for w in self.watermarks]}}
        return json.dumps(doc, indent=2)

    @staticmethod
    def from_json(text: str) -> "Version1Storage":
        doc = json.loads(text)
        if set(doc) != {"Version1"}:
            raise ValueError("expected a WatermarkStorage::Version1 document")
        v = doc["Version1"]
        ie = v["config"]["insert_extract"]
        cfg = Configuration(alpha=float(ie["alpha"]), method=ie["method"], ordering=v["config"]["ordering"])
        marks = [DescribedWatermark(np.array(w["values"], np.float32), w["description"]) for w in v["watermarks"]]
        return Version1Storage(cfg, marks)

    @staticmethod
    def from_legacy(text: str) -> "Version1Storage":
        """`.wm` files of the 2013 Python code (main.rs:321-344): Option2 + Legacy ordering."""
        doc = json.loads(text)
        for key in ("alpha", "length", "version", "wm"):
            if key not in doc:
                raise ValueError(f"legacy watermark lacks {key!r}")
        cfg = Configuration(alpha=float(doc["alpha"]), method="Option2", ordering="Legacy")
        return Version1Storage(cfg, [DescribedWatermark(np.array(doc["wm"], np.float32), "")])

    @staticmethod
    def load(path: str) -> "Version1Storage":
        with open(path) as f:
            text = f.read()
        return Version1Storage.from_legacy(text) if path.endswith(".wm") else Version1Storage.from_json(text)
