"""MI355X-native spread-spectrum watermarking hot path (gfx950 HIP kernels behind a C ABI).

Public surface mirrors the reference crate's re-exports (lib.rs:81-85).
"""
from ._lib import SswError, SswLibraryMissing, LIB_PATH  # noqa: F401
from .api import (Context, DeviceBuffer, Extraction, Insertion, MarkBuf, OrderingMethod, Precision,  # noqa: F401
                  ReadConfig, Reader, ReaderDerived, Similarity, Tester, WriteConfig, Writer,
                  default_context, extract_many, mark_many, tuning)

__all__ = ["MarkBuf", "Tester", "Extraction", "Insertion", "OrderingMethod", "ReadConfig", "Reader",
           "ReaderDerived", "WriteConfig", "Writer", "Similarity", "Context", "Precision", "SswError"]
