"""atracdenc_amd: MI355X-native ATRAC3 encode hot path.

Python is only the test/bench harness around the C-ABI library (include/at3hip.h); the product is
libat3hip.so (hand-written HIP for gfx950, atracdenc_amd/csrc). There is no CPU fallback: importing
works anywhere, but creating an encoder raises if the library or a GPU is missing.
"""
from .binding import At1Hip, At3Hip, At3pHip, At3HipError, LIB_PATH, build_library, load_library  # noqa: F401

__all__ = ["At1Hip", "At3Hip", "At3pHip", "At3HipError", "LIB_PATH", "build_library", "load_library"]
