"""v2a_hip: MI355X-native (gfx950) kernels + host engines for the video-to-action hot path.

Importing this package loads libv2a_hip.so and raises if it is missing (no CPU fallback).
"""
from ._lib import lib, V2AError, LIB_PATH  # noqa: F401


def set_precision(mode: str) -> str:
    """'fp32' (exact-f32 MFMA: the parity configuration, default) or 'bf16' (bf16 MFMA inputs, fp32 accumulate / storage)."""
    code = {"fp32": 0, "f32": 0, "bf16": 1}[mode]
    old = lib.v2a_set_precision(code)
    return "bf16" if old == 1 else "fp32"


def get_precision() -> str:
    return "bf16" if lib.v2a_get_precision() == 1 else "fp32"
