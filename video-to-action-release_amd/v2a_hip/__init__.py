"""v2a_hip: MI355X-native (gfx950) kernels + host engines for the video-to-action hot path.

Importing this package loads libv2a_hip.so and raises if it is missing (no CPU fallback).
"""
from ._lib import lib, V2AError, LIB_PATH  # noqa: F401
