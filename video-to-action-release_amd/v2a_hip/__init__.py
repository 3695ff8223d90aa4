"""v2a_hip: MI355X-native (gfx950) kernels + host engines for the video-to-action hot path.

Importing this package loads libv2a_hip.so and raises if it is missing (no CPU fallback).
"""
from ._lib import lib, V2AError, LIB_PATH  # noqa: F401


def set_precision(mode: str) -> str:
    """'fp32' (fp32 arithmetic: the parity configuration, default), 'bf16' or 'fp16' (16-bit MFMA inputs, fp32 accumulate / storage).
    'fp16' = IEEE half inputs, the reference's own GPU precision (accelerate mixed_precision='fp16': lb_online_trainer_v7.py:72-76);
    PolicyTrainer then runs its fused optimiser with dynamic loss scaling (GradScaler's contract)."""
    code = {"fp32": 0, "f32": 0, "bf16": 1, "fp16": 1}[mode]
    old = get_precision()
    lib.v2a_set_precision(code)
    if code == 1:
        set_policy_half(mode)
    return old


def get_precision() -> str:
    if lib.v2a_get_precision() != 1:
        return "fp32"
    return "fp16" if lib.v2a_get_policy_half() == 1 else "bf16"


def set_policy_half(mode: str) -> str:
    """16-bit format of the policy's MFMA mode: 'bf16' (default) or 'fp16'."""
    import torch
    from . import ops
    f16 = {"bf16": 0, "fp16": 1}[mode]
    old = lib.v2a_set_policy_half(f16)
    ops.POLICY_HALF[0] = torch.float16 if f16 else torch.bfloat16
    return "fp16" if old == 1 else "bf16"


_video_storage = ["f32"]


def set_video_storage(mode: str) -> str:
    """HBM storage type of the video UNet's activations / weight packs: 'f32' (parity configuration, default) or 'bf16' (bf16
    tensors, fp32 accumulation and statistics: the counterpart of the reference's fp16-autocast GPU path).  Applies to every
    video UNet whose channel widths allow it (multiples of 64) and that has no `.storage` attribute of its own."""
    if mode not in ("f32", "bf16", "fp16"):      # "fp16": IEEE half instances of the same kernels (the reference's own 16-bit type)
        raise ValueError(mode)
    old = _video_storage[0]
    _video_storage[0] = mode
    return old


def get_video_storage() -> str:
    return _video_storage[0]


_sampler_graphs = [True]


def sampler_graphs_enabled() -> bool:
    """GoalGaussianDiffusion.sample replays one captured hipGraph per denoise step (set_sampler_graphs(False): eager launches)."""
    return _sampler_graphs[0]


def set_sampler_graphs(on: bool) -> bool:
    old = _sampler_graphs[0]
    _sampler_graphs[0] = bool(on)
    return old
