"""fp32 conv by three bf16 planes (six bf16 MFMAs per product block) against the exact-f32 MFMA kernels: accuracy against an fp64
reference on the same inputs, and time per launch.  Usage (GPU box): python tools/probes/emu_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from v2a_hip import ops
from v2a_hip._lib import lib

dev = "cuda:0"
SHAPES = [
    ("res1 3x3 64->64   32x32", 64, 32, 32, 64, 64, 3),
    ("res2 3x3 128->128 16x16", 64, 16, 16, 128, 128, 3),
    ("res3 3x3 256->256  8x8", 64, 8, 8, 256, 256, 3),
    ("res4 3x3 512->512  4x4", 64, 4, 4, 512, 512, 3),
    ("vid 3x3 256->256 64x64 (B=2x7)", 14, 64, 64, 256, 256, 3),
    ("vid 3x3 128->128 128x128 (B=2x7)", 14, 128, 128, 128, 128, 3),
    ("unet1d k5 512->512 T8 (1x5)", 64, 1, 8, 512, 512, (1, 5)),
]


def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]


for name, N, H, W, Ci, Co, k in SHAPES:
    torch.manual_seed(0)
    kh, kw = (k, k) if isinstance(k, int) else k
    x = torch.randn(N, H, W, Ci, device=dev) * torch.rand(N, H, W, Ci, device=dev).mul(6).exp2()     # wide dynamic range
    w = torch.randn(Co, kh, kw, Ci, device=dev) * 0.05
    wp = w.reshape(Co, -1).contiguous()
    pad = (kh // 2, kw // 2)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=pad).permute(0, 2, 3, 1).contiguous()
    absref = F.conv2d(x.permute(0, 3, 1, 2).double().abs(), w.permute(0, 3, 1, 2).double().abs(), padding=pad).permute(0, 2, 3, 1)
    M = N * H * W
    fl = 2.0 * M * Co * kh * kw * Ci
    out = {}
    f = lambda: ops.conv2d(x, wp, None, Co, kh, kw, (1, 1), pad)
    for mode, (prec, dma, x3) in {"exact-f32 (LDS-DMA, pipelined)": (0, 0, 0), "exact-f32 (register-staged)": (0, 1 << 60, 0),
                                  "bf16x3 planes (naive probe kernel)": (2, 0, 0), "bf16x3 planes (conv_igemm_f32x3)": (0, 0, 1),
                                  "bf16 single plane": (1, 0, 0)}.items():
        lib.v2a_set_precision(prec)
        lib.v2a_set_f32_conv_mode(x3)
        ops._DMA_F32_MIN_WORK[0] = dma if dma else 300000
        y = f()
        t = timeit(f)
        err = (y.double() - ref).abs()
        # error relative to sum |a||b| (the natural scale of a dot product's rounding error) and to the output's RMS
        e_abs = (err / absref).max().item()
        e_rms = (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        print(f"{name:32s} {mode:34s} {t:8.1f} us {fl / t / 1e6:7.1f} TF | max err / sum|a||b| {e_abs:.2e}   rms err / rms {e_rms:.2e}", flush=True)
    lib.v2a_set_precision(0)
    lib.v2a_set_f32_conv_mode(1)
    ops._DMA_F32_MIN_WORK[0] = 300000
