"""Sweep (tile, split) of the fp32 weight-gradient kernel per shape (GPU box); prints the best plan next to the heuristic's.
Shapes: the video-training and policy-step weight gradients.   python tools/wgrad_sweep.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops
from v2a_hip._lib import lib

dev = "cuda:0"
if "--bf16" in sys.argv:          # bf16 MFMA weight-gradient kernels (fp32 operands in HBM, converted while staging)
    lib.v2a_set_precision(1)
# (N, H, W, Cin, kh, kw, Cout), stride 1, "same" padding
SHAPES = [(14, 128, 128, 128, 3, 3, 128), (14, 64, 64, 256, 3, 3, 256), (14, 32, 32, 384, 3, 3, 384), (14, 16, 16, 512, 3, 3, 512),
          (14, 8, 8, 640, 3, 3, 640), (2, 7, 16384, 128, 3, 1, 128), (2, 7, 4096, 256, 3, 1, 256), (2, 7, 1024, 384, 3, 1, 384),
          (2, 7, 256, 512, 3, 1, 512), (2, 7, 64, 640, 3, 1, 640), (14, 64, 64, 640, 3, 3, 256), (14, 32, 32, 896, 3, 3, 384),
          (14, 128, 128, 256, 3, 3, 128), (14, 16, 16, 512, 1, 1, 1536),
          # policy step (B=64): ResNet18 stages and the 1-D UNet
          (64, 32, 32, 64, 3, 3, 64), (64, 16, 16, 128, 3, 3, 128), (64, 8, 8, 256, 3, 3, 256), (64, 4, 4, 512, 3, 3, 512),
          (64, 1, 16, 256, 1, 5, 256), (64, 1, 8, 512, 1, 5, 512), (64, 1, 4, 1024, 1, 5, 1024)]


def timeit(f, n=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for N, H, W, Ci, kh, kw, Co in SHAPES:
    M, K = N * H * W, Ci * kh * kw
    x = torch.randn(N, H, W, Ci, device=dev)
    dy = torch.randn(N, H, W, Co, device=dev)
    f = lambda: ops.conv2d_wgrad(x, dy, (Co, Ci, kh, kw), kh, kw, (1, 1), (kh // 2, kw // 2))
    lib.v2a_debug_force_wgrad_plan(0, 0, 0)
    t0 = timeit(f)
    k0 = ops.last_kernel[0]
    fl = 2.0 * M * K * Co
    res = []
    for bm, bn in ((128, 128), (128, 64), (64, 64)):
        if bm == 128 and Co <= 64:
            continue
        tiles = -(-Co // bm) * -(-K // bn)
        cands = sorted({1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 170, 256} | {max(1, 256 // tiles), max(1, 512 // tiles), max(1, 768 // tiles),
                                                                                           max(1, 1024 // tiles)})
        for s in cands:
            if s > max(1, (M // 32) // 4) or tiles * s > 4096:
                continue
            lib.v2a_debug_force_wgrad_plan(bm, bn, s)
            try:
                t = timeit(f, 5)
            except Exception as e:
                continue
            res.append((t, bm, bn, s, tiles * s))
    lib.v2a_debug_force_wgrad_plan(0, 0, 0)
    res.sort()
    best = res[0]
    top = "  ".join(f"{bm}x{bn}/s{s}(wg{w}):{t*1e6:.0f}" for t, bm, bn, s, w in res[:5])
    print(f"M={M:7d} K={K:5d} Co={Co:5d}  heuristic {k0:30s} {t0*1e6:8.1f} us {fl/t0/1e12:6.1f} TF | best {fl/best[0]/1e12:6.1f} TF | {top}", flush=True)
