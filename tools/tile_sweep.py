"""Which forward tile is fastest for each policy-shaped conv? (graph-replay timing)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from v2a_hip import ops
from v2a_hip._lib import lib
from conv_bench import timeit
dev = "cuda:0"
SH = [("resnet l1 64->64 32^2", 64, 32, 32, 64, 64, 3, 1), ("resnet l2 128 16^2", 64, 16, 16, 128, 128, 3, 1), ("resnet l2.0 s2 64->128", 64, 32, 32, 64, 128, 3, 2),
      ("resnet l3 256 8^2", 64, 8, 8, 256, 256, 3, 1), ("resnet l4 512 4^2", 64, 4, 4, 512, 512, 3, 1),
      ("unet1d 1024 k5 T4", 64, 1, 4, 1024, 1024, (1, 5), 1), ("unet1d 512 k5 T8", 64, 1, 8, 512, 512, (1, 5), 1),
      ("unet1d 256 k5 T16", 64, 1, 16, 256, 256, (1, 5), 1), ("unet1d 2048->512 k5 T4", 64, 1, 4, 2048, 512, (1, 5), 1)]
for name, N, H, W, Ci, Co, k, s in SH:
    kh, kw = (k, k) if isinstance(k, int) else k
    x = torch.randn(N, H, W, Ci, device=dev)
    w = torch.randn(Co, kh * kw * Ci, device=dev) * 0.02
    res = []
    for bm, bn in [(0, 0), (128, 128), (128, 64), (64, 64)]:
        lib.v2a_debug_force_tile(bm, bn)
        y = ops.conv2d(x, w, None, Co, kh, kw, (s, s), (kh // 2, kw // 2))
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Co * kh * kw * Ci
        t = timeit(lambda: ops.conv2d(x, w, None, Co, kh, kw, (s, s), (kh // 2, kw // 2)))
        res.append(f"{bm}x{bn}: {t*1e6:6.1f}us {fl/t/1e12:5.1f}TF")
    lib.v2a_debug_force_tile(0, 0)
    print(f"{name:26s} " + " | ".join(res))
