"""A/B: data gradient through the N-major loader (forward pack, bmode=1) vs the K-contiguous flipped pack (bmode=0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from v2a_hip import ops
from conv_bench import timeit
dev = "cuda:0"
for name, N, H, W, Ci, Co, k in [("resnet l1 64->64 32^2", 64, 32, 32, 64, 64, 3), ("resnet l2 128 16^2", 64, 16, 16, 128, 128, 3),
                                 ("resnet l3 256 8^2", 64, 8, 8, 256, 256, 3), ("resnet l4 512 4^2", 64, 4, 4, 512, 512, 3),
                                 ("unet1d 1024 k5 T4", 64, 1, 4, 1024, 1024, (1, 5)), ("unet1d 512 k5 T8", 64, 1, 8, 512, 512, (1, 5)),
                                 ("unet1d 256 k5 T16", 64, 1, 16, 256, 256, (1, 5))]:
    kh, kw = (k, k) if isinstance(k, int) else k
    w = torch.randn(Co, Ci, kh, kw, device=dev) * 0.02
    dy = torch.randn(N, H, W, Co, device=dev)
    pf, pd = ops.pack_weight(w, 0), ops.pack_weight(w, 1)
    fl = 2.0 * N * H * W * Co * Ci * kh * kw
    t1 = timeit(lambda: ops.conv2d(dy, pf, None, Ci, kh, kw, (1, 1), (kh // 2, kw // 2), bmode=1))
    t0 = timeit(lambda: ops.conv2d(dy, pd, None, Ci, kh, kw, (1, 1), (kh // 2, kw // 2), bmode=0))
    print(f"{name:26s} nmaj {t1*1e6:7.1f} us {fl/t1/1e12:6.1f} TF | flipped pack {t0*1e6:7.1f} us {fl/t0/1e12:6.1f} TF")
