"""Micro-benchmark of the contraction kernels on the shapes that dominate the two workloads (run on the GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops

dev = "cuda:0"
import v2a_hip
if len(sys.argv) > 1:
    v2a_hip.set_precision(sys.argv[1])
print("precision:", v2a_hip.get_precision())
FWD = [  # name, N, H, W, Cin, Cout, k, stride
    ("video 128^2 128->128 3x3 (B16)", 112, 128, 128, 128, 128, 3, 1),
    ("video 64^2 256->256 3x3", 112, 64, 64, 256, 256, 3, 1),
    ("video 32^2 384->384 3x3", 112, 32, 32, 384, 384, 3, 1),
    ("video 16^2 512->512 3x3", 112, 16, 16, 512, 512, 3, 1),
    ("video 8^2 1280->640 3x3", 112, 8, 8, 1280, 640, 3, 1),
    ("video temporal 128ch (3x1)", 16, 7, 16384, 128, 128, (3, 1), 1),
    ("resnet l1 64->64 32^2 (B64)", 64, 32, 32, 64, 64, 3, 1),
    ("resnet l2 128->128 16^2", 64, 16, 16, 128, 128, 3, 1),
    ("resnet l3 256->256 8^2", 64, 8, 8, 256, 256, 3, 1),
    ("resnet l4 512->512 4^2", 64, 4, 4, 512, 512, 3, 1),
    ("unet1d 1024->1024 k5 T4", 64, 1, 4, 1024, 1024, (1, 5), 1),
    ("unet1d 512->512 k5 T8", 64, 1, 8, 512, 512, (1, 5), 1),
    ("unet1d 256->256 k5 T16", 64, 1, 16, 256, 256, (1, 5), 1),
    ("film linear 256->2048 (B64)", 1, 1, 64, 256, 2048, 1, 1),
]


def timeit(fn, iters=20):
    """GPU time per call under hipGraph replay (no host launch overhead between the kernels)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
  for name, N, H, W, Ci, Co, k, s in FWD:
      kh, kw = (k, k) if isinstance(k, int) else k
      x = torch.randn(N, H, W, Ci, device=dev)
      w = torch.randn(Co, kh * kw * Ci, device=dev) * 0.02
      b = torch.randn(Co, device=dev)
      y = ops.conv2d(x, w, b, Co, kh, kw, (s, s), (kh // 2, kw // 2))
      M = y.shape[0] * y.shape[1] * y.shape[2]
      fl = 2.0 * M * Co * kh * kw * Ci
      t = timeit(lambda: ops.conv2d(x, w, b, Co, kh, kw, (s, s), (kh // 2, kw // 2)))
      dy = torch.randn_like(y)
      tw = timeit(lambda: ops.conv2d_wgrad(x, dy, (Co, Ci, kh, kw), kh, kw, (s, s), (kh // 2, kw // 2)))
      print(f"{name:36s} M={M:8d} K={kh*kw*Ci:6d} N={Co:5d}  fwd {t*1e6:9.1f} us {fl/t/1e12:7.1f} TF | wgrad {tw*1e6:9.1f} us {fl/tw/1e12:7.1f} TF")


if __name__ == "__main__":
    main()
