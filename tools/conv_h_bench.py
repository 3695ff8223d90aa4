"""Micro-benchmark of the bf16-storage conv kernel on the video UNet's dominant shapes (run on the GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops
from tools.conv_bench import timeit

dev = "cuda:0"
SHAPES = [  # name, N, H, W, C1, C2, Cout, k, stride
    ("128^2 128->128 3x3 (B16x7)", 112, 128, 128, 128, 0, 128, 3, 1),
    ("64^2 256->256 3x3", 112, 64, 64, 256, 0, 256, 3, 1),
    ("64^2 (256+128)->256 3x3 concat", 112, 64, 64, 256, 128, 256, 3, 1),
    ("32^2 384->384 3x3", 112, 32, 32, 384, 0, 384, 3, 1),
    ("16^2 512->512 3x3", 112, 16, 16, 512, 0, 512, 3, 1),
    ("8^2 1280->640 3x3", 112, 8, 8, 1280, 0, 640, 3, 1),
    ("temporal 128ch (3x1) 128^2", 16, 7, 16384, 128, 0, 128, (3, 1), 1),
    ("temporal 256ch (3x1) 64^2", 16, 7, 4096, 256, 0, 256, (3, 1), 1),
    ("1x1 qkv 512->1536 16^2", 1, 1, 112 * 256, 512, 0, 1536, 1, 1),
]


def main():
    for name, N, H, W, C1, C2, Co, k, s in SHAPES:
        kh, kw = (k, k) if isinstance(k, int) else k
        x = torch.randn(N, H, W, C1, device=dev).to(torch.bfloat16)
        x2 = torch.randn(N, H, W, C2, device=dev).to(torch.bfloat16) if C2 else None
        w = torch.randn(Co, C1 + C2, kh, kw, device=dev) * 0.02
        wp = ops.pack_weight_h(w)
        b = torch.randn(Co, device=dev)
        f = lambda: ops.conv2d_h(x, wp, b, Co, kh, kw, (s, s), (kh // 2, kw // 2), x2=x2)
        y = f()
        M = y.shape[0] * y.shape[1] * y.shape[2]
        fl = 2.0 * M * Co * kh * kw * (C1 + C2)
        t = timeit(f)
        print(f"{name:34s} M={M:8d} K={kh*kw*(C1+C2):6d} N={Co:5d}  {t*1e6:9.1f} us {fl/t/1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
