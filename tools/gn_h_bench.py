"""Micro-benchmark of the bf16-storage GroupNorm and attention kernels on the video UNet's shapes (GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops
from tools.conv_bench import timeit

dev = "cuda:0"
for name, N, S, C1, C2 in [("128^2 C128", 16, 7 * 128 * 128, 128, 0), ("64^2 C256", 16, 7 * 64 * 64, 256, 0),
                           ("64^2 C256+128 concat", 16, 7 * 64 * 64, 256, 128), ("32^2 C384", 16, 7 * 32 * 32, 384, 0),
                           ("16^2 C512", 16, 7 * 256, 512, 0), ("8^2 C640+640", 16, 7 * 64, 640, 640),
                           ("attn-norm 16^2 C512 per frame", 112, 256, 512, 0)]:
    C = C1 + C2
    x = torch.randn(N, S, C1, device=dev).to(torch.bfloat16)
    x2 = torch.randn(N, S, C2, device=dev).to(torch.bfloat16) if C2 else None
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    t = timeit(lambda: ops.groupnorm_fwd_h(x, g, b, 32, "silu", x2=x2))
    byt = N * S * C * 2 * 3
    print(f"GN {name:32s} {t*1e6:8.1f} us  {byt/t/1e12:5.2f} TB/s (3 x 2 B per element)", flush=True)
for name, N, L, heads, hc in [("attn 16^2 C512", 112, 256, 16, 32), ("attn 8^2 C640", 112, 64, 20, 32)]:
    qkv = torch.randn(N * L, 3 * heads * hc, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.attention(qkv, N, L, heads, hc))
    fl = 4.0 * N * heads * L * L * hc
    print(f"{name:35s} {t*1e6:8.1f} us  {fl/t/1e12:6.2f} TFLOP/s", flush=True)
