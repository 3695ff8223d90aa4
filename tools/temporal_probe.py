"""Probe: temporal (3x1) bf16 conv at the sampler's level-1 shape vs a 1x1 control of the same M (GPU box; run under rocprofv3 --pmc)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops
from tools.conv_bench import timeit

dev = "cuda:0"
B, Fr, HW, C = 16, 7, 16384, 128
x = torch.randn(B, Fr, HW, C, device=dev).to(torch.bfloat16)
for name, kh, kw in (("temporal 3x1", 3, 1), ("control 1x1", 1, 1)):
    w = torch.randn(C, C, kh, kw, device=dev) * 0.02
    wp = ops.pack_weight_h(w)
    b = torch.randn(C, device=dev)
    f = lambda: ops.conv2d_h(x, wp, b, C, kh, kw, (1, 1), (kh // 2, 0))
    f()
    t = timeit(f, 5) if not os.environ.get("PROBE_ONCE") else 0.0
    if t:
        M = B * Fr * HW
        print(f"{name:14s} {t*1e6:8.1f} us  {2.0*M*C*C*kh*kw/t/1e12:6.1f} TF  min-bytes {2*M*C*2/1e6:.0f} MB -> {2*M*C*2/t/1e12:.2f} TB/s", flush=True)
