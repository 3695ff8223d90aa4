"""fp32 conv: register-staged kernel vs the LDS-DMA kernel on the dominant shapes (GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops
from tools.conv_bench import timeit, FWD

dev = "cuda:0"
for name, N, H, W, Ci, Co, k, s in FWD:
    kh, kw = (k, k) if isinstance(k, int) else k
    if Ci % 32:
        continue
    x = torch.randn(N, H, W, Ci, device=dev)
    w = torch.randn(Co, kh * kw * Ci, device=dev) * 0.02
    b = torch.randn(Co, device=dev)
    res = {}
    for mode, rows in (("staged", 1 << 60), ("dma", 0)):
        ops._DMA_F32_MIN_WORK[0] = rows
        f = lambda: ops.conv2d(x, w, b, Co, kh, kw, (s, s), (kh // 2, kw // 2))
        y = f()
        res[mode] = (timeit(f), y)
    M = y.shape[0] * y.shape[1] * y.shape[2]
    fl = 2.0 * M * Co * kh * kw * Ci
    err = ((res["dma"][1] - res["staged"][1]).abs().max() / res["staged"][1].abs().max()).item()
    print(f"{name:36s} M={M:8d} K={kh*kw*Ci:6d} N={Co:5d}  staged {res['staged'][0]*1e6:8.1f} us {fl/res['staged'][0]/1e12:6.1f} TF | dma {res['dma'][0]*1e6:8.1f} us {fl/res['dma'][0]/1e12:6.1f} TF | rel diff {err:.1e}", flush=True)
