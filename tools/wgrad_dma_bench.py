"""fp32 weight gradient: register-staged kernel vs the LDS-DMA kernel (GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops
from v2a_hip._lib import lib
from tools.conv_bench import timeit, FWD

dev = "cuda:0"
for name, N, H, W, Ci, Co, k, s in FWD:
    kh, kw = (k, k) if isinstance(k, int) else k
    if Ci % 4 or Co % 4 or "video" in name:
        continue
    x = torch.randn(N, H, W, Ci, device=dev)
    OH, OW = (H + 2 * (kh // 2) - kh) // s + 1, (W + 2 * (kw // 2) - kw) // s + 1
    dy = torch.randn(N, OH, OW, Co, device=dev)
    res = {}
    for mode in (0, 1):
        lib.v2a_debug_wgrad_dma(mode)
        f = lambda: ops.conv2d_wgrad(x, dy, (Co, Ci, kh, kw), kh, kw, (s, s), (kh // 2, kw // 2))
        dw = f()
        res[mode] = (timeit(f), dw.clone(), ops.last_kernel[0])
    M = N * OH * OW
    fl = 2.0 * M * Co * kh * kw * Ci
    err = ((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max()).item()
    print(f"{name:32s} M={M:7d} K={kh*kw*Ci:6d} N={Co:5d} {res[0][2]:24s} staged {res[0][0]*1e6:7.1f} us {fl/res[0][0]/1e12:6.1f} TF | dma {res[1][0]*1e6:7.1f} us {fl/res[1][0]/1e12:6.1f} TF | rel diff {err:.1e}", flush=True)
