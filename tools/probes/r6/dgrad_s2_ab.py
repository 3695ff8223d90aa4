"""Stride-2 data-gradient launches of the policy's ResNet-18 encoders (B = 64 per camera) with the parity classes on / off (run on the GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
from v2a_hip._lib import lib

dev = "cuda:0"
SHAPES = [("layer2.0.conv1", 64, 16, 128, 64, 3), ("layer3.0.conv1", 64, 8, 256, 128, 3), ("layer4.0.conv1", 64, 4, 512, 256, 3),
          ("layer2.0.down", 64, 16, 128, 64, 1), ("layer3.0.down", 64, 8, 256, 128, 1), ("layer4.0.down", 64, 4, 512, 256, 1)]


def bench(f, iters=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for name, N, H, Ci, Co, k in SHAPES:
    dy = torch.randn(N, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    wp = ops.pack_weight(w, 0) if k > 1 else w.reshape(Co, Ci).contiguous()
    res = torch.randn(N, 2 * H, 2 * H, Co, device=dev)
    out = {}
    for on in (1, 0, 1, 0):
        lib.v2a_debug_set_parity_classes(on)
        f = lambda: ops.conv2d(dy, wp, None, Co, k, k, (1, 1), (k - 1 - k // 2,) * 2, idil=2, out_hw=(2 * H, 2 * H), residual=res)
        out.setdefault(on, []).append(bench(f))
        kn = ops.last_kernel[0]
    lib.v2a_debug_set_parity_classes(1)
    print(f"{name:16s} {kn:28s} classes on {min(out[1]):7.1f} us | off {min(out[0]):7.1f} us", flush=True)
