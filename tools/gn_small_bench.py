"""Timing of the one-workgroup-per-slab GroupNorm kernels on the policy's ResNet shapes (GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops

dev = "cuda:0"
for N, S, C, G in [(64, 1024, 64, 4), (64, 256, 128, 8), (64, 64, 256, 16), (64, 16, 512, 32), (128, 1024, 64, 4)]:
    x = torch.randn(N, S, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    dout = torch.randn(N, S, C, device=dev)
    y, mean, rstd = ops.groupnorm_fwd(x, g, b, G, "relu")
    f = lambda: ops.groupnorm_fwd(x, g, b, G, "relu")
    fb = lambda: ops.groupnorm_bwd(x, g, b, G, dout, mean, rstd, act="relu")
    for name, fn in (("fwd", f), ("bwd", fb)):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        nbytes = x.numel() * 4 * (2 if name == "fwd" else 3)
        print(f"N={N} S={S} C={C} G={G} {name}: {best:7.1f} us  {nbytes / best / 1e6:6.2f} TB/s", flush=True)
