"""Device time per launch (hipGraph replay) of ConditionalUnet1D conv shapes: k = 5 (1 x 5) over [B = 64, T] with deep reductions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
dev = "cuda:0"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
for name, T, Ci, Co in [("l0 256->256 T16", 16, 256, 256), ("l1 256->512 T8", 8, 256, 512), ("l1 512->512 T8", 8, 512, 512),
                        ("l2 512->1024 T4", 4, 512, 1024), ("l2 1024->1024 T4", 4, 1024, 1024), ("up 2048->512 T8", 8, 2048, 512)]:
    x = torch.randn(64, 1, T, Ci, device=dev); w = torch.randn(Co, 5 * Ci, device=dev) * 0.02
    for defer in (False, True):
        def f():
            return ops.conv2d(x, w, None, Co, 1, 5, (1, 1), (0, 2), defer=defer)
        t = timeit(f)
        fl = 2.0 * 64 * T * Co * 5 * Ci
        wb = Co * 5 * Ci * 4 / 1e6
        print(f"{name:18s} defer={int(defer)}: {t:6.1f} us  {fl / t / 1e6:6.1f} TF  weights {wb:5.1f} MB -> {wb / t * 1e-0:6.2f} TB/s", flush=True)
