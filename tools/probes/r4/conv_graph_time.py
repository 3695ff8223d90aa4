"""Device time per launch of the policy's 3x3 convs under hipGraph replay (no host launch overhead in the number)."""
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
for name, N, H, C in [("res1", 64, 32, 64), ("res2", 64, 16, 128), ("res3", 64, 8, 256), ("res4", 64, 4, 512)]:
    x = torch.randn(N, H, H, C, device=dev); w = torch.randn(C, 9 * C, device=dev) * 0.05
    y = torch.empty(N, H, H, C, device=dev)
    t = timeit(lambda: ops.conv2d(x, w, None, C, 3, 3, (1, 1), (1, 1), y=y))
    fl = 2.0 * N * H * H * C * 9 * C
    print(f"{name}: {t:6.1f} us  {fl / t / 1e6:6.1f} TF", flush=True)
