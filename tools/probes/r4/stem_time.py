import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
dev = "cuda:0"
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
N, H, W = 64, 128, 128
img = torch.rand(N, 3, H, W, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
xp = torch.zeros((N, H + 6, W + 6, 4), device=dev)
ops.nchw_to_nhwc4p(img, xp, 3, normalize=True)
pw = torch.zeros(64 * 7 * 8 * 4, device=dev); ops.pack_weight(w, 2, pw)
pf = ops.pack_weight(w, 0)
x0 = ops.nchw_to_nhwc(img, normalize=True)
print("window conv        ", timeit(lambda: ops.conv2d_window(xp, pw, 64, 7, 1, (2, 1), (64, 64), xpitch=8, C=32)), "us")
print("scalar-gather conv ", timeit(lambda: ops.conv2d(x0, pf, None, 64, 7, 7, (2, 2), (3, 3))), "us")
dy = torch.randn(N, 64, 64, 64, device=dev)
print("wgrad (padded buf) ", timeit(lambda: ops.conv2d_wgrad(xp, dy, (64, 4, 7, 7), 7, 7, (2, 2), (0, 0))), "us")
print("nchw_to_nhwc4p     ", timeit(lambda: ops.nchw_to_nhwc4p(img, xp, 3, normalize=True)), "us")
print("nchw_to_nhwc       ", timeit(lambda: ops.nchw_to_nhwc(img, normalize=True)), "us")
x1 = torch.randn(64, 32, 32, 64, device=dev); w1 = torch.randn(64, 576, device=dev)
print("res1 conv          ", timeit(lambda: ops.conv2d(x1, w1, None, 64, 3, 3, (1, 1), (1, 1))), "us")
