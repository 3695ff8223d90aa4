"""Stem weight gradient (7 x 7 / stride 2 over the zero-bordered 4-channel image buffer, K = 196, reduction over 262 144 output pixels):
as it is (49 taps of one 16-B pixel) against the same sums through the PAIRED view of the buffer ([N, Hp, Wp / 2, 8]: 7 x 4 taps of two
pixels, stride (2, 1)).  HIP events, same process.  Usage (GPU box): python tools/probes/r6/stem_wgrad_ab.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops

dev = torch.device("cuda:0")
N, H, W, Co = 64, 128, 128, 64
xp = torch.zeros(N, H + 6, W + 6, 4, device=dev)
xp[:, 3:-3, 3:-3, :3] = torch.randn(N, H, W, 3, device=dev)
dy = torch.randn(N, H // 2, W // 2, Co, device=dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


ta, dw4 = t(lambda: ops.conv2d_wgrad(xp, dy, (Co, 4, 7, 7), 7, 7, (2, 2), (0, 0)))
ka = ops.last_kernel[0]
x8 = xp.view(N, H + 6, (W + 6) // 2, 8)
tb, dw8 = t(lambda: ops.conv2d_wgrad(x8, dy, (Co, 8, 7, 4), 7, 4, (2, 1), (0, 0)))
kb = ops.last_kernel[0]
# dw8 [co][p * 4 + c][kh][kw'] -> dw [co][c][kh][2 kw' + p]
d = dw8.view(Co, 2, 4, 7, 4).permute(0, 2, 3, 4, 1).reshape(Co, 4, 7, 8)[..., :7]
err = (d - dw4).abs().max().item() / dw4.abs().max().item()
print(f"4-channel pixels, 7 x 7 taps: {ta:.1f} us ({ka});  pixel pairs, 7 x 4 taps: {tb:.1f} us ({kb});  max rel diff {err:.2e}")
