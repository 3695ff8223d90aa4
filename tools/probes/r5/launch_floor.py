"""What a dependent launch costs inside a replayed hipGraph, and what the ConditionalUnet1D's GroupNorm launches do on top of it: chains of 40
same-stream launches of (a) a 4-element axpy, (b) gn_wavev_fwd over split-K slabs at the UNet's shapes, (c) the same GroupNorm on a finished
tensor.  Usage: python tools/probes/r5/launch_floor.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
dev = "cuda:0"


def timeit(fn, iters=40):
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


a = torch.zeros(4, device=dev); b = torch.ones(4, device=dev)
print(f"4-element axpy chain: {timeit(lambda: ops.axpy(a, b, out=a)):.2f} us per launch")
big = torch.zeros(1 << 20, device=dev); big2 = torch.ones(1 << 20, device=dev)
print(f"4 MB axpy chain: {timeit(lambda: ops.axpy(big, big2, out=big)):.2f} us per launch")
for T, C in ((16, 256), (8, 512), (4, 1024), (8, 256), (4, 512)):
    B, G = 64, 8
    x = torch.randn(B, T, C, device=dev)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    film = torch.randn(B, 2 * C, device=dev)
    w = torch.randn(C, 5 * C, device=dev) * 0.02
    xin = torch.randn(B, 1, T, C, device=dev)
    y, sl = ops.conv2d(xin, w, None, C, 1, 5, (1, 1), (0, 2), defer=True)
    t_s = timeit(lambda: ops.groupnorm_fwd(y.view(B, T, C), gamma, beta, G, "mish", film=film, slabs=sl))
    t_d = timeit(lambda: ops.groupnorm_fwd(x, gamma, beta, G, "mish", film=film))
    mean = torch.zeros(B * G, device=dev); rstd = torch.ones(B * G, device=dev)
    dout = torch.randn(B, T, C, device=dev)
    t_b = timeit(lambda: ops.groupnorm_bwd(x, gamma, beta, G, dout, mean, rstd, "mish", film=film, want_dfilm=True, defer_params=True,
                                           colsum=torch.empty(B, 2, C, device=dev) if False else None))
    print(f"T={T:2d} C={C:4d}: GroupNorm fwd over {sl.n if sl else 0:2d} slabs {t_s:5.2f} us, on a finished tensor {t_d:5.2f} us, bwd (dense) {t_b:5.2f} us")
