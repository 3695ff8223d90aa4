"""Micro-benchmark of the 16-bit sampler's conv kernels on the shapes of the B=16 forward (run on the GPU box).
Usage: halo_probe.py [pmc]   ("pmc": one launch per shape, for rocprofv3 --pmc)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops

dev = "cuda:0"
PMC = len(sys.argv) > 1 and sys.argv[1] == "pmc"
SHAPES = [  # N, H, W, C1, C2, Cout
    (112, 128, 128, 128, 0, 128),
    (112, 128, 128, 128, 128, 128),
    (112, 64, 64, 256, 0, 256),
    (112, 64, 64, 256, 256, 256),
    (112, 64, 64, 128, 0, 256),
    (112, 32, 32, 384, 0, 384),
    (112, 32, 32, 384, 384, 384),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if PMC:
        return 1.0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
    return best


F = 7
for N, H, W, C1, C2, Co in SHAPES:
    C = C1 + C2
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, C1, generator=g).to(torch.bfloat16).to(dev)
    x2 = torch.randn(N, H, W, C2, generator=g).to(torch.bfloat16).to(dev) if C2 else None
    w = ops.pack_weight_h((torch.randn(Co, C, 3, 3, generator=g) * 0.05).to(dev))
    b = torch.randn(Co, generator=g).to(dev)
    gamma, beta = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    S = F * H * W
    fl = 2.0 * N * H * W * Co * 9 * C
    pg = ops.groupnorm_prep_h(x.view(N // F, S, C1), gamma, beta, 32, "silu", x2=None if x2 is None else x2.view(N // F, S, C2))
    t_gn = timeit(lambda: ops.conv2d_h(x, w, b, Co, 3, 3, (1, 1), (1, 1), x2=x2, rows_per_batch=S, want_stats=True, pre_gn=pg))
    k_gn = ops.last_kernel[0]
    xa = pg.apply().view(N, H, W, C)
    t_pl = timeit(lambda: ops.conv2d_h(xa, w, b, Co, 3, 3, (1, 1), (1, 1), rows_per_batch=S, want_stats=True))
    k_pl = ops.last_kernel[0]
    print(f"N={N} {H}x{W} C={C1}+{C2}->{Co}: plain {t_pl*1e6:8.1f} us {fl/t_pl/1e12:7.1f} TF [{k_pl}] | gn {t_gn*1e6:8.1f} us {fl/t_gn/1e12:7.1f} TF [{k_gn}]", flush=True)
