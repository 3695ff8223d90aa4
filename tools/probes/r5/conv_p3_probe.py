"""conv_p3 (pre-split three-plane operands, pure LDS-DMA) against conv_igemm_f32x3<64,64> on the ConditionalUnet1D conv shapes: bit equality
(finished tensor and deferred slabs) and device time per launch under hipGraph replay.  Usage (GPU box): python tools/probes/r5/conv_p3_probe.py"""
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


SHAPES = [("l0 256->256 T16", 16, 256, 0, 256, 5, 1), ("l1 256->512 T8", 8, 256, 0, 512, 5, 1), ("l1 512->512 T8", 8, 512, 0, 512, 5, 1),
          ("l2 512->1024 T4", 4, 512, 0, 1024, 5, 1), ("l2 1024->1024 T4", 4, 1024, 0, 1024, 5, 1), ("up 1024+1024->512 T4", 4, 1024, 1024, 512, 5, 1),
          ("up 512+512->256 T8", 8, 512, 512, 256, 5, 1), ("rc 1024+1024->512 1x1", 4, 1024, 1024, 512, 1, 1), ("ds 512->512 k3 s2 T8", 8, 512, 0, 512, 3, 2)]
for name, T, C1, C2, Co, k, st in SHAPES:
    g = torch.Generator().manual_seed(T * 1000 + Co)
    x = torch.randn(64, 1, T, C1, generator=g).to(dev)
    x2 = torch.randn(64, 1, T, C2, generator=g).to(dev) if C2 else None
    w = (torch.randn(Co, k * (C1 + C2), generator=g) * 0.02).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    x3, w3 = ops.split3(x), ops.split3(w)
    x23 = ops.split3(x2) if C2 else None
    # planes reproduce the tensor: hi + mid + lo == x to the last bit for normal numbers
    rec = x3[0].float() + x3[1].float() + x3[2].float()
    assert torch.equal(rec, x), (name, float((rec - x).abs().max()))
    pad = (0, k // 2)
    y0 = ops.conv2d(x, w, b, Co, 1, k, (1, st), pad, x2=x2)
    k0 = ops.last_kernel[0]
    y1 = ops.conv2d_p3(x3, w3.view(3, -1), b, Co, 1, k, (1, st), pad, x2_3=x23)
    torch.cuda.synchronize()
    eq = torch.equal(y0, y1)
    ya, sa = ops.conv2d(x, w, b, Co, 1, k, (1, st), pad, x2=x2, defer=True)
    sl_a = sa.ws[:sa.n * sa.stride * 4].clone() if sa is not None else None
    yb, sb = ops.conv2d_p3(x3, w3.view(3, -1), b, Co, 1, k, (1, st), pad, x2_3=x23, defer=True)
    sl_b = sb.ws[:sb.n * sb.stride * 4].clone() if sb is not None else None
    eqs = (sa is None and sb is None) or (sa is not None and sb is not None and sa.n == sb.n and torch.equal(sl_a, sl_b))
    t0 = timeit(lambda: ops.conv2d(x, w, b, Co, 1, k, (1, st), pad, x2=x2, defer=True))
    t1 = timeit(lambda: ops.conv2d_p3(x3, w3.view(3, -1), b, Co, 1, k, (1, st), pad, x2_3=x23, defer=True))
    fl = 2.0 * y0.numel() * k * (C1 + C2)
    print(f"{name:26s} {k0:22s} equal={eq} slabs_equal={eqs} nslab={0 if sa is None else sa.n:2d}  x3 {t0:6.1f} us ({fl / t0 / 1e6:6.1f} TF)  "
          f"p3 {t1:6.1f} us ({fl / t1 / 1e6:6.1f} TF)  x{t0 / t1:.2f}", flush=True)
