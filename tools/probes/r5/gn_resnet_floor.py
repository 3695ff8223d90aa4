"""GroupNorm launches of the ResNet-18 encoders (B = 64: layer1 64ch 32x32, layer2 128ch 16x16, layer3 256ch 8x8, layer4 512ch 4x4; 16 channels
per group, ReLU) as chains of 20 launches in a replayed graph: microseconds per launch, forward and backward, against the bytes they move."""
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


B = 64
for name, C, hw in (("layer1", 64, 32), ("layer2", 128, 16), ("layer3", 256, 8), ("layer4", 512, 4)):
    S, G = hw * hw, C // 16
    x = torch.randn(B, S, C, device=dev)
    res = torch.randn(B, S, C, device=dev)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    y, mean, rstd = ops.groupnorm_fwd(x, gamma, beta, G, "relu")
    dout = torch.randn(B, S, C, device=dev)
    mb = x.numel() * 4 / 1e6
    t_f = timeit(lambda: ops.groupnorm_fwd(x, gamma, beta, G, "relu"))
    t_fr = timeit(lambda: ops.groupnorm_fwd(x, gamma, beta, G, "relu", residual=res))
    t_b = timeit(lambda: ops.groupnorm_bwd(x, gamma, beta, G, dout, mean, rstd, "relu", defer_params=True))
    t_br = timeit(lambda: ops.groupnorm_bwd(x, gamma, beta, G, dout, mean, rstd, "relu", residual=res, want_dres=True, defer_params=True))
    print(f"{name}: tensor {mb:5.1f} MB   fwd {t_f:5.1f} us ({2 * mb / t_f:5.2f} TB/s)   fwd+residual {t_fr:5.1f} us   bwd {t_b:5.1f} us ({3 * mb / t_b:5.2f} TB/s)   "
          f"bwd+residual {t_br:5.1f} us", flush=True)
