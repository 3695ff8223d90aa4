"""GroupNorm forward / backward on the policy step's shapes, GPU time per launch under hipGraph replay (run on the GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops

dev = "cuda:0"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
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
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


SHAPES = [  # N, S, C, G, act
    (64, 4096, 64, 4, "relu"), (64, 1024, 64, 4, "relu"), (64, 256, 128, 8, "relu"), (64, 64, 256, 16, "relu"), (64, 16, 512, 32, "relu"),
    (64, 16, 256, 8, "mish"), (64, 8, 512, 8, "mish"), (64, 4, 1024, 8, "mish"), (64, 16, 512, 8, "mish"), (64, 16, 1024, 8, "mish"),
]
for N, S, C, G, act in SHAPES:
    x = torch.randn(N, S, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    dout = torch.randn(N, S, C, device=dev)
    y, mean, rstd = ops.groupnorm_fwd(x, g, b, G, act)
    kf = ops.last_kernel[0] if hasattr(ops, "last_kernel") else ""
    tf = timeit(lambda: ops.groupnorm_fwd(x, g, b, G, act))
    tb = timeit(lambda: ops.groupnorm_bwd(x, g, b, G, dout, mean, rstd, act=act))
    mb = x.numel() * 4 / 1e6
    print(f"N={N} S={S:5d} C={C:5d} G={G:3d} E={S*C//G:6d} {act}: tensor {mb:6.2f} MB  fwd {tf:6.1f} us ({2*mb/tf/1e0:7.1f} GB/s... {2*mb/tf*1e-3:5.2f} TB/s)  bwd {tb:6.1f} us ({3*mb/tb*1e-3:5.2f} TB/s)", flush=True)

print("--- backward with the producing conv's split-K slabs folded in (what the train step launches)")
for N, S, C, G, act in [(64, 64, 256, 16, "relu"), (64, 16, 512, 32, "relu"), (64, 16, 256, 8, "mish"), (64, 256, 128, 8, "relu"), (64, 1024, 64, 4, "relu")]:
    x = torch.randn(N, S, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    res = torch.randn(N, S, C, device=dev)
    y, mean, rstd = ops.groupnorm_fwd(x, g, b, G, act, residual=res)
    cs = torch.empty((N, 2, C), device=dev)
    for nslab in (0, 2, 4, 8):
        ws = torch.randn(max(nslab, 1), N * S * C, device=dev)
        dsum = torch.empty(N, S, C, device=dev)
        sl = ops.Slabs(ws, nslab, N * S * C, None, None) if nslab else None
        dout = ws[0].view(N, S, C)
        for with_res in (False, True):
            fn = lambda: ops.groupnorm_bwd(x, g, b, G, dout, mean, rstd, act, residual=res if with_res else None, want_dres=with_res,
                                           colsum=cs, defer_params=True, dout_slabs=sl, dout_sum=dsum if sl is not None else None)
            t = timeit(fn)
            print(f"N={N} S={S:5d} C={C:5d} G={G:3d} nslab={nslab} residual={int(with_res)}: bwd {t:6.1f} us", flush=True)
