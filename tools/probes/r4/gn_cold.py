"""gn_wave16_bwd<4> (8x8x256, two slabs + residual) with warm and with cold caches (a 1 GB fill between launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
dev = "cuda:0"
big = torch.empty(1 << 28, device=dev)       # 1 GiB of floats
def run(N, S, C, G, nslab, cold, act="relu"):
    x = torch.randn(N, S, C, device=dev); g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    res = torch.randn(N, S, C, device=dev)
    y, mean, rstd = ops.groupnorm_fwd(x, g, b, G, act, residual=res)
    cs = torch.empty((N, 2, C), device=dev)
    ws = torch.randn(max(nslab, 1), N * S * C, device=dev)
    dsum = torch.empty(N, S, C, device=dev)
    sl = ops.Slabs(ws, nslab, N * S * C, None, None) if nslab else None
    dout = ws[0].view(N, S, C)
    fn = lambda: ops.groupnorm_bwd(x, g, b, G, dout, mean, rstd, act, residual=res, want_dres=True, colsum=cs, defer_params=True,
                                   dout_slabs=sl, dout_sum=dsum if sl is not None else None)
    ts = []
    for _ in range(12):
        if cold:
            big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for shape in [(64, 64, 256, 16, 2), (64, 16, 512, 32, 4), (64, 256, 128, 8, 0), (64, 1024, 64, 4, 0)]:
    print(shape, "warm", f"{run(*shape, cold=False):.1f} us", "cold", f"{run(*shape, cold=True):.1f} us", flush=True)
