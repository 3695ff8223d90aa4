"""Weight gradients of the policy step's dominant shapes through the grouped launch: time per launch and error against an fp64 reference.
Run once per kernel family: V2A_WGRAD_X3=1 (three bf16 planes) / unset (exact-f32 halo + 64x64 bodies).
Usage (GPU box): [V2A_WGRAD_X3=1] python tools/probes/wgrad_x3_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from v2a_hip import ops

dev = "cuda:0"
# name, N, H, W, Cin, Cout, (kh, kw), stride
SHAPES = [
    ("res1 3x3 64->64   32x32", 64, 32, 32, 64, 64, (3, 3), 1),
    ("res2 3x3 128->128 16x16", 64, 16, 16, 128, 128, (3, 3), 1),
    ("res2 3x3 64->128 s2", 64, 32, 32, 64, 128, (3, 3), 2),
    ("res3 3x3 256->256  8x8", 64, 8, 8, 256, 256, (3, 3), 1),
    ("res4 3x3 512->512  4x4", 64, 4, 4, 512, 512, (3, 3), 1),
    ("unet l0 k5 256->256 T16", 64, 1, 16, 256, 256, (1, 5), 1),
    ("unet l1 k5 512->512 T8", 64, 1, 8, 512, 512, (1, 5), 1),
    ("unet l2 k5 1024->1024 T4", 64, 1, 4, 1024, 1024, (1, 5), 1),
]
mode = "x3" if os.environ.get("V2A_WGRAD_X3") == "1" else "exact"
col = ops.WgradCollector(dev)
for name, N, H, W, Ci, Co, (kh, kw), s in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(N, Ci, H, W)
    w = torch.zeros(Co, Ci, kh, kw, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), w, stride=s, padding=(kh // 2, kw // 2))
    dy = torch.randn(*y.shape)
    y.backward(dy.double())
    ref = w.grad
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    dw = torch.empty(Co, Ci, kh, kw, device=dev)

    def f():
        b = ops.WgradBatch(col)
        assert b.add(xd, dyd, (Co, Ci, kh, kw), kh, kw, (s, s), (kh // 2, kw // 2), dw=dw, slab_key=("p", name))
        b.launch()
        col.flush()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = sorted(ts)[2]
    fl = 2.0 * y.shape[0] * y.shape[2] * y.shape[3] * Co * kh * kw * Ci
    err = (dw.cpu().double() - ref)
    print(f"{name:28s} {mode:6s} {t:8.1f} us (main + reduce) {fl / t / 1e6:7.1f} TF | rms err / rms {(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e}"
          f"  max err / max {(err.abs().max() / ref.abs().max()).item():.2e}", flush=True)
