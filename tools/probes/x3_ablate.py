"""conv_igemm_f32x3 with parts switched off (v2a_debug_x3_ablate): which of split VALU / LDS stores / MFMA bounds the kernel?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd")); sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops
from v2a_hip._lib import lib
dev = "cuda:0"
SHAPES = [("res1 64->64 32x32 B64", 64, 32, 32, 64, 64), ("res2 128->128 16x16", 64, 16, 16, 128, 128), ("res3 256->256 8x8", 64, 8, 8, 256, 256),
          ("res4 512->512 4x4", 64, 4, 4, 512, 512), ("vid 256->256 64x64 (2x7)", 14, 64, 64, 256, 256)]
MODES = [("full", 0), ("no A split", 1), ("no B split", 2), ("no split", 3), ("no LDS stores", 4), ("no split, no stores", 7), ("no MFMA", 8),
         ("loads + barriers only", 15)]
for name, N, H, W, Ci, Co in SHAPES:
    x = torch.randn(N, H, W, Ci, device=dev); w = torch.randn(Co, 9 * Ci, device=dev) * 0.05
    f = lambda: ops.conv2d(x, w, None, Co, 3, 3, (1, 1), (1, 1))
    fl = 2.0 * N * H * W * Co * 9 * Ci
    out = []
    for mn, bits in MODES:
        lib.v2a_debug_x3_ablate(bits)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        out.append(f"{mn} {sorted(ts)[2]:.1f}")
    lib.v2a_debug_x3_ablate(0)
    print(f"{name:28s} {ops.last_kernel[0]:26s} us: " + " | ".join(out) + f" | bf16x6 matrix floor {fl * 6 / 2.5e15 * 1e6:.1f}", flush=True)
