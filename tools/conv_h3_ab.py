"""A/B of the halo-tile 3x3 kernel (csrc/igemm_h3.hip) against the tap-by-tap kernels on the video UNet's 3x3 shapes, interleaved in one
process (cdna_hip_programming.md 5.4 rule 24).  Run on the GPU box."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops

dev = "cuda:0"
SHAPES = [  # name, N, H, W, C, Cout
    ("128^2 128->128", 112, 128, 128, 128, 128),
    ("128^2 256->128", 112, 128, 128, 256, 128),
    ("64^2 128->256", 112, 64, 64, 128, 256),
    ("64^2 256->256", 112, 64, 64, 256, 256),
    ("64^2 512->256", 112, 64, 64, 512, 256),
    ("32^2 256->384", 112, 32, 32, 256, 384),
    ("32^2 384->384", 112, 32, 32, 384, 384),
    ("32^2 768->384", 112, 32, 32, 768, 384),
    ("16^2 512->512", 112, 16, 16, 512, 512),
    ("16^2 1024->512", 112, 16, 16, 1024, 512),
]


def bench(f, rounds=5, iters=6):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
    return best


for name, N, H, W, C, Co in SHAPES:
    x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.02
    wp = ops.pack_weight_h(w)
    b = torch.randn(Co, device=dev)
    f = lambda: ops.conv2d_h(x, wp, b, Co, 3, 3, (1, 1), (1, 1))
    fl = 2.0 * N * H * W * Co * 9 * C
    res = {}
    for rnd in range(2):
        for tag in ("h3", "old"):
            os.environ.pop("V2A_CONV_H3_OFF_FOR_TEST", None)
            if tag == "old":
                os.environ["V2A_CONV_H3_OFF_FOR_TEST"] = "1"
            t = bench(f)
            kn = ops.last_kernel[0]
            res[tag] = min(res.get(tag, (1e9, ""))[0], t), kn
    os.environ.pop("V2A_CONV_H3_OFF_FOR_TEST", None)
    print(f"{name:18s} h3 {fl/res['h3'][0]/1e12:7.1f} TF ({res['h3'][1]:24s}) | old {fl/res['old'][0]/1e12:7.1f} TF ({res['old'][1]})", flush=True)
