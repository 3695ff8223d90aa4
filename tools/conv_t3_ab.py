"""A/B of the frame-stack temporal kernel (csrc/igemm_h3.hip conv_frames_h3) against the tap-by-tap kernels on the video UNet's
temporal-conv shapes, interleaved in one process.  Run on the GPU box."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from v2a_hip import ops

dev = "cuda:0"
SHAPES = [("128^2 128", 16, 16384, 128, 128), ("128^2 256", 16, 16384, 256, 256), ("64^2 256", 16, 4096, 256, 256), ("64^2 384", 16, 4096, 384, 384),
          ("32^2 384", 16, 1024, 384, 384), ("16^2 512", 16, 256, 512, 512)]


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


for name, B, HW, C, Co in SHAPES:
    x = torch.randn(B, 7, HW, C, device=dev).to(torch.bfloat16)
    wp = ops.pack_weight_h(torch.randn(Co, C, 3, 1, device=dev) * 0.02)
    b = torch.randn(Co, device=dev)
    f = lambda: ops.conv2d_h(x, wp, b, Co, 3, 1, (1, 1), (1, 0), want_stats=not os.environ.get("T3_NOSTATS"))
    fl = 2.0 * B * 7 * HW * Co * 3 * C
    by = 2.0 * B * 7 * HW * (C + Co)
    res = {}
    for rnd in range(2):
        for tag in ("t3", "old"):
            if tag == "old":
                os.environ["V2A_CONV_H3_OFF_FOR_TEST"] = "1"
            else:
                os.environ.pop("V2A_CONV_H3_OFF_FOR_TEST", None)
            t = bench(f)
            res[tag] = min(res.get(tag, (1e9, ""))[0], t), ops.last_kernel[0]
    os.environ.pop("V2A_CONV_H3_OFF_FOR_TEST", None)
    print(f"{name:12s} t3 {res['t3'][0]*1e6:7.1f} us {fl/res['t3'][0]/1e12:7.1f} TF {by/res['t3'][0]/1e12:5.2f} TB/s ({res['t3'][1]}) | "
          f"old {res['old'][0]*1e6:7.1f} us {fl/res['old'][0]/1e12:7.1f} TF ({res['old'][1]})", flush=True)
