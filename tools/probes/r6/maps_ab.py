"""A/B of conv_maps_x3 (csrc/igemm_x3m.hip) against conv_halo_x3 on the four encoder layer shapes at batch 64, HIP events on the stream.
Usage (GPU box): python tools/probes/r6/maps_ab.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
from v2a_hip._lib import lib

dev = torch.device("cuda:0")
for (N, S, C) in ((64, 32, 64), (64, 16, 128), (64, 8, 256), (64, 4, 512)):
    x = torch.randn(N, S, S, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wp = ops.pack_weight(w, 0)
    out = {}
    for on in (0, 1):
        old = lib.v2a_debug_set_maps_kernel(on)
        for defer in (False,):
            for _ in range(5):
                y = ops.conv2d(x, wp, None, C, 3, 3, (1, 1), (1, 1))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                y = ops.conv2d(x, wp, None, C, 3, 3, (1, 1), (1, 1))
            e1.record()
            torch.cuda.synchronize()
            out[on] = (e0.elapsed_time(e1) / 50 * 1e3, ops.last_kernel[0])
        lib.v2a_debug_set_maps_kernel(old)
    fl = 2.0 * N * S * S * C * C * 9
    print(f"N={N} {S}x{S} C={C}: halo {out[0][0]:.1f} us ({out[0][1]})  maps {out[1][0]:.1f} us ({out[1][1]})  "
          f"{fl / out[1][0] / 1e6:.0f} TFLOP/s (incl. split-K reduce)")
