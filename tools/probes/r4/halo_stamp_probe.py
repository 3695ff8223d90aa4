"""Where does a conv_igemm_h launch spend its time?  In-kernel 100-MHz stamps per workgroup (v2a_debug_conv_stamps) on the policy
step's dominant shapes: entry skew, first-DMA latency, main loop, epilogue, store acknowledgement, and the gap to the next launch.
Usage (GPU box): python tools/probes/conv_stamp_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from v2a_hip import ops
from v2a_hip._lib import lib, check

dev = "cuda:0"
# (name, N, H, W, Cin, Cout, k, stride)  -- ResNet18-GN stages at 128 x 128 input, batch 64 (one camera encoder) and the UNet1D levels
SHAPES = [
    ("res1 3x3 64->64   32x32", 64, 32, 32, 64, 64, 3, 1),
    ("res2 3x3 128->128 16x16", 64, 16, 16, 128, 128, 3, 1),
    ("res3 3x3 256->256  8x8", 64, 8, 8, 256, 256, 3, 1),
    ("res4 3x3 512->512  4x4", 64, 4, 4, 512, 512, 3, 1),
    ("vid 3x3 384->384 32x32", 112, 32, 32, 384, 384, 3, 1),
]
REP = 12
MAXWG = 8192
VARIANTS = [("x3", (1, 2, 4))]
if len(sys.argv) > 1 and sys.argv[1] == "video":      # the fp32 sampler's dominant shapes (B = 16 x 7 frames)
    SHAPES = [("vid 3x3 128->128 128x128", 112, 128, 128, 128, 128, 3, 1), ("vid 3x3 256->256 64x64", 112, 64, 64, 256, 256, 3, 1),
              ("vid 3x3 384->384 32x32", 112, 32, 32, 384, 384, 3, 1), ("vid t3 256->256 64x64", 16, 7, 4096, 256, 256, (3, 1), 1)]
for name, N, H, W, Ci, Co, k, s in SHAPES:
    kh, kw = (k, k) if isinstance(k, int) else k
    x = torch.randn(N, H, W, Ci, device=dev)
    w = torch.randn(Co, kh * kw * Ci, device=dev) * 0.02
    f = lambda: ops.conv2d(x, w, None, Co, kh, kw, (s, s), (kh // 2, kw // 2), defer=True)
    M = N * (H // s) * (W // s)
    fl = 2.0 * M * Co * kh * kw * Ci
    yref = None
    for vname, (on, s128, s64) in VARIANTS:
        lib.v2a_debug_f32p(on, s128, s64)
        for _ in range(3):
            y = f()[0]
        torch.cuda.synchronize()
        if yref is None:
            yref = y.clone()
        same = bool(torch.equal(y, yref))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        t_plain = sorted(ts)[len(ts) // 2]
        buf = torch.zeros(REP * MAXWG * 8, dtype=torch.int64, device=dev)
        check(lib.v2a_debug_conv_stamps(buf.data_ptr(), MAXWG * 8), "stamps")
        for _ in range(REP):
            f()
        check(lib.v2a_debug_conv_stamps(None, 0), "stamps off")
        torch.cuda.synchronize()
        st = buf.view(REP, MAXWG, 8).cpu().numpy().astype(np.float64) / 100.0       # microseconds
        nwg = int((st[REP - 1, :, 0] > 0).sum())
        st = st[:, :nwg, :]
        rows = []
        for r in range(2, REP):
            a = st[r]
            t0 = a[:, 0].min()
            prev_end = st[r - 1][:, 5].max()
            rows.append([t0 - prev_end, np.median(a[:, 1] - a[:, 0]), np.median(a[:, 2] - a[:, 1]), np.median(a[:, 3] - a[:, 2]),
                         np.median(a[:, 4] - a[:, 3]), np.median(a[:, 5] - a[:, 4]), np.median(a[:, 5] - a[:, 0]), a[:, 5].max() - t0])
        m = np.median(np.array(rows), axis=0)
        print(f"{name:28s} {vname:18s} wgs {nwg:5d}  {t_plain:6.1f} us ({fl / t_plain / 1e6:6.1f} TF, floor {fl / 157.3e6:5.1f}) bit-equal {same} | gap {m[0]:5.2f} "
              f"decode+issue {m[1]:5.2f} first-tile {m[2]:5.2f} loop {m[3]:6.2f} epilogue {m[4]:5.2f} store-ack {m[5]:5.2f} | wg life {m[6]:6.2f} span {m[7]:6.2f}",
              flush=True)
lib.v2a_debug_f32p(1, 2, 4)
