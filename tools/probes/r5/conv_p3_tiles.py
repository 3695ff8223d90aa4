"""Tile-shape / split / stage A/B of conv_p3 through the temporary plan hook v2a_tmp_p3_plan (bm, bn, splits, stages): device time per launch
(hipGraph replay) on the ConditionalUnet1D shapes, result checked against conv_igemm_f32x3.  Usage: python tools/probes/r5/conv_p3_tiles.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops, _lib
from v2a_hip._lib import lib, check
dll = ctypes.CDLL(_lib.LIB_PATH)
dev = "cuda:0"
WS = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


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


def tiled(w3, Co, K, bn):
    """[3, Co, K] bf16 planes -> [Co / bn][K / 32][3][bn rows][4 positions][8]: position p of row r holds chunk p ^ ((r >> 2) & 3)"""
    w = w3.view(3, Co // bn, bn, K // 32, 4, 8)
    r = torch.arange(bn, device=w.device)
    pos = torch.arange(4, device=w.device)
    src = (pos[None, :] ^ ((r[:, None] >> 2) & 3))                      # [bn, 4] chunk index stored at (r, p)
    w = w.permute(1, 3, 0, 2, 4, 5).contiguous()                        # [Co/bn, K/32, 3, bn, 4, 8]
    idx = src.view(1, 1, 1, bn, 4, 1).expand(w.shape[0], w.shape[1], 3, bn, 4, 8)
    return torch.gather(w, 4, idx).contiguous()


def p3(x3, w3, Co, k, y, T, C1, wstride=None):
    ns = ctypes.c_int(0)
    check(lib.v2a_conv2d_fwd_p3(x3.data_ptr(), x3.numel() // 3, None, 0, w3.data_ptr(), (w3.numel() // 3) if wstride is None else wstride, None, None, y.data_ptr(),
                                ops._zero_line(dev).data_ptr(), 64, 1, T, C1, 0, Co, 1, k, 1, 1, 0, k // 2, 1, T, ctypes.byref(ns),
                                WS.data_ptr(), WS.numel(), ops._stream()), "p3")
    return ns.value


SHAPES = [("256->256 T16", 16, 256, 256, 5), ("512->512 T8", 8, 512, 512, 5), ("1024->1024 T4", 4, 1024, 1024, 5), ("2048->512 T4", 4, 2048, 512, 5),
          ("1024->256 T8", 8, 1024, 256, 5), ("256->512 T8", 8, 256, 512, 5)]
for name, T, C1, Co, k in SHAPES:
    g = torch.Generator().manual_seed(T * 1000 + Co)
    x = torch.randn(64, 1, T, C1, generator=g).to(dev)
    w = (torch.randn(Co, k * C1, generator=g) * 0.02).to(dev)
    x3, w3 = ops.split3(x), ops.split3(w).view(3, -1)
    M, K = 64 * T, k * C1
    y0 = ops.conv2d(x, w, None, Co, 1, k, (1, 1), (0, k // 2))
    t_ref = timeit(lambda: ops.conv2d(x, w, None, Co, 1, k, (1, 1), (0, k // 2), defer=True))
    y = torch.empty_like(y0)
    fl = 2.0 * M * Co * K
    print(f"--- {name}: M={M} N={Co} K={K}  x3 register kernel {t_ref:.1f} us ({fl / t_ref / 1e6:.0f} TF)")
    nkt = K // 32
    for bm, bn, st, targets in ((64, 64, 3, (512,)), (64, 64, 2, (512,)), (64, 128, 3, (256,)), (128, 128, 8, (256,))):
        if bm > M:
            continue
        tiles = -(-M // bm) * -(-Co // bn)
        for target in targets:
            s = max(1, min(target // tiles, nkt // 3, 32))
            dll.v2a_tmp_p3_plan(bm, bn, s, st)
            ns = p3(x3, w3, Co, k, y, T, C1)
            if ns > 0:
                y.copy_(WS[:ns * M * Co * 4].view(torch.float32).view(ns, M * Co).sum(0).view(y.shape))
            torch.cuda.synchronize()
            err = float((y - y0).abs().max() / y0.abs().max())
            t = timeit(lambda: p3(x3, w3, Co, k, y, T, C1))
            wt = tiled(w3, Co, K, bn)
            y.zero_()
            ns2 = p3(x3, wt, Co, k, y, T, C1, wstride=0)
            if ns2 > 0:
                y.copy_(WS[:ns2 * M * Co * 4].view(torch.float32).view(ns2, M * Co).sum(0).view(y.shape))
            torch.cuda.synchronize()
            err2 = float((y - y0).abs().max() / y0.abs().max())
            t2 = timeit(lambda: p3(x3, wt, Co, k, y, T, C1, wstride=0))
            print(f"   tile {bm:3d}x{bn:3d} variant {st} splits {ns or 1:2d} wgs {tiles * max(ns, 1):4d} kt/wg {-(-nkt // max(ns, 1)):3d}: {t:6.1f} us "
                  f"({fl / t / 1e6:5.0f} TF)  rel err {err:.1e} | tiled weights {t2:6.1f} us ({fl / t2 / 1e6:5.0f} TF) err {err2:.1e}", flush=True)
dll.v2a_tmp_p3_plan(0, 0, 1, 3)
