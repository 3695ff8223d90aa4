"""A/B of a temporary integer hook exported by libv2a_hip.so (argv[1] = symbol, argv[2:] = values): device time per launch (hipGraph replay)
of the ConditionalUnet1D conv shapes, then the whole policy step, for every value.  Usage: python tools/probes/r5/ab_hook_probe.py v2a_tmp_x3_pf 2 4"""
import ctypes, os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import numpy as np
import torch
from v2a_hip import ops, _lib
import v2a_hip
dll = ctypes.CDLL(_lib.LIB_PATH)
hook = getattr(dll, sys.argv[1])
vals = [int(v) for v in sys.argv[2:]]
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


SHAPES = [("l0 256->256 T16", 16, 256, 256, 5), ("l1 256->512 T8", 8, 256, 512, 5), ("l1 512->512 T8", 8, 512, 512, 5),
          ("l2 512->1024 T4", 4, 512, 1024, 5), ("l2 1024->1024 T4", 4, 1024, 1024, 5), ("up 2048->512 T4", 4, 2048, 512, 5),
          ("up 1024->256 T8", 8, 1024, 256, 5), ("rc 2048->512 T4 1x1", 4, 2048, 512, 1), ("enc 1x1 s2 64x64x64->128", -1, 64, 128, 1)]
for v in vals:
    hook(v)
    print(f"--- {sys.argv[1]} = {v}")
    for name, T, Ci, Co, k in SHAPES:
        if T > 0:
            x = torch.randn(64, 1, T, Ci, device=dev); w = torch.randn(Co, k * Ci, device=dev) * 0.02
            f = lambda: ops.conv2d(x, w, None, Co, 1, k, (1, 1), (0, k // 2), defer=True)
            fl = 2.0 * 64 * T * Co * k * Ci
        else:
            x = torch.randn(64, 32, 32, Ci, device=dev); w = torch.randn(Co, Ci, device=dev) * 0.02
            f = lambda: ops.conv2d(x, w, None, Co, 1, 1, (2, 2), (0, 0), defer=True)
            fl = 2.0 * 64 * 16 * 16 * Co * Ci
        t = timeit(f)
        print(f"{name:28s} {t:6.1f} us  {fl / t / 1e6:6.1f} TF", flush=True)

import bench
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from v2a_hip.trainer import PolicyTrainer
for v in vals + vals:
    hook(v)
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    pol = build_policy(DEFAULT_CONF).to(dev)
    store = bench.build_store(torch, dev, 64, seed=100)
    tr = PolicyTrainer(pol, store, batch_size=64, seed=0, use_graph=True)
    for _ in range(6): tr.step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30): tr.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"{sys.argv[1]} = {v}: policy step {min(ts):.3f} ms (runs {['%.3f' % t for t in ts]}) loss {float(tr.loss.item()):.5f}", flush=True)
    del tr, pol, store
    torch.cuda.empty_cache()
