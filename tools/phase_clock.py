"""REAL phase timeline of the captured policy train step (no profiler: rocprofv3's kernel trace serialises the hardware queues).
V2A_TSTAMP=1 makes the engine drop wall-clock probes (one-lane kernels) at its phase boundaries, on whichever stream reaches them.
Usage (GPU box): V2A_TSTAMP=1 python tools/phase_clock.py [fp32|bf16] [batch]"""
import os
import sys
os.environ.setdefault("V2A_TSTAMP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import time
import numpy as np
import random
import torch
import v2a_hip
from v2a_hip import ops
from v2a_hip.trainer import PolicyTrainer
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
v2a_hip.set_precision(prec)
torch.manual_seed(0); np.random.seed(0); random.seed(0)
dev = "cuda:0"
pol = build_policy(DEFAULT_CONF).to(dev)
store = bench.build_store(torch, dev, B, seed=100)
tr = PolicyTrainer(pol, store, batch_size=B, seed=0, use_graph=True)
for _ in range(6):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    tr.step()
torch.cuda.synchronize()
print(f"{prec} B={B}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step (with probes)")
tab = sorted(ops.tstamp_table(), key=lambda kv: kv[1])
for n, t in tab:
    print(f"{t:10.1f} us  {n}")
