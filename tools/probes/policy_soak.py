"""Soak: N graph-replayed policy train steps twice from the same seeds; losses finite, final parameters bitwise equal, device memory flat."""
import os
import random
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import numpy as np
import torch
import bench
import v2a_hip
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from v2a_hip.trainer import PolicyTrainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
v2a_hip.set_precision(sys.argv[2] if len(sys.argv) > 2 else "fp32")
runs = []
for rep in range(2):
    torch.manual_seed(0)
    np.random.seed(1)
    random.seed(1)
    pol = build_policy(DEFAULT_CONF).to("cuda:0")
    store = bench.build_store(torch, "cuda:0", 64, 0)
    tr = PolicyTrainer(pol, store, batch_size=64, seed=3)
    mem0 = None
    losses = []
    for i in range(N):
        l = tr.step()
        if i % 100 == 99:
            losses.append(l.item())
            m = torch.cuda.memory_allocated()
            mem0 = mem0 or m
            assert m == mem0, (i, m, mem0)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for p in pol.parameters()]).cpu()
    runs.append((losses, flat))
    print("run", rep, "losses", [round(x, 5) for x in losses[:3]], "...", round(losses[-1], 5), "finite", bool(np.isfinite(losses).all()), flush=True)
    del tr, pol, store
    torch.cuda.empty_cache()
print("losses equal", runs[0][0] == runs[1][0], "parameters bitwise equal", bool(torch.equal(runs[0][1], runs[1][1])))
