"""A/B of conv_smallk (csrc/igemm.hip) against the tile kernels inside ONE process on one box: the captured B = 64 policy step with
v2a_debug_set_smallk(1) and (0), alternating, fresh trainer each time.  Run on the GPU box."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
import bench
from v2a_hip._lib import lib
from v2a_hip.trainer import PolicyTrainer
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF

dev = "cuda:0"
store = bench.build_store(torch, dev, 64, seed=100)
res = {0: [], 1: []}
for rnd in range(3):
    for mode in (1, 0):
        lib.v2a_debug_set_smallk(mode)
        torch.manual_seed(0)
        pol = build_policy(DEFAULT_CONF).to(dev)
        tr = PolicyTrainer(pol, store, batch_size=64, seed=0, use_graph=True)
        for _ in range(5):
            tr.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            tr.step()
        e1.record()
        torch.cuda.synchronize()
        res[mode].append(e0.elapsed_time(e1) / 40)
        del tr, pol
        torch.cuda.empty_cache()
lib.v2a_debug_set_smallk(1)
print("direct small-K kernel on :", " ".join(f"{v:.3f}" for v in res[1]), "ms per step")
print("direct small-K kernel off:", " ".join(f"{v:.3f}" for v in res[0]), "ms per step")
