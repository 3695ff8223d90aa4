"""The captured B = 64 policy step with each of round 6's kernel routes switched off in turn (v2a_debug_set_parity_classes /
_maps_kernel / _smallk), alternating configurations inside ONE process on one box, fresh trainer each time.  Run on the GPU box."""
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
CONFIGS = {"all on": (1, 1, 1), "no parity classes": (0, 1, 1), "no maps kernel": (1, 0, 1), "no small-K kernel": (1, 1, 0), "all off (round 5 routes)": (0, 0, 0)}
res = {k: [] for k in CONFIGS}
for rnd in range(3):
    for name, (pc, mp, sk) in CONFIGS.items():
        lib.v2a_debug_set_parity_classes(pc)
        lib.v2a_debug_set_maps_kernel(mp)
        lib.v2a_debug_set_smallk(sk)
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
        res[name].append(e0.elapsed_time(e1) / 40)
        del tr, pol
        torch.cuda.empty_cache()
lib.v2a_debug_set_parity_classes(1); lib.v2a_debug_set_maps_kernel(1); lib.v2a_debug_set_smallk(1)
for name, v in res.items():
    print(f"{name:28s}", " ".join(f"{x:.3f}" for x in v), "ms per step")
