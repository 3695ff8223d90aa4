"""Which call sites of one bf16-MFMA policy train step still launch an fp32 -> bf16 cast (no twin left by the producer)?"""
import collections
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
import bench
import v2a_hip
from v2a_hip import ops

v2a_hip.set_precision("bf16")
dev = "cuda:0"
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from v2a_hip.trainer import PolicyTrainer
torch.manual_seed(0)
pol = build_policy(DEFAULT_CONF).to(dev)
store = bench.build_store(torch, dev, 64, 0)
tr = PolicyTrainer(pol, store, batch_size=64, use_graph=False) if "use_graph" in PolicyTrainer.__init__.__code__.co_varnames else PolicyTrainer(pol, store, batch_size=64)
for _ in range(2):
    tr.step()
sites = collections.Counter()
orig = ops.cast_h


def spy(x, *a, **k):
    st = traceback.extract_stack()[:-1]
    loc = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-4:])
    sites[(loc, tuple(x.shape))] += 1
    return orig(x, *a, **k)


ops.cast_h = spy
try:
    tr._graph = None
except Exception:
    pass
os.environ["V2A_NO_GRAPH"] = "1"
tr.step()
torch.cuda.synchronize()
for (loc, shp), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(n, shp, loc)
print("total casts", sum(sites.values()))
