"""Does the stem's forward kernel (channel-window three-plane conv vs scalar-gather exact-f32 conv) flip ReLU / max-pool decisions on the golden batch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
from test_policy_gpu import _policy, _batch
from v2a_hip import ops, policy_engine as PE
g = np.load(f"{ROOT}/tests/golden/policy.npz", allow_pickle=True)
noise, ts = torch.from_numpy(g["noise"]), torch.from_numpy(g["timesteps"])
res = {}
o_wg = PE.PolicyEngine._wg
o_mp = ops.maxpool_fwd
cur = [None]
def wg(self, *a, **k):
    if k.get("immediate"):
        res.setdefault(cur[0], {}).setdefault("dc1", []).append(a[1].clone())
    return o_wg(self, *a, **k)
def mp(x):
    h, idx = o_mp(x)
    res.setdefault(cur[0], {}).setdefault("pidx", []).append(idx.clone())
    res.setdefault(cur[0], {}).setdefault("a1", []).append(x.clone())
    return h, idx
PE.PolicyEngine._wg = wg
ops.maxpool_fwd = mp
for win in (True, False):
    cur[0] = win
    PE._STEM_WINDOW = win
    pol, sd = _policy()
    pol.__dict__["_rng_hook"] = lambda shape, kind: {"noise": noise, "timesteps": ts}[kind]
    pol.train()
    loss = pol.compute_loss(_batch(g)); loss.backward()
    torch.cuda.synchronize()
for i, nm in enumerate(("first encoder", "second encoder")):
    a, b = res[True], res[False]
    print(nm, "| pool argmax differs at", int((a["pidx"][i] != b["pidx"][i]).sum()), "of", a["pidx"][i].numel(),
          "| relu mask differs at", int(((a["a1"][i] > 0) != (b["a1"][i] > 0)).sum()),
          "| a1 max diff", float((a["a1"][i] - b["a1"][i]).abs().max()),
          "| dc1 max diff / max", float((a["dc1"][i] - b["dc1"][i]).abs().max() / b["dc1"][i].abs().max()))
