import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
from test_policy_gpu import _policy, _batch
from v2a_hip import ops, policy_engine as PE
g = np.load(f"{ROOT}/tests/golden/policy.npz", allow_pickle=True)
pol, sd = _policy()
batch = _batch(g)
noise, ts = torch.from_numpy(g["noise"]), torch.from_numpy(g["timesteps"])
pol.__dict__["_rng_hook"] = lambda shape, kind: {"noise": noise, "timesteps": ts}[kind]
pol.train()
true = {"img_obs_1": torch.from_numpy(g["img_obs"]).reshape(-1, 3, 128, 128).cuda() * 2 - 1,
        "img_goal_1": torch.from_numpy(g["img_goal"]).reshape(-1, 3, 128, 128).cuda() * 2 - 1}
o_stem = PE.PolicyEngine._stem_fwd
def stem(self, key, img, conv1, w0):
    x0, c1 = o_stem(self, key, img, conv1, w0)
    t = (img.float() * 2 - 1)
    print("fwd", key, "img given vs golden", float((t - true[key]).abs().max()), "| buffer interior vs given",
          float((x0[:, 3:-3, 3:-3, :3].permute(0, 3, 1, 2) - t).abs().max()), "img dtype", img.dtype, tuple(img.shape), flush=True)
    wref = F.conv2d(t.double(), conv1.w.detach().double(), stride=2, padding=3).permute(0, 2, 3, 1)
    print("    conv out rel err", float((c1.double() - wref).abs().max() / wref.abs().max()))
    return x0, c1
PE.PolicyEngine._stem_fwd = stem
o_wg = PE.PolicyEngine._wg
def wg(self, *a, **k):
    if k.get("immediate"):
        x = a[0]
        for key, t in true.items():
            print("bwd: stem buffer interior vs golden", key, float((x[:, 3:-3, 3:-3, :3].permute(0, 3, 1, 2) - t).abs().max()), flush=True)
    return o_wg(self, *a, **k)
PE.PolicyEngine._wg = wg
loss = pol.compute_loss(batch); loss.backward()
torch.cuda.synchronize()
