import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
from test_policy_gpu import _policy, _batch
from v2a_hip import ops, policy_engine as PE
g = np.load(f"{ROOT}/tests/golden/policy.npz", allow_pickle=True)
mode = sys.argv[1]
pol, sd = _policy()
batch = _batch(g)
noise, ts = torch.from_numpy(g["noise"]), torch.from_numpy(g["timesteps"])
pol.__dict__["_rng_hook"] = lambda shape, kind: {"noise": noise, "timesteps": ts}[kind]
pol.train()
orig = PE.PolicyEngine._wg
saved = []
def wg(self, *a, **k):
    if k.get("immediate"):
        if mode == "before": torch.cuda.synchronize()
        saved.append((a[0].clone(), a[1].clone(), k["dw"], a))
    r = orig(self, *a, **k)
    if k.get("immediate"):
        if mode == "after": torch.cuda.synchronize()
        saved[-1] = saved[-1] + (k["dw"].clone(),)
    return r
PE.PolicyEngine._wg = wg
loss = pol.compute_loss(batch); loss.backward()
torch.cuda.synchronize()
P = dict(pol.named_parameters())
names = ["obs_encoder.key_model_map.img_obs_1.backbone.nets.0.weight", "obs_encoder.key_model_map.img_goal_1.backbone.nets.0.weight"]
for (x, dy, dw, a, dwc) in saved:
  for n in names:
    with torch.enable_grad():
        wd = torch.zeros(tuple(a[2]), dtype=torch.float64, device=x.device, requires_grad=True)
        F.conv2d(x.permute(0, 3, 1, 2).double(), wd, stride=a[5], padding=a[6]).backward(dy.permute(0, 3, 1, 2).double())
    ref = wd.grad
    sc = ref.abs().max().item()
    d = (P[n].grad.double() - ref[:, :3]).abs()
    print("   worst element", np.unravel_index(int(d.argmax()), d.shape), "count > 1e-5 sc", int((d > 1e-5 * sc).sum()))
    print(mode, n.split(".")[3], "dw4 right after the call", f"{(dwc.double() - ref).abs().max().item() / sc:.3e}",
          "| param.grad at the end", f"{(P[n].grad.double() - ref[:, :3]).abs().max().item() / sc:.3e}")
