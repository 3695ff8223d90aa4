"""Which gradient tensors of the policy differ from the CPU oracle (the comparison of tests/test_policy_gpu.py, per parameter)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_policy_gpu import _policy, _batch
from oracle import policy as OP
g = np.load(f"{ROOT}/tests/golden/policy.npz", allow_pickle=True)
pol, sd = _policy()
batch = _batch(g)
print("img shape", batch["obs"]["img_obs_1"].shape, batch["obs"]["img_obs_1"].dtype)
noise, ts = torch.from_numpy(g["noise"]), torch.from_numpy(g["timesteps"])
pol.__dict__["_rng_hook"] = lambda shape, kind: {"noise": noise, "timesteps": ts}[kind]
pol.train()
loss = pol.compute_loss(batch); loss.backward()
names = [str(n) for n in g["param_names"]]
P = dict(pol.named_parameters())
gsc = float(np.max(g["grad_norms"]))
l2, og = OP.loss_and_grads(sd, batch, noise, ts, names=names)
errs = sorted(((P[n].grad.double().cpu() - og[n].double()).abs().max().item() / max(og[n].abs().max().item(), 1e-3 * gsc), n) for n in names)
for e, n in errs[-8:]:
    print(f"{e:.3e} {n} {tuple(P[n].shape)}")
n = "obs_encoder.key_model_map.img_goal_1.backbone.nets.0.weight"
d = (P[n].grad.double().cpu() - og[n].double()).abs()
sc = og[n].abs().max().item()
print("own scale", sc, "gsc*1e-3", 1e-3 * gsc, "max diff", d.max().item())
print("per kh max diff / sc:", [f"{d[:, :, i].max().item() / sc:.1e}" for i in range(7)])
print("per kw max diff / sc:", [f"{d[:, :, :, i].max().item() / sc:.1e}" for i in range(7)])
print("per ci max diff / sc:", [f"{d[:, i].max().item() / sc:.1e}" for i in range(3)])
n2 = "obs_encoder.key_model_map.img_obs_1.backbone.nets.0.weight"
d2 = (P[n2].grad.double().cpu() - og[n2].double()).abs()
print("obs: own scale", og[n2].abs().max().item(), "max diff", d2.max().item())
