"""One scheduler step of the persistent denoiser against torch.nn.functional, tensor by tensor (B = 1)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from oracle.param_fill import fill_module
from v2a_hip.policy_persist import PersistentDenoiser
import math

torch.manual_seed(0)
pol = build_policy(DEFAULT_CONF)
fill_module(pol, seed=13)
pol = pol.to("cuda:0").eval()
eng = pol.engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator().manual_seed(5)
traj0 = torch.randn(B, 16, 7, generator=g).cuda()
traj = traj0.clone()
t = 87
pd = PersistentDenoiser(eng, B, [t], True, 8, traj0)
traj = pd.traj
gc = torch.randn(B, pd.gcond.shape[1], generator=g).cuda()
pd.launch(gc)
torch.cuda.synchronize()
P = {k: v.double() for k, v in eng.P.items()}


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def show(name, got, want):            # got [B,T,C] channels-last, want [B,C,T]
    w = want.transpose(1, 2) if want.dim() == 3 else want
    print(f"{name:46s} {rel(got, w):.3e}   shape {tuple(got.shape)}", flush=True)


dsed = eng.cfg.dsed
half = dsed // 2
freq = torch.exp(torch.arange(half, dtype=torch.float64, device="cuda") * -(math.log(10000.0) / (half - 1)))
emb = torch.cat([torch.sin(t * freq), torch.cos(t * freq)])[None].repeat(B, 1)
m = "model."
e1 = F.linear(emb, P[m + "diffusion_step_encoder.1.weight"], P[m + "diffusion_step_encoder.1.bias"])
show("e1", pd.named["e1"], e1)
e2 = F.linear(F.mish(e1), P[m + "diffusion_step_encoder.3.weight"], P[m + "diffusion_step_encoder.3.bias"])
show("e2", pd.named["e2"], e2)
cond = torch.cat([e2, gc.double()], 1)


def block(r, x, j):
    pre = r["pre"]
    film = F.linear(F.mish(cond), P[pre + ".cond_encoder.1.weight"], P[pre + ".cond_encoder.1.bias"])
    show(pre + ".film", pd.named["film"][j], film)
    k = eng.cfg.kernel_size
    raw0 = F.conv1d(x, P[pre + ".blocks.0.block.0.weight"], P[pre + ".blocks.0.block.0.bias"], padding=k // 2)
    show(pre + ".raw0", pd.named[pre + ".raw0"], raw0)
    co = raw0.shape[1]
    h = F.mish(F.group_norm(raw0, 8, P[pre + ".blocks.0.block.1.weight"], P[pre + ".blocks.0.block.1.bias"]))
    h = film[:, :co, None] * h + film[:, co:, None]
    raw1 = F.conv1d(h, P[pre + ".blocks.1.block.0.weight"], P[pre + ".blocks.1.block.0.bias"], padding=k // 2)
    show(pre + ".raw1", pd.named[pre + ".raw1"], raw1)
    h = F.mish(F.group_norm(raw1, 8, P[pre + ".blocks.1.block.1.weight"], P[pre + ".blocks.1.block.1.bias"]))
    if r["rc"] is not None:
        res = F.conv1d(x, P[pre + ".residual_conv.weight"], P[pre + ".residual_conv.bias"])
        show(pre + ".res", pd.named[pre + ".res"], res)
    else:
        res = x
    out = h + res
    show(pre + ".out (stored by its first consumer)", pd.named[pre + ".out"], out)
    return out


x = traj0.double().transpose(1, 2)
hs = []
j = 0
for i, lvl in enumerate(eng.down):
    x = block(lvl["r0"], x, j); j += 1
    x = block(lvl["r1"], x, j); j += 1
    hs.append(x)
    if lvl["ds"] is not None:
        x = F.conv1d(x, P[lvl["ds"].wname], P[lvl["ds"].bname], stride=2, padding=1)
        show(f"down{i}.ds", pd.named[f"down{i}.ds"], x)
for r in eng.mid:
    x = block(r, x, j); j += 1
for i, lvl in enumerate(eng.up):
    x = torch.cat([x, hs.pop()], 1)
    x = block(lvl["r0"], x, j); j += 1
    x = block(lvl["r1"], x, j); j += 1
    x = F.conv_transpose1d(x, P[lvl["us"].wname], P[lvl["us"].bname], stride=2, padding=1)
    show(f"up{i}.us", pd.named[f"up{i}.us"], x)
raw = F.conv1d(x, P[eng.fin0.wname], P[eng.fin0.bname], padding=2)
show("final.raw", pd.named["final.raw"], raw)
h = F.mish(F.group_norm(raw, 8, P[m + "final_conv.0.block.1.weight"], P[m + "final_conv.0.block.1.bias"]))
eps = F.conv1d(h, P[eng.fin1.wname], P[eng.fin1.bname]).transpose(1, 2)
from v2a_hip.policy_sched import ddim_coeffs
c = ddim_coeffs(eng.ac_host, t, 100, 8)
x0 = ((traj0.double() - c[0] * eps) / c[1]).clamp(-1, 1)
want = c[2] * x0 + c[3] * eps
print(f"{'trajectory after the step':46s} {rel(traj, want):.3e}")
