"""n DDIM steps of the persistent denoiser (eager launch, no graph) against torch.nn.functional in fp64.  argv: B nsteps"""
import os, sys, math
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from oracle.param_fill import fill_module
from v2a_hip.policy_persist import PersistentDenoiser
from v2a_hip.policy_sched import ddim_coeffs, ddim_timesteps

torch.manual_seed(0)
pol = build_policy(DEFAULT_CONF)
fill_module(pol, seed=13)
pol = pol.to("cuda:0").eval()
eng = pol.engine
B, n = int(sys.argv[1]), int(sys.argv[2])
steps = ddim_timesteps(100, 8)[:n]
g = torch.Generator().manual_seed(5)
traj0 = torch.randn(B, 16, 7, generator=g).cuda()
traj = traj0.clone()
pd = PersistentDenoiser(eng, B, steps, True, 8, traj0, nwg=int(sys.argv[3]) if len(sys.argv) > 3 else None)
traj = pd.traj
gc = torch.randn(B, pd.gcond.shape[1], generator=g).cuda()
pd.launch(gc)
torch.cuda.synchronize()
P = {k: v.detach().double() for k, v in eng.P.items()}
m = "model."
dsed = eng.cfg.dsed
half = dsed // 2
freq = torch.exp(torch.arange(half, dtype=torch.float64, device="cuda") * -(math.log(10000.0) / (half - 1)))


def unet(x, t):
    emb = torch.cat([torch.sin(t * freq), torch.cos(t * freq)])[None].repeat(B, 1)
    e1 = F.linear(emb, P[m + "diffusion_step_encoder.1.weight"], P[m + "diffusion_step_encoder.1.bias"])
    e2 = F.linear(F.mish(e1), P[m + "diffusion_step_encoder.3.weight"], P[m + "diffusion_step_encoder.3.bias"])
    cond = torch.cat([e2, gc.double()], 1)

    def block(r, x):
        pre = r["pre"]
        film = F.linear(F.mish(cond), P[pre + ".cond_encoder.1.weight"], P[pre + ".cond_encoder.1.bias"])
        raw0 = F.conv1d(x, P[pre + ".blocks.0.block.0.weight"], P[pre + ".blocks.0.block.0.bias"], padding=2)
        co = raw0.shape[1]
        h = F.mish(F.group_norm(raw0, 8, P[pre + ".blocks.0.block.1.weight"], P[pre + ".blocks.0.block.1.bias"]))
        h = film[:, :co, None] * h + film[:, co:, None]
        raw1 = F.conv1d(h, P[pre + ".blocks.1.block.0.weight"], P[pre + ".blocks.1.block.0.bias"], padding=2)
        h = F.mish(F.group_norm(raw1, 8, P[pre + ".blocks.1.block.1.weight"], P[pre + ".blocks.1.block.1.bias"]))
        res = F.conv1d(x, P[pre + ".residual_conv.weight"], P[pre + ".residual_conv.bias"]) if r["rc"] is not None else x
        return h + res

    hs = []
    for lvl in eng.down:
        x = block(lvl["r1"], block(lvl["r0"], x))
        hs.append(x)
        if lvl["ds"] is not None:
            x = F.conv1d(x, P[lvl["ds"].wname], P[lvl["ds"].bname], stride=2, padding=1)
    for r in eng.mid:
        x = block(r, x)
    for lvl in eng.up:
        x = block(lvl["r1"], block(lvl["r0"], torch.cat([x, hs.pop()], 1)))
        x = F.conv_transpose1d(x, P[lvl["us"].wname], P[lvl["us"].bname], stride=2, padding=1)
    raw = F.conv1d(x, P[eng.fin0.wname], P[eng.fin0.bname], padding=2)
    h = F.mish(F.group_norm(raw, 8, P[m + "final_conv.0.block.1.weight"], P[m + "final_conv.0.block.1.bias"]))
    return F.conv1d(h, P[eng.fin1.wname], P[eng.fin1.bname]).transpose(1, 2)


x = traj0.double()
for t in steps:
    eps = unet(x.transpose(1, 2), t)
    c = ddim_coeffs(eng.ac_host, t, 100, 8)
    x0 = ((x - c[0] * eps) / c[1]).clamp(-1, 1)
    x = c[2] * x0 + c[3] * eps
d = (traj.double() - x).abs().max() / x.abs().max()
print(f"B {B}, {n} steps {steps}: trajectory rel diff {float(d):.3e}", flush=True)
tr = pd.trace(gc)
import collections
npro, nst = pd.n_ops
load = sum(r[1] - r[0] for r in tr); prod = sum(r[2] - r[1] for r in tr); bar = sum(r[3] - r[2] for r in tr)
print(f"timeline of workgroup 0: total {tr[-1][3]:.1f} us = loaders {load:.1f} + products {prod:.1f} + barriers {bar:.1f} (+ gaps); prologue {tr[npro - 1][3]:.1f} us")
print("step 1, per op (load / product / barrier us):")
for i in range(npro + nst, npro + 2 * nst) if n > 1 else range(npro, npro + nst):
    r = tr[i]
    print(f"   op {i - npro - (nst if n > 1 else 0):2d}: {r[1] - r[0]:6.1f} {r[2] - r[1]:6.1f} {r[3] - r[2]:6.1f}")
