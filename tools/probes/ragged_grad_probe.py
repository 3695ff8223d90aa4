"""Which parameter gradients of a ragged-batch policy step disagree with the CPU oracle?  usage: ragged_grad_probe.py B [B ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from oracle import policy as OP
from oracle.param_fill import fill_module
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF

"""usage: ragged_grad_probe.py B[:seed] ...   (seed defaults to 100 + B; the test uses the same generator recipe)"""
for arg in sys.argv[1:] or ["3"]:
    B, _, sdg = arg.partition(":")
    B = int(B)
    sdg = int(sdg) if sdg else 100 + B
    torch.manual_seed(0)
    pol = build_policy(DEFAULT_CONF)
    sd = fill_module(pol, seed=21 + B)
    pol = pol.to("cuda:0")
    g = torch.Generator().manual_seed(sdg)
    batch = {"obs": {"img_obs_1": torch.rand(B, 1, 3, 128, 128, generator=g), "img_goal_1": torch.rand(B, 1, 3, 128, 128, generator=g)},
             "action": torch.rand(B, 16, 7, generator=g) * 2 - 1}
    noise, ts = torch.randn(B, 16, 7, generator=g), torch.randint(0, 100, (B,), generator=g)
    pol.__dict__["_rng_hook"] = lambda shape, kind: {"noise": noise, "timesteps": ts}[kind]
    pol.train()
    loss = pol.compute_loss(batch)
    loss.backward()
    names = pol.trainable_names()
    ref_loss, ref_g = OP.loss_and_grads(sd, batch, noise, ts, names=names)
    P = dict(pol.named_parameters())
    gsc = max(float(v.double().norm()) for v in ref_g.values())
    errs = sorted(((((P[n].grad.double().cpu() - ref_g[n].double()).abs().max() / max(ref_g[n].abs().max().item(), 1e-3 * gsc)).item(), n,
                    tuple(ref_g[n].shape), ref_g[n].abs().max().item()) for n in names), reverse=True)
    # the same gradient in fp64: a ReLU input that fp32 rounding puts on the other side of zero flips one mask element, which at small B
    # is 1e-3 of a weight gradient -- if the fp32 ORACLE is as far from the fp64 one as the HIP path is, neither is "wrong"
    sd64 = {k: (v.double() if torch.is_floating_point(v) else v) for k, v in sd.items()}
    b64 = {"obs": {k: v.double() for k, v in batch["obs"].items()}, "action": batch["action"].double()}
    _, g64 = OP.loss_and_grads(sd64, b64, noise.double(), ts, names=names)
    e_hip = max(((P[n].grad.double().cpu() - g64[n]).abs().max() / max(g64[n].abs().max().item(), 1e-3 * gsc)).item() for n in names)
    e_o32 = max(((ref_g[n].double() - g64[n]).abs().max() / max(g64[n].abs().max().item(), 1e-3 * gsc)).item() for n in names)
    print(f"B={B} seed={sdg} loss {loss.item():.6f} vs {ref_loss.item():.6f}; gsc {gsc:.3e};  worst vs fp64 oracle: HIP {e_hip:.2e}, fp32 oracle {e_o32:.2e}")
    for e, n, shp, mx in errs[:10]:
        print(f"   {e:9.2e}  {n:70s} {shp} max|g| {mx:.3e}")
