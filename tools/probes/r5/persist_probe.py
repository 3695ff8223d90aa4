"""Persistent predict_action (csrc/policy_persist.hip) against the layer-by-layer path: difference and latency.  argv: [nwg ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from oracle.param_fill import fill_module
from v2a_hip.inference import GraphedPredictAction


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


torch.manual_seed(0)
pol = build_policy(DEFAULT_CONF)
fill_module(pol, seed=13)
pol = pol.to("cuda:0").eval()
for B in (1, 2):
    for ddim in (True, False):
        g = torch.Generator().manual_seed(5)
        obs = {k: torch.rand(B, 1, 3, 128, 128, generator=g).cuda() for k in pol._cfg.rgb_keys}
        init = torch.randn(B, 16, 7, generator=g).cuda()
        n = 8 if ddim else 100
        noises = None if ddim else [torch.randn(B, 16, 7, generator=g).cuda() for _ in range(n)]
        ref = GraphedPredictAction(pol, B, use_ddim=ddim, persistent=False)
        per = GraphedPredictAction(pol, B, use_ddim=ddim, persistent=True)
        a = ref(obs, init_noise=init, step_noises=noises)["action_pred"].clone()
        b = per(obs, init_noise=init, step_noises=noises)["action_pred"].clone()
        print(f"B {B} {'ddim8' if ddim else 'ddpm100'}: rel diff {rel(b, a):.3e}   ops {per.pp.n_ops} barriers {per.pp.n_barriers} lds {per.pp.lds}", flush=True)
        for name, gp in (("layers", ref), ("persistent", per)):
            for _ in range(3):
                gp(obs, init_noise=init, step_noises=noises)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gp.graph.replay()
            e1.record()
            torch.cuda.synchronize()
            print(f"    {name}: {e0.elapsed_time(e1) / 10:.3f} ms per call", flush=True)
for nwg in [int(a) for a in sys.argv[1:]]:
    from v2a_hip.policy_persist import PersistentDenoiser
    g = torch.Generator().manual_seed(5)
    init = torch.randn(1, 16, 7, generator=g).cuda()
    gc = torch.randn(1, pol.engine.film_gd - pol._cfg.dsed if hasattr(pol._cfg, "dsed") else 128, generator=g).cuda()
    pd = PersistentDenoiser(pol.engine, 1, [87, 75, 62, 50, 37, 25, 12, 0], True, 8, init.clone(), nwg=nwg)
    gc = torch.randn(1, pd.gcond.shape[1], generator=g).cuda()
    for _ in range(3):
        pd.launch(gc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pd.launch(gc)
    e1.record()
    torch.cuda.synchronize()
    print(f"nwg {nwg}: denoiser alone {e0.elapsed_time(e1) / 10:.3f} ms ({pd.n_barriers} barriers)", flush=True)
