"""Sampler timing probe (GPU box): graph vs eager, B=16 / B=1, DDIM-50 / DDPM-100, f32 / bf16 storage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
import v2a_hip
from flowdiffusion.flowdiffusion.unet import Unet_Libero
from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion

dev = "cuda:0"
torch.manual_seed(0)
unet = Unet_Libero().to(dev).eval()
cfgs = [a.split(":") for a in sys.argv[1:]] or [["bf16", "16", "50"], ["bf16", "1", "100"], ["bf16", "1", "50"]]
for storage, B, steps in cfgs:
    B, steps = int(B), int(steps)
    v2a_hip.set_video_storage(storage)
    unet.__dict__.pop("_eng", None)
    d = GoalGaussianDiffusion(unet, image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=steps, loss_type="l2",
                              objective="pred_v", beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x_cond = torch.rand(B, 3, 128, 128, device=dev, generator=g)
    te = torch.randn(B, 10, 512, device=dev, generator=g)
    for use_graph in (True, False):
        d.__dict__["_use_graph"] = use_graph
        torch.manual_seed(5)
        out = d.sample(x_cond, te, batch_size=B)      # warm-up / capture
        ts = []
        for _ in range(3):
            torch.manual_seed(5)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            out2 = d.sample(x_cond, te, batch_size=B)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"{storage} B={B} steps={steps} graph={use_graph}: median {ts[1]:.1f} ms  ({B * 7 / ts[1] * 1e3:.2f} frames/s, {ts[1] / steps:.2f} ms/step) "
              f"same={torch.equal(out, out2)} chk={float(out2.double().sum()):.6f}", flush=True)
