"""Race screen for the bf16 sampler kernels (GPU box): the full B=16 / 50-step sampler twice from the same noise; any DMA-landing or
LDS-publication assumption that fails occasionally shows up as a bitwise difference between the two runs."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
import v2a_hip
from flowdiffusion.flowdiffusion.unet import Unet_Libero
from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion

v2a_hip.set_video_storage(sys.argv[2] if len(sys.argv) > 2 else "bf16")      # usage: sampler_stress.py [steps] [bf16|fp16|f32]
dev = "cuda:0"
torch.manual_seed(0)
m = Unet_Libero().to(dev).eval()
d = GoalGaussianDiffusion(m, image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=int(sys.argv[1]) if len(sys.argv) > 1 else 50,
                          loss_type="l2", objective="pred_v", beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
x_cond = torch.rand(16, 3, 128, 128, device=dev, generator=g)
te = torch.randn(16, 10, 512, device=dev, generator=g)
gen = torch.Generator().manual_seed(7)
steps = d.sampling_timesteps
nz = [torch.randn(16, 21, 128, 128, generator=gen) for _ in range(steps + 1)]
outs = []
for rep in range(3):
    it = iter(nz)
    d.__dict__["_noise_hook"] = lambda shape: next(it)[:shape[0]]
    outs.append(d.sample(x_cond, te, batch_size=16).clone())
    torch.cuda.synchronize()
print("checksum", int(outs[0].view(torch.int32).to(torch.int64).sum().item()), "finite", bool(torch.isfinite(outs[0]).all()), "range", float(outs[0].min()), float(outs[0].max()))
print("run0 == run1", bool(torch.equal(outs[0], outs[1])), "run0 == run2", bool(torch.equal(outs[0], outs[2])),
      "max diff", float((outs[0] - outs[1]).abs().max()), float((outs[0] - outs[2]).abs().max()))
