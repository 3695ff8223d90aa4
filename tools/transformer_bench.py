"""Forward + backward time of the Transformer policy backbone at TransformerNet's trunk size (GPU box)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
from flowdiffusion.flowdiffusion.diffusion_policy_baseline.transformer_for_diffusion import TransformerForDiffusion

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
m = TransformerForDiffusion(input_dim=4, output_dim=4, horizon=10, n_obs_steps=3, cond_dim=512, n_cond_layers=2, n_layer=8, n_head=8, n_emb=384,
                            causal_attn=True, time_as_cond=True, obs_as_cond=True).to("cuda:0").train()
x, c = torch.randn(B, 10, 4, device="cuda:0"), torch.randn(B, 3, 512, device="cuda:0")
t = torch.randint(0, 100, (B,), device="cuda:0")
opt = m.configure_optimizers()


def step():
    opt.zero_grad(set_to_none=True)
    loss = (m(x, t, c) ** 2).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print({"batch": B, "ms_per_step": dt * 1e3, "params": sum(p.numel() for p in m.parameters()), "loss": float(loss)})
