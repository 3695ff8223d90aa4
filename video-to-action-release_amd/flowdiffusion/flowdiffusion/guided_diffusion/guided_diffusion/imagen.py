"""Parameter shell of the text-token pooler (reference guided_diffusion/imagen.py:198-211, 254-372, 1009-1017).
Only PerceiverResampler and its helpers are on the hot path; the imagen-video Unet3D stack of that file is dead code."""
import torch
import torch.nn as nn


class LayerNorm(nn.Module):
    def __init__(self, dim, stable=False):
        super().__init__()
        self.stable = stable
        self.g = nn.Parameter(torch.ones(dim))


def FeedForward(dim, mult=2):
    hidden = int(dim * mult)
    return nn.Sequential(LayerNorm(dim), nn.Linear(dim, hidden, bias=False), nn.GELU(), LayerNorm(hidden), nn.Linear(hidden, dim, bias=False))


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8, scale=8):
        super().__init__()
        self.scale = scale
        self.heads = heads
        inner = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), nn.LayerNorm(dim))


class PerceiverResampler(nn.Module):
    def __init__(self, *, dim, depth, dim_head=64, heads=8, num_latents=64, num_latents_mean_pooled=4, max_seq_len=512, ff_mult=4):
        super().__init__()
        self.pos_emb = nn.Embedding(max_seq_len, dim)
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.to_latents_from_mean_pooled_seq = None
        if num_latents_mean_pooled > 0:
            self.to_latents_from_mean_pooled_seq = nn.Sequential(LayerNorm(dim), nn.Linear(dim, dim * num_latents_mean_pooled), nn.Identity())
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads), FeedForward(dim=dim, mult=ff_mult)]))
        self.cfg = dict(depth=depth, dim_head=dim_head, heads=heads, num_latents=num_latents, num_mean_pooled=num_latents_mean_pooled,
                        ff_mult=ff_mult)
