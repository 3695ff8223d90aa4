"""CPU restatement (torch fp32) of the AVDC pseudo-3D video UNet forward.  TEST INFRASTRUCTURE.

Functional: consumes a flat {name: tensor} dict with the reference's parameter names, so the
same dict drives the reference module (tools/make_golden.py), this oracle and the HIP engine.

Follows (paths relative to /root/reference/flowdiffusion/flowdiffusion/):
  Unet_Libero.forward            unet.py:216-222
  UNetModel.__init__/forward     guided_diffusion/guided_diffusion/unet.py:435-632, 650-684
  ResBlock._forward              guided_diffusion/guided_diffusion/unet.py:239-260
  AttentionBlock._forward        guided_diffusion/guided_diffusion/unet.py:303-309
  QKVAttentionLegacy.forward     guided_diffusion/guided_diffusion/unet.py:341-358
  Upsample / Downsample          guided_diffusion/guided_diffusion/unet.py:105-115, 134-145
  Conv3d.forward                 guided_diffusion/guided_diffusion/nn.py:53-87
  GroupNorm32 / normalization    guided_diffusion/guided_diffusion/nn.py:26-28, 161-168
  timestep_embedding             guided_diffusion/guided_diffusion/nn.py:171-189
  PerceiverResampler/Attention   guided_diffusion/guided_diffusion/imagen.py:254-372
  LayerNorm / FeedForward        guided_diffusion/guided_diffusion/imagen.py:198-211, 1009-1017
"""
from dataclasses import dataclass
import math
import torch
import torch.nn.functional as F


# The reference casts to fp32 at three places (GroupNorm32, nn.py:26-28; the attention softmax, unet.py:356; `h = x.type(self.dtype)`,
# unet.py:666).  Tolerance studies run this restatement in fp64 (tests: "how far is the reference's own fp32 run from exact
# arithmetic?") and set WORK_DTYPE = torch.float64 for the duration; the default reproduces the reference bit for bit.
WORK_DTYPE = torch.float32


@dataclass(frozen=True)
class UNetCfg:
    in_channels: int = 6
    model_channels: int = 128
    out_channels: int = 3
    num_res_blocks: int = 2
    attention_resolutions: tuple = (8, 16)
    channel_mult: tuple = (1, 2, 3, 4, 5)
    num_head_channels: int = 32
    task_token_channels: int = 512
    # PerceiverResampler defaults (imagen.py:322-333)
    pr_depth: int = 2
    pr_dim_head: int = 64
    pr_heads: int = 8
    pr_num_latents: int = 64
    pr_num_mean_pooled: int = 4
    pr_ff_mult: int = 4


LIBERO_CFG = UNetCfg()


def build_program(cfg: UNetCfg):
    """Flatten UNetModel.__init__ into (input_ops, middle_ops, output_ops); each block is a list of
    ('conv', prefix) | ('res', prefix, cin, cout) | ('attn', prefix, C) | ('down', prefix, C) | ('up', prefix, C)."""
    mc = cfg.model_channels
    ch = int(cfg.channel_mult[0] * mc)
    inp = [[("conv", "input_blocks.0.0", cfg.in_channels, ch)]]
    chans = [ch]
    ds = 1
    idx = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            blk = [("res", f"input_blocks.{idx}.0", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in cfg.attention_resolutions:
                blk.append(("attn", f"input_blocks.{idx}.1", ch))
            inp.append(blk)
            chans.append(ch)
            idx += 1
        if level != len(cfg.channel_mult) - 1:
            inp.append([("down", f"input_blocks.{idx}.0", ch)])
            chans.append(ch)
            idx += 1
            ds *= 2
    mid = [("res", "middle_block.0", ch, ch), ("attn", "middle_block.1", ch), ("res", "middle_block.2", ch, ch)]
    out = []
    oidx = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            blk = [("res", f"output_blocks.{oidx}.0", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            j = 1
            if ds in cfg.attention_resolutions:
                blk.append(("attn", f"output_blocks.{oidx}.{j}", ch))
                j += 1
            if level and i == cfg.num_res_blocks:
                blk.append(("up", f"output_blocks.{oidx}.{j}", ch))
                ds //= 2
            out.append(blk)
            oidx += 1
    return inp, mid, out, ch


# ----------------------------------------------------------------------------- primitives
def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def conv3d(P, pre, x, stride=1):
    """Factorised conv: Conv2d over (h,w) per frame, then (k>1 only) Conv1d(k=3) over frames with
    symmetric zero pad 1+1.  x: [B,C,F,H,W]."""
    B, C, Fr, H, W = x.shape
    w = P[pre + ".spatial_conv.weight"]
    k = w.shape[-1]
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W), w, P[pre + ".spatial_conv.bias"],
                 stride=stride, padding=k // 2)
    Co, Ho, Wo = y.shape[1:]
    y = y.reshape(B, Fr, Co, Ho, Wo).permute(0, 2, 1, 3, 4)
    if (pre + ".temporal_conv.weight") not in P:
        return y
    z = y.permute(0, 3, 4, 1, 2).reshape(B * Ho * Wo, Co, Fr)
    z = F.conv1d(F.pad(z, (1, 1)), P[pre + ".temporal_conv.weight"], P[pre + ".temporal_conv.bias"])
    return z.reshape(B, Ho, Wo, Co, Fr).permute(0, 3, 4, 1, 2)


def gn32(P, pre, x):
    return F.group_norm(x.to(WORK_DTYPE), 32, P[pre + ".weight"], P[pre + ".bias"], eps=1e-5)


def resblock(P, pre, x, emb):
    h = conv3d(P, pre + ".in_layers.2", F.silu(gn32(P, pre + ".in_layers.0", x)))
    e = F.linear(F.silu(emb), P[pre + ".emb_layers.1.weight"], P[pre + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None, None]
    h = conv3d(P, pre + ".out_layers.3", F.silu(gn32(P, pre + ".out_layers.0", h)))
    if (pre + ".skip_connection.spatial_conv.weight") in P:
        x = conv3d(P, pre + ".skip_connection", x)
    return x + h


def attention_block(P, pre, x, head_ch):
    B, C, Fr, H, W = x.shape
    xf = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H * W)
    qkv = F.conv1d(gn32(P, pre + ".norm", xf), P[pre + ".qkv.weight"], P[pre + ".qkv.bias"])
    nh = C // head_ch
    N, _, L = qkv.shape
    q, k, v = qkv.reshape(N * nh, 3 * head_ch, L).split(head_ch, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(head_ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).to(WORK_DTYPE), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(N, C, L)
    h = F.conv1d(a, P[pre + ".proj_out.weight"], P[pre + ".proj_out.bias"])
    return (xf + h).reshape(B, Fr, C, H, W).permute(0, 2, 1, 3, 4)


def _ln_g(x, g):
    var = torch.var(x, dim=-1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=-1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def perceiver_resampler(P, pre, x, cfg: UNetCfg):
    """x: [B,L,D] -> latents [B, n_mean+n_lat, D]."""
    B, L, D = x.shape
    xp = x + P[pre + ".pos_emb.weight"][:L]
    lat = P[pre + ".latents"][None].expand(B, -1, -1)
    if cfg.pr_num_mean_pooled > 0:
        mp = x.sum(dim=1) / torch.tensor(float(L)).clamp(min=1e-5)  # masked_mean with an all-true mask
        mp = _ln_g(mp, P[pre + ".to_latents_from_mean_pooled_seq.0.g"])
        mp = F.linear(mp, P[pre + ".to_latents_from_mean_pooled_seq.1.weight"],
                      P[pre + ".to_latents_from_mean_pooled_seq.1.bias"]).reshape(B, cfg.pr_num_mean_pooled, D)
        lat = torch.cat([mp, lat], dim=1)
    h, dh = cfg.pr_heads, cfg.pr_dim_head
    for li in range(cfg.pr_depth):
        a = f"{pre}.layers.{li}.0"
        xn = F.layer_norm(xp, (D,), P[a + ".norm.weight"], P[a + ".norm.bias"])
        ln = F.layer_norm(lat, (D,), P[a + ".norm_latents.weight"], P[a + ".norm_latents.bias"])
        q = F.linear(ln, P[a + ".to_q.weight"])
        kv = F.linear(torch.cat([xn, ln], dim=1), P[a + ".to_kv.weight"])
        k, v = kv.chunk(2, dim=-1)
        sp = lambda t: t.reshape(B, t.shape[1], h, dh).permute(0, 2, 1, 3)
        q, k, v = sp(q), sp(k), sp(v)
        q = F.normalize(q, dim=-1) * P[a + ".q_scale"]
        k = F.normalize(k, dim=-1) * P[a + ".k_scale"]
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * 8.0
        out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
        out = out.permute(0, 2, 1, 3).reshape(B, -1, h * dh)
        out = F.linear(out, P[a + ".to_out.0.weight"])
        out = F.layer_norm(out, (D,), P[a + ".to_out.1.weight"], P[a + ".to_out.1.bias"])
        lat = out + lat
        f = f"{pre}.layers.{li}.1"
        y = F.linear(_ln_g(lat, P[f + ".0.g"]), P[f + ".1.weight"])
        y = F.gelu(y)
        y = F.linear(_ln_g(y, P[f + ".3.g"]), P[f + ".4.weight"])
        lat = y + lat
    return lat


def label_embedding(P, y, cfg: UNetCfg, pre=""):
    """task_attnpool(y).mean(1): t-independent part of emb.  [B,L,512] -> [B, 4*mc]."""
    lat = perceiver_resampler(P, pre + "task_attnpool.0", y, cfg)
    return F.linear(lat, P[pre + "task_attnpool.1.weight"], P[pre + "task_attnpool.1.bias"]).mean(dim=1)


def time_embedding(P, t, cfg: UNetCfg, pre=""):
    e = timestep_embedding(t, cfg.model_channels).to(P[pre + "time_embed.0.weight"].dtype)     # (fp64 runs of the oracle: tolerance studies)
    e = F.linear(e, P[pre + "time_embed.0.weight"], P[pre + "time_embed.0.bias"])
    return F.linear(F.silu(e), P[pre + "time_embed.2.weight"], P[pre + "time_embed.2.bias"])


def _run_block(P, pre, blk, h, emb, cfg):
    for op in blk:
        kind, name = op[0], pre + op[1]
        if kind == "conv":
            h = conv3d(P, name, h)
        elif kind == "res":
            h = resblock(P, name, h, emb)
        elif kind == "attn":
            h = attention_block(P, name, h, cfg.num_head_channels)
        elif kind == "down":
            h = conv3d(P, name + ".op", h, stride=2)
        elif kind == "up":
            B, C, Fr, H, W = h.shape
            h = F.interpolate(h, (Fr, H * 2, W * 2), mode="nearest")
            h = conv3d(P, name + ".conv", h)
    return h


def unet_forward(P, x, t, y, cfg: UNetCfg = LIBERO_CFG, pre=""):
    """UNetModel.forward.  x: [B,Cin,F,H,W], t: [B] long, y: [B,L,512] -> [B,Cout,F,H,W]."""
    inp, mid, out, _ = build_program(cfg)
    emb = time_embedding(P, t, cfg, pre) + label_embedding(P, y, cfg, pre)
    hs = []
    h = x.to(WORK_DTYPE)
    for blk in inp:
        h = _run_block(P, pre, blk, h, emb, cfg)
        hs.append(h)
    h = _run_block(P, pre, mid, h, emb, cfg)
    for blk in out:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(P, pre, blk, h, emb, cfg)
    h = F.silu(gn32(P, pre + "out.0", h))
    return conv3d(P, pre + "out.2", h)


def unet_libero_forward(P, x, t, task_embed, cfg: UNetCfg = LIBERO_CFG, pre="unet."):
    """Unet_Libero.forward: x [B,(f+1)*3,H,W] (noisy frames then cond image) -> [B,f*3,H,W]."""
    B, C, H, W = x.shape
    f = C // 3 - 1
    cond = x[:, -3:, None].expand(B, 3, f, H, W)
    xx = x[:, :-3].reshape(B, f, 3, H, W).permute(0, 2, 1, 3, 4)
    out = unet_forward(P, torch.cat([xx, cond], dim=1), t, task_embed, cfg, pre)
    return out.permute(0, 2, 1, 3, 4).reshape(B, f * cfg.out_channels, H, W)
