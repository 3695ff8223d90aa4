"""CPU restatement (torch fp32, autograd for gradients) of the goal-conditioned diffusion policy.
TEST INFRASTRUCTURE.

Functional over a {name: tensor} dict carrying the reference's parameter names.  Follows
(paths relative to /root/reference/diffuser/diffusion_policy/):
  LimitsConstNormalizer.normalize/unnormalize   normalizer.py:139-161
  MultiImageObsEncoder.forward (sorted keys)    model/multi_image_obs_encoder.py:132,144-196; GN swap :66-74
  ResNet18Conv / VisualCore                     common/vision_nets.py:9-39, 65-177
      (torchvision 0.15.1 resnet18 topology restated from the published architecture: third-party,
       absent from /root/reference -> parity unpinned for the topology, pinned for everything in-tree)
  SpatialSoftmax.forward                        common/base_nets.py:234-285
  SinusoidalPosEmb                              model/positional_embedding.py:10-17
  Conv1dBlock / Downsample1d / Upsample1d       model/conv1d_components.py:7-40
  ConditionalResidualBlock1D.forward            model/conditional_unet1d.py:46-66
  ConditionalUnet1D.forward                     model/conditional_unet1d.py:178-246
  DiffusionUnetImagePolicy.compute_loss         diffusion_unet_image_policy.py:204-277
  DiffusionUnetImagePolicy.predict_action       diffusion_unet_image_policy.py:88-201
"""
from dataclasses import dataclass
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import schedulers as S


@dataclass(frozen=True)
class PolicyCfg:
    image_hw: tuple = (128, 128)
    action_dim: int = 7
    horizon: int = 16
    n_action_steps: int = 8
    n_obs_steps: int = 1
    dsed: int = 128                      # diffusion_step_embed_dim
    down_dims: tuple = (256, 512, 1024)
    kernel_size: int = 5
    n_groups: int = 8
    num_kp: int = 32
    feature_dim: int = 64
    rgb_keys: tuple = ("img_goal_1", "img_obs_1")   # sorted(), multi_image_obs_encoder.py:132
    num_train_timesteps: int = 100
    num_inference_steps_ddim: int = 8
    # ResNet-18 widths (torchvision): only overridden by tiny test configs
    widths: tuple = (64, 128, 256, 512)
    act_min: tuple = None        # per-channel action limits of shape_meta (None = the Libero -1 / +1, diffuser/datasets/__init__.py:21-27)
    act_max: tuple = None


LIBERO_POLICY = PolicyCfg()


def normalize_img(x):
    """LimitsConstNormalizer.normalize with min 0 / max 1."""
    return 2 * ((x - 0.0) / (1.0 - 0.0)) - 1


def _limits(cfg, like):
    lo = torch.full((like.shape[-1],), -1.0) if cfg is None or cfg.act_min is None else torch.tensor(cfg.act_min, dtype=torch.float32)
    hi = torch.full((like.shape[-1],), 1.0) if cfg is None or cfg.act_max is None else torch.tensor(cfg.act_max, dtype=torch.float32)
    return lo, hi


def normalize_act(a, cfg=None):
    """LimitsConstNormalizer.normalize (normalizer.py:139-146)."""
    lo, hi = _limits(cfg, a)
    return 2 * ((a - lo) / (hi - lo)) - 1


def unnormalize_act(x, cfg=None):
    """LimitsConstNormalizer.unnormalize (normalizer.py:148-161): clamp only when something is out of range."""
    lo, hi = _limits(cfg, x)
    if x.max() > 1 or x.min() < -1:
        x = torch.clamp(x, -1, 1)
    x = (x + 1) / 2.0
    return x * (hi - lo) + lo


# ----------------------------------------------------------------------------- image encoder
def _gn(P, pre, x, groups):
    return F.group_norm(x, groups, P[pre + ".weight"], P[pre + ".bias"], eps=1e-5)


# The encoders' only DISCRETE decisions are the ReLU masks and the max-pool winners.  Two correct fp32 implementations can take one of
# them differently where a pre-activation is within rounding of zero / two window elements are within rounding of each other, and one
# flipped decision moves a small weight gradient by 1 / (B H W) of its terms (tests/test_policy_gpu.py).  `dec` lets a test take that
# out of a gradient comparison:  dec = {"record": {}}  stores this run's decisions (masks as NCHW bool, pool winners as window taps 0..8
# int64) and the pre-activations they were taken on;  dec = {"use": {...}}  routes forward and backward through GIVEN decisions
# (y = z * mask, y = the window element the given tap names) instead of taking its own.
def _relu(z, name, dec):
    if dec is None:
        return F.relu(z)
    if "record" in dec:
        dec["record"][name] = (z > 0).detach()
        dec["record"][name + ":z"] = z.detach()
        return F.relu(z)
    return z * dec["use"][name].to(z.dtype)


def _maxpool3x3s2(z, name, dec):
    if dec is None:
        return F.max_pool2d(z, 3, 2, 1)
    N, C, H, W = z.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    if "record" in dec:
        y, flat = F.max_pool2d(z, 3, 2, 1, return_indices=True)          # flat = ih * W + iw of the winner
        ih, iw = flat // W, flat % W
        oh = torch.arange(OH).view(1, 1, OH, 1)
        ow = torch.arange(OW).view(1, 1, 1, OW)
        dec["record"][name] = ((ih - (2 * oh - 1)) * 3 + (iw - (2 * ow - 1))).detach()
        dec["record"][name + ":z"] = z.detach()
        return y
    tap = dec["use"][name].long()
    oh = torch.arange(OH).view(1, 1, OH, 1)
    ow = torch.arange(OW).view(1, 1, 1, OW)
    ih, iw = 2 * oh - 1 + tap // 3, 2 * ow - 1 + tap % 3
    assert bool(((ih >= 0) & (ih < H) & (iw >= 0) & (iw < W)).all()), "a given max-pool winner lies in the padding"
    return z.flatten(2).gather(2, (ih * W + iw).flatten(2)).view(N, C, OH, OW)


def _basic_block(P, pre, x, stride, cout, dec=None):
    g = cout // 16
    out = F.conv2d(x, P[pre + ".conv1.weight"], None, stride=stride, padding=1)
    out = _relu(_gn(P, pre + ".bn1", out, g), pre + ".relu1", dec)
    out = F.conv2d(out, P[pre + ".conv2.weight"], None, stride=1, padding=1)
    out = _gn(P, pre + ".bn2", out, g)
    if (pre + ".downsample.0.weight") in P:
        x = _gn(P, pre + ".downsample.1", F.conv2d(x, P[pre + ".downsample.0.weight"], None, stride=stride), g)
    return _relu(out + x, pre + ".relu2", dec)


def resnet18_gn(P, pre, x, widths=(64, 128, 256, 512), dec=None):
    """children()[:-2] of resnet18 with BatchNorm2d -> GroupNorm(C//16, C).  pre = '...backbone.nets'."""
    x = F.conv2d(x, P[pre + ".0.weight"], None, stride=2, padding=3)
    x = _relu(_gn(P, pre + ".1", x, widths[0] // 16), pre + ".relu0", dec)
    x = _maxpool3x3s2(x, pre + ".pool", dec)
    for li, c in enumerate(widths):
        for bi in range(2):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = _basic_block(P, f"{pre}.{4 + li}.{bi}", x, stride, c, dec)
    return x


def spatial_softmax(P, pre, feat):
    """1x1 conv to K keypoints, softmax over H*W (temperature 1), expected (x, y)."""
    B, C, H, W = feat.shape
    f = F.conv2d(feat, P[pre + ".nets.weight"], P[pre + ".nets.bias"])
    K = f.shape[1]
    px, py = np.meshgrid(np.linspace(-1.0, 1.0, W), np.linspace(-1.0, 1.0, H))
    px = torch.from_numpy(px.reshape(1, H * W)).float()
    py = torch.from_numpy(py.reshape(1, H * W)).float()
    att = F.softmax(f.reshape(-1, H * W) / 1.0, dim=-1)
    ex = torch.sum(px * att, dim=1, keepdim=True)
    ey = torch.sum(py * att, dim=1, keepdim=True)
    return torch.cat([ex, ey], 1).view(-1, K, 2)


def visual_core(P, pre, img, cfg: PolicyCfg, dec=None):
    feat = resnet18_gn(P, pre + ".backbone.nets", img, cfg.widths, dec)
    kp = spatial_softmax(P, pre + ".pool", feat).flatten(1)
    return F.linear(kp, P[pre + ".nets.3.weight"], P[pre + ".nets.3.bias"])


def obs_encoder(P, nobs: dict, cfg: PolicyCfg, pre="obs_encoder.", dec=None):
    feats = [visual_core(P, f"{pre}key_model_map.{k}", nobs[k], cfg, dec) for k in cfg.rgb_keys]
    return torch.cat(feats, dim=-1)


# ----------------------------------------------------------------------------- ConditionalUnet1D
def sinusoidal_pos_emb(t, dim):
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = t[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def _conv1d_block(P, pre, x, k, groups):
    x = F.conv1d(x, P[pre + ".block.0.weight"], P[pre + ".block.0.bias"], padding=k // 2)
    return F.mish(F.group_norm(x, groups, P[pre + ".block.1.weight"], P[pre + ".block.1.bias"], eps=1e-5))


def _cond_res_block(P, pre, x, cond, k, groups):
    out = _conv1d_block(P, pre + ".blocks.0", x, k, groups)
    e = F.linear(F.mish(cond), P[pre + ".cond_encoder.1.weight"], P[pre + ".cond_encoder.1.bias"])
    C = out.shape[1]
    e = e.reshape(e.shape[0], 2, C, 1)
    out = e[:, 0] * out + e[:, 1]
    out = _conv1d_block(P, pre + ".blocks.1", out, k, groups)
    if (pre + ".residual_conv.weight") in P:
        x = F.conv1d(x, P[pre + ".residual_conv.weight"], P[pre + ".residual_conv.bias"])
    return out + x


def cond_unet1d(P, sample, t, global_cond, cfg: PolicyCfg, pre="model."):
    """sample [B,T,Da], t [B] long, global_cond [B,G] -> [B,T,Da]."""
    k, g = cfg.kernel_size, cfg.n_groups
    x = sample.permute(0, 2, 1)
    B = x.shape[0]
    if t.dim() == 0:
        t = t[None]
    t = t.expand(B)
    e = sinusoidal_pos_emb(t, cfg.dsed).to(P[pre + "diffusion_step_encoder.1.weight"].dtype)   # (fp64 runs of the oracle: tolerance studies)
    e = F.linear(e, P[pre + "diffusion_step_encoder.1.weight"], P[pre + "diffusion_step_encoder.1.bias"])
    e = F.linear(F.mish(e), P[pre + "diffusion_step_encoder.3.weight"], P[pre + "diffusion_step_encoder.3.bias"])
    gf = torch.cat([e, global_cond], dim=-1)
    n = len(cfg.down_dims)
    hs = []
    for i in range(n):
        x = _cond_res_block(P, f"{pre}down_modules.{i}.0", x, gf, k, g)
        x = _cond_res_block(P, f"{pre}down_modules.{i}.1", x, gf, k, g)
        hs.append(x)
        if i < n - 1:
            x = F.conv1d(x, P[f"{pre}down_modules.{i}.2.conv.weight"], P[f"{pre}down_modules.{i}.2.conv.bias"],
                         stride=2, padding=1)
    for i in range(2):
        x = _cond_res_block(P, f"{pre}mid_modules.{i}", x, gf, k, g)
    for i in range(n - 1):
        x = torch.cat((x, hs.pop()), dim=1)
        x = _cond_res_block(P, f"{pre}up_modules.{i}.0", x, gf, k, g)
        x = _cond_res_block(P, f"{pre}up_modules.{i}.1", x, gf, k, g)
        # is_last = ind >= len(in_out)-1 is never true inside this loop (conditional_unet1d.py:148-160)
        x = F.conv_transpose1d(x, P[f"{pre}up_modules.{i}.2.conv.weight"], P[f"{pre}up_modules.{i}.2.conv.bias"],
                               stride=2, padding=1)
    x = _conv1d_block(P, pre + "final_conv.0", x, k, 8)   # Conv1dBlock default n_groups=8 (:163)
    x = F.conv1d(x, P[pre + "final_conv.1.weight"], P[pre + "final_conv.1.bias"])
    return x.permute(0, 2, 1)


# ----------------------------------------------------------------------------- policy level
def compute_loss(P, batch, noise, timesteps, cfg: PolicyCfg = LIBERO_POLICY, dec=None):
    """compute_loss with the two RNG draws (randn then randint, :246-252) injected.  dec: see _relu / _maxpool3x3s2 (tests only)."""
    nobs = {k: normalize_img(v)[:, 0] for k, v in batch["obs"].items()}
    nact = normalize_act(batch["action"], cfg)
    gc = obs_encoder(P, nobs, cfg, dec=dec).reshape(nact.shape[0], -1)
    ac = S.squaredcos_alphas_cumprod(cfg.num_train_timesteps)
    noisy = S.add_noise(ac, nact, noise, timesteps)
    pred = cond_unet1d(P, noisy, timesteps, gc, cfg)
    loss = F.mse_loss(pred, noise, reduction="none")
    return loss.reshape(loss.shape[0], -1).mean(dim=1).mean()


def loss_and_grads(P, batch, noise, timesteps, cfg: PolicyCfg = LIBERO_POLICY, names=None, dec=None):
    """Loss + dLoss/dparam for every (deduplicated) floating tensor in `names` (default: all of P)."""
    names = list(names) if names is not None else [k for k, v in P.items() if torch.is_floating_point(v)]
    Q = dict(P)
    leaves = {}
    for n in names:
        leaves[n] = P[n].detach().clone().requires_grad_(True)
        Q[n] = leaves[n]
    loss = compute_loss(Q, batch, noise, timesteps, cfg, dec=dec)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    return loss.detach(), {n: (g if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, grads)}


@torch.no_grad()
def predict_action(P, obs, init_noise, step_noises, cfg: PolicyCfg = LIBERO_POLICY, use_ddim=False):
    """predict_action with injected RNG: init_noise = trajectory draw (:97-101); step_noises[i] = the
    variance noise of DDPM step i (t>0 only; unused for DDIM, eta=0)."""
    nobs = {k: normalize_img(v)[:, 0] for k, v in obs.items()}
    B = init_noise.shape[0]
    gc = obs_encoder(P, nobs, cfg).reshape(B, -1)
    ac = S.squaredcos_alphas_cumprod(cfg.num_train_timesteps)
    traj = init_noise
    if use_ddim:
        ts = S.ddim_timesteps(cfg.num_train_timesteps, cfg.num_inference_steps_ddim)
        for t in ts:
            eps = cond_unet1d(P, traj, torch.tensor(t), gc, cfg)
            traj = S.ddim_step(ac, eps, t, traj, cfg.num_train_timesteps, cfg.num_inference_steps_ddim)
    else:
        k = 0
        for t in range(cfg.num_train_timesteps - 1, -1, -1):
            eps = cond_unet1d(P, traj, torch.tensor(t), gc, cfg)
            nz = None
            if t > 0:
                nz = step_noises[k]
                k += 1
            traj = S.ddpm_step(ac, eps, t, traj, nz)
    act = unnormalize_act(traj[..., :cfg.action_dim], cfg)
    start = cfg.n_obs_steps - 1
    return {"action": act[:, start:start + cfg.n_action_steps], "action_pred": act}
