"""Restatement of the two `diffusers` schedulers the policy uses.  TEST INFRASTRUCTURE.

THIRD-PARTY, PARITY UNPINNED: `diffusers` is an un-pinned dependency of the reference
(/root/reference/requirements.txt:4) and is absent from /root/reference and from this image.
What is restated here is the published DDPM/DDIM scheduler algorithm as configured by
/root/reference/config/diff_policy/lb_train_diffusion_unet_image_orn10.yaml:45-53,101-113
(squaredcos_cap_v2 betas, T=100, epsilon prediction, clip_sample, fixed_small variance; DDIM
"leading" spacing, eta=0, set_alpha_to_one) and anchored on the reference's call sites
diffusion_unet_image_policy.py:106-128 (set_timesteps/step) and :250-256 (add_noise).
"""
import math
import numpy as np
import torch


def squaredcos_betas(n=100, max_beta=0.999):
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)], dtype=torch.float32)


def squaredcos_alphas_cumprod(n=100):
    return torch.cumprod(1.0 - squaredcos_betas(n), dim=0)


def add_noise(ac, x, noise, t):
    a = ac[t] ** 0.5
    b = (1 - ac[t]) ** 0.5
    while a.dim() < x.dim():
        a = a.unsqueeze(-1)
        b = b.unsqueeze(-1)
    return a * x + b * noise


def ddim_timesteps(T=100, n=8):
    return [int(v) for v in (np.arange(0, n) * (T // n)).round()[::-1]]


def ddpm_coeffs(ac, t, T=100, n=None):
    """Per-step scalars of DDPMScheduler.step (fp32 tensor arithmetic, as diffusers does it)."""
    n = n or T
    prev_t = t - T // n
    one = torch.tensor(1.0)
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else one
    b_t = 1 - a_t
    b_prev = 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    return dict(sqrt_b_t=b_t ** 0.5, sqrt_a_t=a_t ** 0.5, c0=(a_prev ** 0.5 * cur_b) / b_t,
                ct=cur_a ** 0.5 * b_prev / b_t,
                sigma=torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20) ** 0.5)


def ddpm_step(ac, eps, t, sample, noise, T=100):
    c = ddpm_coeffs(ac, t, T)
    x0 = ((sample - c["sqrt_b_t"] * eps) / c["sqrt_a_t"]).clamp(-1, 1)
    prev = c["c0"] * x0 + c["ct"] * sample
    if t > 0:
        prev = prev + c["sigma"] * noise
    return prev


def ddim_coeffs(ac, t, T=100, n=8):
    prev_t = t - T // n
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t = 1 - a_t
    return dict(sqrt_b_t=b_t ** 0.5, sqrt_a_t=a_t ** 0.5, sqrt_a_prev=a_prev ** 0.5,
                dir=(1 - a_prev) ** 0.5)


def ddim_step(ac, eps, t, sample, T=100, n=8):
    c = ddim_coeffs(ac, t, T, n)
    x0 = ((sample - c["sqrt_b_t"] * eps) / c["sqrt_a_t"]).clamp(-1, 1)
    return c["sqrt_a_prev"] * x0 + c["dir"] * eps
