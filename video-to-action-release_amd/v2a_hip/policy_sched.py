"""Per-step scalar coefficients of the policy's DDPM / DDIM scheduler steps (fp32 tensor arithmetic, as diffusers evaluates
them), handed to the fused HIP step kernel.  Restated third-party algorithm -- see oracle/schedulers.py for provenance."""
import numpy as np
import torch


def ddim_timesteps(T=100, n=8):
    return [int(v) for v in (np.arange(0, n) * (T // n)).round()[::-1]]


def ddpm_coeffs(ac, t, T=100, n=None):
    n = n or T
    prev_t = t - T // n
    one = torch.tensor(1.0)
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else one
    b_t = 1 - a_t
    b_prev = 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    sigma = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20) ** 0.5
    return (float(b_t ** 0.5), float(a_t ** 0.5), float((a_prev ** 0.5 * cur_b) / b_t), float(cur_a ** 0.5 * b_prev / b_t), float(sigma))


def ddim_coeffs(ac, t, T=100, n=8):
    prev_t = t - T // n
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t = 1 - a_t
    return (float(b_t ** 0.5), float(a_t ** 0.5), float(a_prev ** 0.5), float((1 - a_prev) ** 0.5), 0.0)
