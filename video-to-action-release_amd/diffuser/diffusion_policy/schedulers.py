"""Minimal DDPM / DDIM scheduler objects with the `diffusers` attribute surface the policy touches
(.config.num_train_timesteps, .config.prediction_type, .set_timesteps, .timesteps, .alphas_cumprod).
`diffusers` is an un-pinned third-party dependency of the reference (requirements.txt:4) and is not installed here; when it
is importable the user's scheduler objects are accepted as they are -- the HIP path only reads their config."""
import math
import numpy as np
import torch


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class _SchedulerBase:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=True,
                 prediction_type="epsilon", **kw):
        if beta_schedule != "squaredcos_cap_v2" or prediction_type != "epsilon" or not clip_sample:
            raise NotImplementedError("HIP policy sampler implements squaredcos_cap_v2 / epsilon / clip_sample (Libero yaml)")
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, prediction_type=prediction_type, **kw)
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        n = num_train_timesteps
        self.betas = torch.tensor([min(1 - ab((i + 1) / n) / ab(i / n), 0.999) for i in range(n)], dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_inference_steps = None
        self.timesteps = torch.arange(n - 1, -1, -1)

    def set_timesteps(self, n):
        T = self.config.num_train_timesteps
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy((np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64))


class DDPMScheduler(_SchedulerBase):
    def __init__(self, variance_type="fixed_small", **kw):
        if variance_type != "fixed_small":
            raise NotImplementedError("fixed_small variance only")
        super().__init__(variance_type=variance_type, **kw)


class DDIMScheduler(_SchedulerBase):
    def __init__(self, set_alpha_to_one=True, steps_offset=0, **kw):
        if not set_alpha_to_one or steps_offset != 0:
            raise NotImplementedError("set_alpha_to_one=True, steps_offset=0 only")
        super().__init__(set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, **kw)
