"""Constant min/max normalisers (reference normalizer.py:6-161).  Host API only: inside the policy the same
arithmetic is fused into the HIP loaders (nchw_to_nhwc normalize=1, add_noise, unnormalize_action)."""
import numpy as np
import torch


class ConstNormalizer:
    def __init__(self, min_max, in_shape):
        self.mins = min_max[0].reshape(*in_shape)
        self.maxs = min_max[1].reshape(*in_shape)

    def __call__(self, x):
        return self.normalize(x)


class LimitsConstNormalizer(ConstNormalizer):
    """maps [xmin, xmax] to [-1, 1]"""

    def normalize(self, x):
        x = (x - self.mins) / (self.maxs - self.mins)
        return 2 * x - 1

    def unnormalize(self, x, eps=0):
        if x.max() > 1 + eps or x.min() < -1 - eps:
            x = torch.clamp(x, -1, 1) if torch.is_tensor(x) else np.clip(x, -1, 1)
        x = (x + 1) / 2.0
        return x * (self.maxs - self.mins) + self.mins


class ConstNormalizerGroup:
    def __init__(self, normalizer, shape_meta, n_obs_steps, use_tensor=True):
        if isinstance(normalizer, str):
            normalizer = {"LimitsConstNormalizer": LimitsConstNormalizer}[normalizer]
        self.normalizers = {}
        consts = {**shape_meta["obs"], "action": shape_meta["action"]}
        for key, val in consts.items():
            mn, mx, shp = val["minmax_shape"]
            if use_tensor:
                mn, mx = torch.as_tensor(np.asarray(mn)), torch.as_tensor(np.asarray(mx))
            v_shape = (shp[0], 1, *shp[1:])
            self.normalizers[key] = normalizer((mn, mx), v_shape)

    def __call__(self, *a, **k):
        return self.normalize(*a, **k)

    def normalize(self, x, key):
        return self.normalizers[key].normalize(x)

    def normalize_d(self, obs_dict):
        return {k: self.normalize(v, k) for k, v in obs_dict.items()}

    def __getitem__(self, key):
        return self.normalizers[key]

    def unnormalize(self, x, key):
        return self.normalizers[key].unnormalize(x)

    def to_device(self, *args, **kwargs):
        for val in self.normalizers.values():
            val.mins = val.mins.to(*args, **kwargs)
            val.maxs = val.maxs.to(*args, **kwargs)

    @property
    def device(self):
        m = next(iter(self.normalizers.values())).mins
        return m.device if torch.is_tensor(m) else None
