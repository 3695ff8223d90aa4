"""Make the reference's OWN modules importable in the build container (no GPU, many
third-party packages absent) so golden vectors can be generated from them.

Nothing here ships to the GPU box and nothing from /root/reference is copied: the reference
is imported from where it lies (read-only), under placeholder modules for packages that are
not installed.  Functional stand-ins (torchvision.resnet18, the
diffusers schedulers, ema_pytorch.EMA) are restatements of the *published* algorithms; any
fixture derived through them is tagged "third-party restated" by tools/make_golden.py.

Usage:  import tools.ref_shims as rs; rs.install(); import flowdiffusion.flowdiffusion.unet
"""
import os
import sys
import types
import math
import copy

REFERENCE_ROOT = "/root/reference"


class _Dummy:
    """Attribute sink: any attribute is a dummy class, callable, subclassable."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Dummy,), {})


def _mod(name, **attrs):
    import importlib.machinery
    m = _DummyModule(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    m.__path__ = []  # so that submodule imports resolve through sys.modules
    sys.modules[name] = m
    return m


def _install_einops_exts():
    import einops

    def rearrange_many(tensors, pattern, **kw):
        return tuple(einops.rearrange(t, pattern, **kw) for t in tensors)

    def repeat_many(tensors, pattern, **kw):
        return tuple(einops.repeat(t, pattern, **kw) for t in tensors)

    def check_shape(*a, **k):
        return None

    import torch.nn as nn

    class EinopsToAndFrom(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _mod("einops_exts", rearrange_many=rearrange_many, repeat_many=repeat_many, check_shape=check_shape)
    _mod("einops_exts.torch", EinopsToAndFrom=EinopsToAndFrom)


# ---------------------------------------------------------------- torchvision.resnet18
def _install_torchvision():
    import torch
    import torch.nn as nn

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.downsample = downsample
            self.stride = stride

        def forward(self, x):
            identity = x
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            if self.downsample is not None:
                identity = self.downsample(x)
            out = out + identity
            return self.relu(out)

    class ResNet18(nn.Module):
        """torchvision.models.resnet18 topology + child order (conv1,bn1,relu,maxpool,layer1-4,avgpool,fc)."""

        def __init__(self):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = self._make(64, 2, 1)
            self.layer2 = self._make(128, 2, 2)
            self.layer3 = self._make(256, 2, 2)
            self.layer4 = self._make(512, 2, 2)
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512, 1000)
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

        def _make(self, planes, blocks, stride):
            down = None
            if stride != 1 or self.inplanes != planes:
                down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
            layers = [BasicBlock(self.inplanes, planes, stride, down)]
            self.inplanes = planes
            for _ in range(1, blocks):
                layers.append(BasicBlock(planes, planes))
            return nn.Sequential(*layers)

    def resnet18(weights=None, **kw):
        return ResNet18()

    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models", resnet18=resnet18)
    tv.transforms = _mod("torchvision.transforms")
    tv.utils = _mod("torchvision.utils")


# ---------------------------------------------------------------- diffusers schedulers
def _install_diffusers():
    import torch

    def _betas_squaredcos(n, max_beta=0.999):
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return torch.tensor([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)], dtype=torch.float32)

    class _Cfg(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    class _Out:
        def __init__(self, prev_sample, pred_original_sample):
            self.prev_sample = prev_sample
            self.pred_original_sample = pred_original_sample

    class _Base:
        def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                     beta_schedule="linear", clip_sample=True, prediction_type="epsilon", **kw):
            assert beta_schedule == "squaredcos_cap_v2" and prediction_type == "epsilon"
            self.config = _Cfg(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type,
                               clip_sample=clip_sample, beta_schedule=beta_schedule, **kw)
            self.betas = _betas_squaredcos(num_train_timesteps)
            self.alphas = 1.0 - self.betas
            self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
            self.one = torch.tensor(1.0)
            self.num_inference_steps = None
            self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

        def add_noise(self, x, noise, t):
            ac = self.alphas_cumprod.to(device=x.device, dtype=x.dtype)
            a = ac[t] ** 0.5
            b = (1 - ac[t]) ** 0.5
            while a.dim() < x.dim():
                a = a.unsqueeze(-1)
                b = b.unsqueeze(-1)
            return a * x + b * noise

    class DDPMScheduler(_Base):
        def __init__(self, variance_type="fixed_small", **kw):
            super().__init__(**kw)
            assert variance_type == "fixed_small"

        def set_timesteps(self, n):
            T = self.config.num_train_timesteps
            self.num_inference_steps = n
            ratio = T // n
            import numpy as np
            self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

        def step(self, model_output, timestep, sample, generator=None, **kw):
            t = int(timestep)
            T = self.config.num_train_timesteps
            n = self.num_inference_steps or T
            prev_t = t - T // n
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
            b_t = 1 - a_t
            b_prev = 1 - a_prev
            cur_a = a_t / a_prev
            cur_b = 1 - cur_a
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            if self.config.clip_sample:
                x0 = x0.clamp(-1, 1)
            c0 = (a_prev ** 0.5 * cur_b) / b_t
            ct = cur_a ** 0.5 * b_prev / b_t
            prev = c0 * x0 + ct * sample
            if t > 0:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
                var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
                prev = prev + (var ** 0.5) * noise
            return _Out(prev, x0)

    class DDIMScheduler(_Base):
        def __init__(self, set_alpha_to_one=True, steps_offset=0, **kw):
            super().__init__(**kw)
            self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
            self.steps_offset = steps_offset

        def set_timesteps(self, n):
            import numpy as np
            T = self.config.num_train_timesteps
            self.num_inference_steps = n
            ratio = T // n
            ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
            self.timesteps = torch.from_numpy(ts)

        def step(self, model_output, timestep, sample, eta=0.0, generator=None, **kw):
            t = int(timestep)
            T = self.config.num_train_timesteps
            prev_t = t - T // self.num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            b_t = 1 - a_t
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            if self.config.clip_sample:
                x0 = x0.clamp(-1, 1)
            var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
            std = eta * var ** 0.5
            direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
            prev = a_prev ** 0.5 * x0 + direction
            return _Out(prev, x0)

    _mod("diffusers")
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_ddpm", DDPMScheduler=DDPMScheduler)
    _mod("diffusers.schedulers.scheduling_ddim", DDIMScheduler=DDIMScheduler)


# ---------------------------------------------------------------- ema_pytorch.EMA (0.2.3 behaviour)
def _install_ema_pytorch():
    import torch
    import torch.nn as nn

    class EMA(nn.Module):
        def __init__(self, model, ema_model=None, beta=0.9999, update_after_step=100, update_every=10,
                     inv_gamma=1.0, power=2 / 3, min_value=0.0, include_online_model=True, **kw):
            super().__init__()
            self.beta = beta
            self.include_online_model = include_online_model
            if include_online_model:
                self.online_model = model
            else:
                self.online_model = [model]
            self.ema_model = ema_model if ema_model is not None else copy.deepcopy(model)
            self.ema_model.requires_grad_(False)
            self.update_every = update_every
            self.update_after_step = update_after_step
            self.inv_gamma = inv_gamma
            self.power = power
            self.min_value = min_value
            self.register_buffer("initted", torch.Tensor([False]))
            self.register_buffer("step", torch.tensor([0]))

        @property
        def model(self):
            return self.online_model if self.include_online_model else self.online_model[0]

        def copy_params_from_model_to_ema(self):
            for (_, m), (_, c) in zip(self.ema_model.named_parameters(), self.model.named_parameters()):
                m.data.copy_(c.data)
            for (_, m), (_, c) in zip(self.ema_model.named_buffers(), self.model.named_buffers()):
                m.data.copy_(c.data)

        def get_current_decay(self):
            epoch = max(self.step.item() - self.update_after_step - 1, 0.0)
            value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
            if epoch <= 0:
                return 0.0
            return min(max(value, self.min_value), self.beta)

        def update(self):
            step = self.step.item()
            self.step += 1
            if (step % self.update_every) != 0:
                return
            if step <= self.update_after_step:
                self.copy_params_from_model_to_ema()
                return
            if not self.initted.item():
                self.copy_params_from_model_to_ema()
                self.initted.data.copy_(torch.Tensor([True]))
            decay = self.get_current_decay()
            with torch.no_grad():
                for (_, c), (_, m) in zip(self.model.named_parameters(), self.ema_model.named_parameters()):
                    if m.dtype in (torch.float, torch.float16):
                        d = m.data - c.data
                        d.mul_(1.0 - decay)
                        m.sub_(d)
                for (_, c), (_, m) in zip(self.model.named_buffers(), self.ema_model.named_buffers()):
                    if m.dtype in (torch.float, torch.float16):
                        d = m.data - c.data
                        d.mul_(1.0 - decay)
                        m.sub_(d)

        def forward(self, *a, **k):
            return self.ema_model(*a, **k)

    _mod("ema_pytorch", EMA=EMA)


def install():
    """Idempotent: install shims and put the reference tree on sys.path."""
    if getattr(install, "_done", False):
        return
    sys.dont_write_bytecode = True
    os.environ.setdefault("CONDA_DEFAULT_ENV", "probe")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    try:  # real packages that probe for optional deps must be imported before the placeholders exist
        import accelerate  # noqa: F401
        import matplotlib  # noqa: F401
    except Exception:
        pass
    _install_einops_exts()
    _install_torchvision()
    _install_diffusers()
    _install_ema_pytorch()

    class Env:
        pass

    class Wrapper:
        def __init__(self, *a, **k):
            pass

    g = _mod("gym", Env=Env, Wrapper=Wrapper)
    _mod("gym.envs")
    _mod("gym.envs.registration", register=lambda *a, **k: None)
    _mod("gym.utils")
    _mod("gym.spaces")
    for name in ["mujoco_py", "git", "h5py", "termcolor", "imageio", "wandb", "cv2", "mediapy",
                 "libero", "libero.libero", "libero.libero.envs", "libero.libero.benchmark",
                 "robosuite", "robosuite.macros", "torchvideotransforms", "pynvml_utils"]:
        _mod(name)

    class _Fore:
        def __getattr__(self, n):
            return ""

    _mod("colorama", Fore=_Fore(), Style=_Fore(), init=lambda *a, **k: None)
    pv = _mod("pynvml")
    pv.__all__ = []

    class Tap:
        def __init__(self, *a, **k):
            pass

    _mod("tap", Tap=Tap)

    class OmegaConf:
        @staticmethod
        def register_new_resolver(*a, **k):
            return None

        @staticmethod
        def load(*a, **k):
            raise RuntimeError("omegaconf shim: load not available")

    _mod("omegaconf", OmegaConf=OmegaConf)

    class CLIPTextModel(_Dummy):
        pass

    class CLIPTokenizer(_Dummy):
        pass

    _mod("transformers", CLIPTextModel=CLIPTextModel, CLIPTokenizer=CLIPTokenizer)
    install._done = True
