"""DiffusionUnetImagePolicy with the reference's plugin surface (diffusion_unet_image_policy.py:15-282), executed by
hand-written HIP kernels (v2a_hip.policy_engine).  The nn.Module tree only owns identically named parameters so that
AdamW / ema_pytorch.EMA (deepcopy) / state_dict checkpoints work unchanged; compute_loss returns a scalar whose
backward() delivers the hand-written gradients through one torch.autograd.Function.
"""
import copy
import weakref
from types import SimpleNamespace
from typing import Dict
import numpy as np
import torch
import torch.nn as nn

from .base_image_policy import BaseImagePolicy
from .normalizer import ConstNormalizerGroup, LimitsConstNormalizer
from .model.conditional_unet1d import ConditionalUnet1D
from .model.multi_image_obs_encoder import MultiImageObsEncoder


class _PolicyLossFn(torch.autograd.Function):
    """loss = compute_loss(batch); the forward already runs the HIP backward and parks the gradients in one arena."""

    @staticmethod
    def forward(ctx, policy, names, imgs, action, noise, timesteps, *params):
        need = any(p.requires_grad for p in params)
        loss, grads, arena = policy.engine.loss_fwd_bwd(imgs, action, noise, timesteps, need_grad=need, names=names)
        ctx.grads = None if grads is None else [grads[n] for n in names]
        ctx.arena = arena
        return loss.view(())

    @staticmethod
    def backward(ctx, gloss):
        if ctx.grads is None:
            raise RuntimeError("compute_loss was evaluated without gradients")
        from v2a_hip import ops
        ops.scale_by_device_scalar(ctx.arena, gloss)       # one kernel over the whole arena (GradScaler-safe)
        return (None, None, None, None, None, None) + tuple(ctx.grads)


class DiffusionUnetImagePolicy(BaseImagePolicy):
    def __init__(self, shape_meta: dict, noise_scheduler, noise_scheduler_ddim, obs_encoder: MultiImageObsEncoder, horizon,
                 n_action_steps, n_obs_steps, num_inference_steps=None, num_inference_steps_ddim=8, obs_as_global_cond=True,
                 diffusion_step_embed_dim=256, down_dims=(256, 512, 1024), kernel_size=5, n_groups=8, cond_predict_scale=True,
                 _target_=None, cond_unet1d_config={}, **kwargs):
        super().__init__()
        action_shape = shape_meta["action"]["shape"]
        assert len(action_shape) == 1
        action_dim = action_shape[0]
        obs_feature_dim = obs_encoder.output_shape()[0]
        if not obs_as_global_cond:
            raise NotImplementedError("obs_as_global_cond=False is not on the Libero path (the reference asserts too)")
        assert n_obs_steps == 1, "temporally"
        model = ConditionalUnet1D(input_dim=action_dim, local_cond_dim=None, global_cond_dim=obs_feature_dim * n_obs_steps,
                                  diffusion_step_embed_dim=diffusion_step_embed_dim, down_dims=down_dims, kernel_size=kernel_size,
                                  n_groups=n_groups, cond_predict_scale=cond_predict_scale, cond_unet1d_config=cond_unet1d_config)
        self.obs_encoder = obs_encoder
        self.model = model
        self.noise_scheduler = noise_scheduler
        self.noise_scheduler_ddim = noise_scheduler_ddim
        self.ddpm_var_temp = 1.0
        self.cond_unet1d_config = cond_unet1d_config
        self.normalizer = ConstNormalizerGroup(LimitsConstNormalizer, shape_meta, n_obs_steps)
        self.horizon = horizon
        self.obs_feature_dim = obs_feature_dim
        self.action_dim = action_dim
        self.n_action_steps = n_action_steps
        self.n_obs_steps = n_obs_steps
        self.obs_as_global_cond = obs_as_global_cond
        self.kwargs = kwargs
        if num_inference_steps is None:
            num_inference_steps = noise_scheduler.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        self.num_inference_steps_ddim = num_inference_steps_ddim
        self._check_fused_constants(shape_meta, noise_scheduler, noise_scheduler_ddim)
        a_min, a_max, _ = shape_meta["action"]["minmax_shape"]
        a_min, a_max = np.asarray(a_min, dtype=np.float32).reshape(-1), np.asarray(a_max, dtype=np.float32).reshape(-1)
        act_limits = None if (np.all(a_min == -1.0) and np.all(a_max == 1.0)) else (a_min, a_max)
        img_shape = tuple(next(iter(shape_meta["obs"].values()))["shape"])
        core = next(iter(obs_encoder.key_model_map.values()))
        self._cfg = SimpleNamespace(
            image_hw=img_shape[1:], action_dim=action_dim, horizon=horizon, dsed=diffusion_step_embed_dim,
            down_dims=tuple(down_dims), kernel_size=kernel_size, n_groups=n_groups, num_kp=core.pool._num_kp,
            feature_dim=core.feature_dimension, rgb_keys=tuple(obs_encoder.rgb_keys),
            num_train_timesteps=noise_scheduler.config.num_train_timesteps, widths=tuple(core.backbone._widths),
            act_limits=act_limits)
        self.__dict__["_engine"] = None
        # injected-RNG hook for parity tests: callable(shape, kind) -> tensor, kind in {"noise", "timesteps", "init", "step"}
        self.__dict__["_rng_hook"] = None

    @staticmethod
    def _check_fused_constants(shape_meta, sched, sched_ddim):
        """The HIP loaders fuse the image normalisation (2x-1 of [0,1] pixels) and evaluate the squaredcos_cap_v2 / epsilon /
        clip_sample schedule from its closed form.  Anything else must fail HERE, not come out silently mis-scaled."""
        from v2a_hip.policy_engine import squaredcos_alphas_cumprod
        for key, val in shape_meta["obs"].items():
            lo, hi, _ = val["minmax_shape"]
            if not (np.all(np.asarray(lo) == 0.0) and np.all(np.asarray(hi) == 1.0)):
                raise NotImplementedError(f"image limits of {key!r} are {lo}..{hi}: the fused image loader implements [0, 1] -> [-1, 1] only")
        a_lo, a_hi, _ = shape_meta["action"]["minmax_shape"]
        if not np.all(np.asarray(a_hi, dtype=np.float64) > np.asarray(a_lo, dtype=np.float64)):
            raise ValueError("action limits need max > min in every channel")
        n = sched.config.num_train_timesteps
        want = squaredcos_alphas_cumprod(n)
        for name, sc in (("noise_scheduler", sched), ("noise_scheduler_ddim", sched_ddim)):
            if sc is None:
                continue
            cfg = sc.config
            if getattr(cfg, "prediction_type", "epsilon") != "epsilon" or not getattr(cfg, "clip_sample", True):
                raise NotImplementedError(f"{name}: the fused scheduler step implements prediction_type='epsilon' with clip_sample=True")
            if cfg.num_train_timesteps != n:
                raise ValueError("both schedulers must share num_train_timesteps")
            ac = torch.as_tensor(sc.alphas_cumprod, dtype=torch.float32).cpu()
            if ac.shape != want.shape or float((ac - want).abs().max()) > 1e-6:
                raise NotImplementedError(f"{name}: alphas_cumprod is not the squaredcos_cap_v2 table the HIP kernels evaluate "
                                          f"(beta_schedule={getattr(cfg, 'beta_schedule', '?')!r})")

    # ------------------------------------------------------------------ engine plumbing
    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        d = dict(self.__dict__)
        d["_engine"] = None
        for k, v in d.items():
            if k in ("_engine",):
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    @property
    def engine(self):
        from v2a_hip.policy_engine import PolicyEngine
        dev = self._dummy_variable.device
        if dev.type != "cuda":
            raise RuntimeError("the MI355X-native policy runs on a HIP device only: call .to('cuda') first "
                               "(there is deliberately no CPU fallback)")
        eng = self.__dict__.get("_engine")
        if eng is None or eng.device != dev:
            params = {n: p for n, p in self.named_parameters()}
            eng = PolicyEngine(self._cfg, params)
            self.__dict__["_engine"] = eng
        ref = weakref.ref(self)
        self.model.__dict__["_owner_ref"] = ref
        self.obs_encoder.__dict__["_owner_ref"] = ref
        return eng

    def trainable_names(self):
        return [n for n in self.engine.trainable_names() if dict(self.named_parameters())[n].requires_grad]

    def _encode_normalized(self, nobs):
        """obs already in [-1,1] (NCHW): undo the affine map once so the fused loader can re-apply it exactly."""
        from v2a_hip import ops
        imgs = {k: (nobs[k].float().contiguous() + 1) * 0.5 for k in self._cfg.rgb_keys}
        return self.engine.global_cond(imgs)

    def _unet_forward(self, sample, timestep, global_cond):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=sample.device)
        elif timestep.dim() == 0:
            timestep = timestep[None].to(sample.device)
        t = timestep.expand(sample.shape[0]).contiguous().long()
        return self.engine.unet_fwd(sample.float().contiguous(), t, global_cond.float().contiguous())

    def _draw(self, shape, kind, device, high=None):
        hook = self.__dict__.get("_rng_hook")
        if hook is not None:
            return hook(shape, kind).to(device)
        if kind == "timesteps":
            return torch.randint(0, high, shape, device=device).long()
        return torch.randn(shape, device=device)

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def predict_action(self, obs_dict: Dict[str, torch.Tensor], use_ddim=False) -> Dict[str, torch.Tensor]:
        assert "past_action" not in obs_dict
        from v2a_hip import ops
        from v2a_hip.policy_sched import ddpm_coeffs, ddim_coeffs, ddim_timesteps
        eng = self.engine
        dev = eng.device
        To = self.n_obs_steps
        imgs = {k: obs_dict[k][:, :To].reshape(-1, *obs_dict[k].shape[2:]).to(dev).contiguous() for k in self._cfg.rgb_keys}
        B = next(iter(imgs.values())).shape[0]
        gc = eng.global_cond(imgs)
        T, Da = self.horizon, self.action_dim
        traj = self._draw((B, T, Da), "init", dev).float().contiguous()
        Ttr = self.noise_scheduler.config.num_train_timesteps
        if use_ddim:
            n = self.num_inference_steps_ddim
            for t in ddim_timesteps(Ttr, n):
                tt = torch.full((B,), t, dtype=torch.long, device=dev)
                eps = eng.unet_fwd(traj, tt, gc)
                traj = ops.policy_sched_step(eps, traj, None, ddim_coeffs(eng.ac_host, t, Ttr, n), mode=1)
        else:
            n = self.num_inference_steps
            assert n == Ttr, "DDPM inference uses every training timestep (yaml num_inference_steps: 100)"
            for t in range(Ttr - 1, -1, -1):
                tt = torch.full((B,), t, dtype=torch.long, device=dev)
                eps = eng.unet_fwd(traj, tt, gc)
                noise = self._draw((B, T, Da), "step", dev).float().contiguous() if t > 0 else None
                traj = ops.policy_sched_step(eps, traj, noise, ddpm_coeffs(eng.ac_host, t, Ttr), mode=0)
        action_pred = ops.unnormalize_action(traj, eng.act_limits).detach()
        start = To - 1
        return {"action": action_pred[:, start:start + self.n_action_steps], "action_pred": action_pred}

    # ------------------------------------------------------------------ training
    def compute_loss(self, batch: dict):
        assert "valid_mask" not in batch
        eng = self.engine
        dev = eng.device
        To = self.n_obs_steps
        imgs = {k: batch["obs"][k][:, :To].reshape(-1, *batch["obs"][k].shape[2:]).to(dev).contiguous() for k in self._cfg.rgb_keys}
        action = batch["action"].to(dev).float().contiguous()
        assert action.shape[-1] == self.action_dim
        B = action.shape[0]
        # RNG order of the reference: noise = randn(B,T,Da) then timesteps = randint(0, T_train, (B,))  (:246-252)
        noise = self._draw(tuple(action.shape), "noise", dev).float().contiguous()
        timesteps = self._draw((B,), "timesteps", dev, high=self.noise_scheduler.config.num_train_timesteps).long().contiguous()
        pred_type = self.noise_scheduler.config.prediction_type
        if pred_type != "epsilon":
            raise ValueError(f"Unsupported prediction type {pred_type}")
        named = dict(self.named_parameters())
        names = [n for n in eng.trainable_names() if named[n].requires_grad]
        params = [named[n] for n in names]
        if torch.is_grad_enabled() and params:
            return _PolicyLossFn.apply(self, names, imgs, action, noise, timesteps, *params)
        loss, _, _ = eng.loss_fwd_bwd(imgs, action, noise, timesteps, need_grad=False)
        return loss.view(())

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.normalizer.to_device(*args, **kwargs)
        return self
