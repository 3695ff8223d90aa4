"""Init_Diffusion_Policy: YAML -> policy factory with the reference's surface (get_dp.py:24-101): attributes
`.diffusion_policy`, `.all_conf`, `.policy_conf`; horizon / n_action_steps overridden from args.trainer_dict (:93-95);
image size checked against args.input_img_size (:98-101).  omegaconf is not required: the yaml is read with PyYAML and the
three custom resolvers the Libero yaml uses (${image_minmax_01:}, ${lb_action_minmax:}, ${name}) are resolved here."""
import os
import re
from types import SimpleNamespace
import numpy as np
import yaml

from .diffusion_unet_image_policy import DiffusionUnetImagePolicy
from .model.multi_image_obs_encoder import MultiImageObsEncoder
from .common.vision_nets import VisualCore
from . import schedulers as _sched

_RESOLVERS = {
    "image_minmax_01": lambda: (np.array([0, 0, 0], dtype=np.float32), np.array([1, 1, 1], dtype=np.float32), [1, 3, 1, 1]),
    "lb_action_minmax": lambda: (np.array([-1.0] * 7, dtype=np.float32), np.array([1.0] * 7, dtype=np.float32), [1, 7]),
    "lb_action_minmax_orn01": lambda: (np.array([-1.0] * 3 + [-0.1] * 3 + [-1.0], dtype=np.float32),
                                       np.array([1.0] * 3 + [0.1] * 3 + [1.0], dtype=np.float32), [1, 7]),
}

# values of config/diff_policy/lb_train_diffusion_unet_image_orn10.yaml, used when no yaml path is given
DEFAULT_CONF = {
    "image_shape": [3, 128, 128], "horizon": 16, "n_obs_steps": 1, "n_action_steps": 8, "obs_as_global_cond": True,
    "policy": {
        "noise_scheduler": dict(num_train_timesteps=100, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2",
                                variance_type="fixed_small", clip_sample=True, prediction_type="epsilon"),
        "noise_scheduler_ddim": dict(num_train_timesteps=100, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2",
                                     clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon"),
        "obs_encoder": {"rgb_model": dict(backbone_class="ResNet18Conv", backbone_kwargs=dict(pretrained=None, input_coord_conv=False),
                                          pool_class="SpatialSoftmax",
                                          pool_kwargs=dict(num_kp=32, learnable_temperature=False, temperature=1.0, noise_std=0.0,
                                                           output_variance=False), flatten=True, feature_dimension=64),
                        "resize_shape": None, "crop_shape": None, "random_crop": None, "use_group_norm": True,
                        "share_rgb_model": False, "imagenet_norm": False},
        "num_inference_steps": 100, "diffusion_step_embed_dim": 128, "down_dims": [256, 512, 1024], "kernel_size": 5,
        "n_groups": 8, "cond_predict_scale": True, "num_inference_steps_ddim": 8,
    },
}


def _ns(d):
    if isinstance(d, dict):
        return SimpleNamespace(**{k: _ns(v) for k, v in d.items()})
    return d


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = re.fullmatch(r"\$\{([A-Za-z0-9_]+):\}", node)
        if m:
            return _RESOLVERS[m.group(1)]()
        m = re.fullmatch(r"\$\{([A-Za-z0-9_.]+)\}", node)
        if m:
            cur = root
            for part in m.group(1).split("."):
                cur = cur[part]
            return _resolve(cur, root)
    return node


def read_policy_yaml(path):
    with open(path) as f:
        raw = yaml.safe_load(f)
    return _resolve(raw, raw)


def build_policy(conf: dict, horizon=None, n_action_steps=None, widths=None):
    pc = conf["policy"]
    image_shape = conf.get("image_shape", [3, 128, 128])
    shape_meta = conf.get("shape_meta") or {
        "obs": {"img_obs_1": {"shape": image_shape, "minmax_shape": _RESOLVERS["image_minmax_01"](), "type": "rgb"},
                "img_goal_1": {"shape": image_shape, "minmax_shape": _RESOLVERS["image_minmax_01"](), "type": "rgb"}},
        "action": {"shape": [7], "minmax_shape": _RESOLVERS["lb_action_minmax"]()}}
    strip = lambda d: {k: v for k, v in d.items() if k != "_target_"}
    ns = _sched.DDPMScheduler(**strip(pc["noise_scheduler"]))
    nsd = _sched.DDIMScheduler(**strip(pc["noise_scheduler_ddim"]))
    oe = strip(pc["obs_encoder"])
    rm = strip(oe.pop("rgb_model"))
    rm.pop("input_shape", None)
    oe.pop("shape_meta", None)
    if widths is not None:
        rm = dict(rm)
        rm["backbone_kwargs"] = dict(rm["backbone_kwargs"], widths=tuple(widths))
    rgb_model = VisualCore(input_shape=image_shape, **rm)
    obs_encoder = MultiImageObsEncoder(shape_meta=shape_meta, rgb_model=rgb_model, **oe)
    skip = {"_target_", "noise_scheduler", "noise_scheduler_ddim", "obs_encoder", "shape_meta", "horizon", "n_action_steps", "n_obs_steps",
            "obs_as_global_cond"}
    kw = {k: v for k, v in pc.items() if k not in skip}
    return DiffusionUnetImagePolicy(shape_meta=shape_meta, noise_scheduler=ns, noise_scheduler_ddim=nsd, obs_encoder=obs_encoder,
                                    horizon=horizon if horizon is not None else conf.get("horizon", 16),
                                    n_action_steps=n_action_steps if n_action_steps is not None else conf.get("n_action_steps", 8),
                                    n_obs_steps=conf.get("n_obs_steps", 1), obs_as_global_cond=conf.get("obs_as_global_cond", True), **kw)


class Init_Diffusion_Policy:
    """a wrapper class to init diffusion policy (same attributes as the reference's)."""

    def __init__(self, args) -> None:
        self.args = args
        fname = getattr(args, "model_yl_path", None)
        conf = read_policy_yaml(fname) if fname and os.path.exists(fname) else DEFAULT_CONF
        self.all_conf = _ns(conf)
        self.policy_conf = self.all_conf.policy
        td = getattr(args, "trainer_dict", {}) or {}
        horizon = td.get("model_act_horizon", conf.get("horizon", 16))
        n_act = td.get("n_acts_per_pred", conf.get("n_action_steps", 8))
        self.policy_conf.horizon = horizon
        self.policy_conf.n_action_steps = n_act
        if hasattr(args, "input_img_size"):
            assert tuple(args.input_img_size) == tuple(conf.get("image_shape", [3, 128, 128])[1:])
        self.diffusion_policy: DiffusionUnetImagePolicy = build_policy(conf, horizon=horizon, n_action_steps=n_act)
