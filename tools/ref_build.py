"""Builders for the reference's own modules (imported from /root/reference under tools/ref_shims).
Build-container only; used by tools/make_golden.py."""
import numpy as np
import torch
import tools.ref_shims as rs

rs.install()


def build_ref_policy(horizon=16, n_action_steps=8):
    """Mirror of Init_Diffusion_Policy (diffuser/diffusion_policy/get_dp.py:24-101) with the values of
    config/diff_policy/lb_train_diffusion_unet_image_orn10.yaml, without omegaconf."""
    from diffuser.diffusion_policy.diffusion_unet_image_policy import DiffusionUnetImagePolicy
    from diffuser.diffusion_policy.model.multi_image_obs_encoder import MultiImageObsEncoder
    from diffuser.diffusion_policy.common.vision_nets import VisualCore
    from diffuser.datasets import image_minmax_01_f, lb_action_minmax_f
    from diffusers.schedulers.scheduling_ddpm import DDPMScheduler
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler
    image_shape = [3, 128, 128]
    shape_meta = {
        "obs": {
            "img_obs_1": {"shape": image_shape, "minmax_shape": image_minmax_01_f(), "type": "rgb"},
            "img_goal_1": {"shape": image_shape, "minmax_shape": image_minmax_01_f(), "type": "rgb"},
        },
        "action": {"shape": [7], "minmax_shape": lb_action_minmax_f()},
    }
    ns = DDPMScheduler(num_train_timesteps=100, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2",
                       variance_type="fixed_small", clip_sample=True, prediction_type="epsilon")
    nsd = DDIMScheduler(num_train_timesteps=100, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2",
                        clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon")
    rgb = VisualCore(input_shape=image_shape, backbone_class="ResNet18Conv",
                     backbone_kwargs={"pretrained": None, "input_coord_conv": False},
                     pool_class="SpatialSoftmax",
                     pool_kwargs={"num_kp": 32, "learnable_temperature": False, "temperature": 1.0,
                                  "noise_std": 0.0, "output_variance": False},
                     flatten=True, feature_dimension=64)
    enc = MultiImageObsEncoder(shape_meta=shape_meta, rgb_model=rgb, resize_shape=None, crop_shape=None,
                               random_crop=None, use_group_norm=True, share_rgb_model=False, imagenet_norm=False)
    pol = DiffusionUnetImagePolicy(shape_meta=shape_meta, noise_scheduler=ns, noise_scheduler_ddim=nsd,
                                   obs_encoder=enc, horizon=horizon, n_action_steps=n_action_steps, n_obs_steps=1,
                                   num_inference_steps=100, num_inference_steps_ddim=8, obs_as_global_cond=True,
                                   diffusion_step_embed_dim=128, down_dims=[256, 512, 1024], kernel_size=5,
                                   n_groups=8, cond_predict_scale=True)
    return pol


def build_ref_unet(tiny=True):
    from flowdiffusion.flowdiffusion.guided_diffusion.guided_diffusion.unet import UNetModel
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    if not tiny:
        return Unet_Libero()
    m = Unet_Libero.__new__(Unet_Libero)
    torch.nn.Module.__init__(m)
    m.unet = UNetModel(image_size=(32, 32), in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1,
                       attention_resolutions=(2,), dropout=0, channel_mult=(1, 2), conv_resample=True, dims=3,
                       num_classes=None, task_tokens=True, task_token_channels=512, use_checkpoint=False,
                       use_fp16=False, num_head_channels=16)
    return m
