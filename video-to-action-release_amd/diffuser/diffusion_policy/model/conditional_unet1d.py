"""Parameter shell of ConditionalUnet1D (reference model/conditional_unet1d.py:14-246), Libero configuration:
global conditioning only, cond_predict_scale (FiLM), down/up sampling enabled.  Registration order of the sub-modules
follows the reference so that state_dict / parameters() orders coincide."""
import torch
import torch.nn as nn
from .conv1d_components import Downsample1d, Upsample1d, Conv1dBlock


class _CondEncoder(nn.Sequential):
    pass


class ConditionalResidualBlock1D(nn.Module):
    def __init__(self, in_channels, out_channels, cond_dim, kernel_size=3, n_groups=8, cond_predict_scale=False):
        super().__init__()
        self.blocks = nn.ModuleList([Conv1dBlock(in_channels, out_channels, kernel_size, n_groups=n_groups),
                                     Conv1dBlock(out_channels, out_channels, kernel_size, n_groups=n_groups)])
        if not cond_predict_scale:
            raise NotImplementedError("only cond_predict_scale=True (FiLM) is on the Libero path")
        self.cond_predict_scale = True
        self.out_channels = out_channels
        self.cond_encoder = _CondEncoder(nn.Mish(), nn.Linear(cond_dim, out_channels * 2), nn.Identity())
        self.residual_conv = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()


class ConditionalUnet1D(nn.Module):
    def __init__(self, input_dim, local_cond_dim=None, global_cond_dim=None, diffusion_step_embed_dim=256,
                 down_dims=(256, 512, 1024), kernel_size=3, n_groups=8, cond_predict_scale=False, cond_unet1d_config={}):
        super().__init__()
        if local_cond_dim is not None or cond_unet1d_config.get("no_down_up", False):
            raise NotImplementedError("local conditioning / no_down_up are not on the Libero path")
        all_dims = [input_dim] + list(down_dims)
        dsed = diffusion_step_embed_dim
        diffusion_step_encoder = nn.Sequential(nn.Identity(), nn.Linear(dsed, dsed * 4), nn.Mish(), nn.Linear(dsed * 4, dsed))
        cond_dim = dsed + (global_cond_dim or 0)
        in_out = list(zip(all_dims[:-1], all_dims[1:]))
        kw = dict(cond_dim=cond_dim, kernel_size=kernel_size, n_groups=n_groups, cond_predict_scale=cond_predict_scale)
        mid_dim = all_dims[-1]
        self.mid_modules = nn.ModuleList([ConditionalResidualBlock1D(mid_dim, mid_dim, **kw),
                                          ConditionalResidualBlock1D(mid_dim, mid_dim, **kw)])
        down_modules = nn.ModuleList([])
        for ind, (din, dout) in enumerate(in_out):
            is_last = ind >= (len(in_out) - 1)
            down_modules.append(nn.ModuleList([ConditionalResidualBlock1D(din, dout, **kw), ConditionalResidualBlock1D(dout, dout, **kw),
                                               Downsample1d(dout) if not is_last else nn.Identity()]))
        up_modules = nn.ModuleList([])
        for ind, (din, dout) in enumerate(reversed(in_out[1:])):
            up_modules.append(nn.ModuleList([ConditionalResidualBlock1D(dout * 2, din, **kw), ConditionalResidualBlock1D(din, din, **kw),
                                             Upsample1d(din)]))
        start_dim = down_dims[0]
        final_conv = nn.Sequential(Conv1dBlock(start_dim, start_dim, kernel_size=kernel_size), nn.Conv1d(start_dim, input_dim, 1))
        self.diffusion_step_encoder = diffusion_step_encoder
        self.local_cond_encoder = None
        self.up_modules = up_modules
        self.down_modules = down_modules
        self.final_conv = final_conv

    @torch.no_grad()
    def forward(self, sample, timestep, local_cond=None, global_cond=None, **kwargs):
        owner = object.__getattribute__(self, "__dict__").get("_owner_ref")
        if owner is None or owner() is None:
            raise RuntimeError("ConditionalUnet1D.forward needs its owning DiffusionUnetImagePolicy (HIP engine)")
        return owner()._unet_forward(sample, timestep, global_cond)
