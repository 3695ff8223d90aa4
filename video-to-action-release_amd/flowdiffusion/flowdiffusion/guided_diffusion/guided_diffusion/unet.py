"""Parameter shell of the guided-diffusion UNetModel in its pseudo-3D video form (reference
guided_diffusion/guided_diffusion/unet.py:71-145 blocks, :148-260 ResBlock, :263-309 AttentionBlock, :404-684 UNetModel).
Construction order, module names and initialisers follow the reference so released checkpoints
(`ema_model.model.unet.input_blocks.1.0.in_layers.2.spatial_conv.weight` ...) load unchanged; forward() is executed by
v2a_hip.unet_engine.UNetEngine on hand-written HIP kernels.  Libero-path options only (no class labels, no scale-shift
norm, no resblock up/down, legacy attention order, conv resampling)."""
from types import SimpleNamespace
import torch
import torch.nn as nn
from .nn import conv_nd, normalization
from .imagen import PerceiverResampler


class TimestepEmbedSequential(nn.Sequential):
    pass


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=1)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise NotImplementedError("average-pool downsampling is not on the Libero path (conv_resample=True)")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2 if dims != 3 else (1, 2, 2), padding=1)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv:
            raise NotImplementedError("scale-shift norm / resblock up-down are not on the Libero path")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if self.out_channels == channels else conv_nd(dims, channels, self.out_channels, 1)


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        if use_new_attention_order:
            raise NotImplementedError("only the legacy head order (QKVAttentionLegacy) is on the Libero path")
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = conv_nd(1, channels, channels * 3, 1)
        self.proj_out = conv_nd(1, channels, channels, 1)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, task_tokens=True, task_token_channels=512,
                 use_checkpoint=False, use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False):
        super().__init__()
        if dims != 3 or num_classes is not None or not task_tokens or resblock_updown or use_fp16 or dropout != 0 or num_head_channels == -1:
            raise NotImplementedError("HIP UNetModel implements the AVDC video configuration (dims=3, task tokens, head channels)")
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, tuple(attention_resolutions), tuple(channel_mult)
        self.num_head_channels, self.task_tokens, self.num_classes = num_head_channels, task_tokens, num_classes
        self.dtype = torch.float32
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.task_attnpool = nn.Sequential(PerceiverResampler(dim=task_token_channels, depth=2), nn.Linear(task_token_channels, ted))
        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, ch, 3, padding=1))])
        chans = [ch]
        ds = 1
        att = dict(use_checkpoint=use_checkpoint, num_heads=num_heads, num_head_channels=num_head_channels)
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, out_channels=int(mult * model_channels), dims=dims)]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, **att))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, dropout, dims=dims), AttentionBlock(ch, **att),
                                                    ResBlock(ch, ted, dropout, dims=dims))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=int(model_channels * mult), dims=dims)]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, **att))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), conv_nd(dims, input_ch, out_channels, 3, padding=1))

    def engine_cfg(self):
        pr = self.task_attnpool[0].cfg
        return SimpleNamespace(in_channels=self.in_channels, model_channels=self.model_channels, out_channels=self.out_channels,
                               num_res_blocks=self.num_res_blocks, attention_resolutions=self.attention_resolutions,
                               channel_mult=self.channel_mult, num_head_channels=self.num_head_channels, pr_depth=pr["depth"],
                               pr_dim_head=pr["dim_head"], pr_heads=pr["heads"], pr_num_latents=pr["num_latents"],
                               pr_num_mean_pooled=pr["num_mean_pooled"], pr_ff_mult=pr["ff_mult"])
