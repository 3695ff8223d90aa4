"""Unet_Libero with the reference's surface (flowdiffusion/flowdiffusion/unet.py:195-222): no-arg constructor, `.unet`
(UNetModel, 201,087,649 parameters), forward(x [B,24,H,W], t [B], task_embed [B,L,512]) -> [B,21,H,W].  Executed by
v2a_hip.unet_engine on HIP kernels (sampling); training goes through GoalGaussianDiffusion.forward -> v2a_hip.video_train."""
import torch
import torch.nn as nn
from .guided_diffusion.guided_diffusion.unet import UNetModel


class _HipUnetWrapper(nn.Module):
    def _engine(self):
        from v2a_hip.unet_engine import UNetEngine
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the MI355X-native video UNet runs on a HIP device only: call .to('cuda') first (no CPU fallback)")
        eng = self.__dict__.get("_eng")
        if eng is None or eng.device != dev:
            eng = UNetEngine(self.unet.engine_cfg(), {n: p for n, p in self.named_parameters()}, prefix="unet.")
            self.__dict__["_eng"] = eng
        import v2a_hip
        want = getattr(self, "storage", None)          # per-model override ('f32' / 'bf16' / 'fp16'); else the process-wide default
        if want is None:
            want = v2a_hip.get_video_storage()
            if want != "f32" and any((self.unet.model_channels * m) % 64 for m in self.unet.channel_mult):
                want = "f32"                           # widths the bf16 kernels cannot tile (e.g. the 32-channel test model)
        if eng.storage != want:
            eng.set_storage(want)
        return eng

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_eng", "_train_eng") else copy.deepcopy(v, memo)
        return new

    frame_channels = 3          # channels of one generated frame in the packed input/output (2 for optical-flow models)

    @torch.no_grad()
    def forward(self, x, t, task_embed=None, **kwargs):
        assert task_embed is not None, "must specify y if and only if the model is class-conditional"
        label = kwargs.pop("_label_emb", None)
        return self._engine().forward_libero(x, t, task_embed, label_emb=label, frame_ch=self.frame_channels)


class Unet_Libero(_HipUnetWrapper):
    def __init__(self):
        super().__init__()
        self.unet = UNetModel(image_size=(128, 128), in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2,
                              attention_resolutions=(8, 16), dropout=0, channel_mult=(1, 2, 3, 4, 5), conv_resample=True, dims=3,
                              num_classes=None, task_tokens=True, task_token_channels=512, use_checkpoint=False, use_fp16=False,
                              num_head_channels=32)


def _avdc(image_size, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(8, 16),
          channel_mult=(1, 2, 3, 4, 5)):
    return UNetModel(image_size=image_size, in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                     num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, dropout=0, channel_mult=channel_mult,
                     conv_resample=True, dims=3, num_classes=None, task_tokens=True, task_token_channels=512, use_checkpoint=False,
                     use_fp16=False, num_head_channels=32)


class UnetMW(_HipUnetWrapper):
    """MetaWorld checkpoint shape (reference unet.py:37-64): same hyper-parameters as Unet_Libero."""

    def __init__(self):
        super().__init__()
        self.unet = _avdc((128, 128))


class UnetThor_Luo(UnetMW):
    """reference unet.py:163-192"""


class UnetMWFlow(_HipUnetWrapper):
    """Optical-flow variant (reference unet.py:66-93): 2 flow channels per frame + the RGB conditioning image."""
    frame_channels = 2

    def __init__(self):
        super().__init__()
        self.unet = _avdc((128, 128), in_channels=5, out_channels=2)


class UnetThor(_HipUnetWrapper):
    """iTHOR checkpoint shape (reference unet.py:122-152): 64x64, 3 res-blocks, channel_mult (1,2,4), attention at 4/8."""

    def __init__(self):
        super().__init__()
        self.unet = _avdc((64, 64), num_res_blocks=3, attention_resolutions=(4, 8), channel_mult=(1, 2, 4))


class UnetBridge(_HipUnetWrapper):
    """Bridge checkpoint shape (reference unet.py:7-35): 48x64, 160 base channels, 3 res-blocks, channel_mult (1,2,4)."""

    def __init__(self):
        super().__init__()
        self.unet = _avdc((48, 64), model_channels=160, num_res_blocks=3, attention_resolutions=(4, 8), channel_mult=(1, 2, 4))


class Unet_Tiny(_HipUnetWrapper):
    """Small same-architecture model used by the parity tests (matches tools/ref_build.build_ref_unet(tiny=True))."""

    def __init__(self):
        super().__init__()
        self.unet = UNetModel(image_size=(32, 32), in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1,
                              attention_resolutions=(2,), dropout=0, channel_mult=(1, 2), conv_resample=True, dims=3, num_classes=None,
                              task_tokens=True, task_token_channels=512, use_checkpoint=False, use_fp16=False, num_head_channels=16)
