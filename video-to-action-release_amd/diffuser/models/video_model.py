"""Video_PredModel: the frozen goal-video generator the policy trainer queries (surface of the reference's
diffuser/models/video_model.py:9-85).

What the surface has to guarantee:
  * constructor `(diffusion_model, tokenizer, text_encoder, single_img_channels, results_folder)`;
  * `.ema` wraps the diffusion model with ema_pytorch's key layout (`online_model.*`, `ema_model.*`, `initted`, `step`), so the released
    `model-{milestone}.pt['ema']` loads with strict=True -- through ema_pytorch when it is installed, through `_EmaKeys` otherwise;
  * `.forward(x_conds [B,3,H,W] in [0,1], tasks)` -> `[B, frames, 3, H, W]` in [0,1] sampled by the EMA copy, where `tasks` is a list of
    task names (dashes / underscores become spaces before tokenisation) or, for CLIP-free benchmarks, pre-encoded tokens `[B,L,512]`;
  * `.video_future_horizon`, `.image_size`, `.load_trained_model(milestone)`.
The sampling itself is GoalGaussianDiffusion.sample on the HIP UNet engine."""
import copy
from pathlib import Path
import torch
import torch.nn as nn


class _EmaKeys(nn.Module):
    """Container with ema_pytorch.EMA's state-dict keys (no averaging logic: this model is only ever loaded and sampled)."""

    def __init__(self, model, **unused):
        super().__init__()
        self.online_model = model
        self.ema_model = copy.deepcopy(model).requires_grad_(False)
        self.register_buffer("initted", torch.Tensor([False]))
        self.register_buffer("step", torch.tensor([0]))


def _wrap_ema(model, **kw):
    try:
        from ema_pytorch import EMA
    except ImportError:
        return _EmaKeys(model, **kw)
    return EMA(model, **kw)


def _spaced(names):
    """'put-the_red mug' -> 'put the red mug' (reference helpers.py:29-50 applied for '-' then '_')."""
    return [n.replace("-", " ").replace("_", " ") for n in names]


class Video_PredModel(nn.Module):
    def __init__(self, diffusion_model, tokenizer, text_encoder, single_img_channels=3, results_folder="./results"):
        super().__init__()
        if not isinstance(diffusion_model, nn.Module):
            raise TypeError("diffusion_model must be an nn.Module")
        self.ema = _wrap_ema(diffusion_model, beta=0.995, update_every=10)
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.single_img_channels = single_img_channels
        self.results_folder = Path(results_folder)
        self.image_size = diffusion_model.image_size
        self.video_future_horizon = round(diffusion_model.channels / single_img_channels)

    @property
    def device(self):
        return next(self.ema.parameters()).device

    def load_trained_model(self, milestone):
        ckpt = torch.load(str(self.results_folder / f"model-{milestone}.pt"), map_location="cpu")
        self.ema.load_state_dict(ckpt["ema"], strict=True)
        if "version" in ckpt:
            print(f"loading from version {ckpt['version']}")

    def encode_batch_text(self, batch_text):
        tok = self.tokenizer(batch_text, return_tensors="pt", padding=True, truncation=True, max_length=128).to(self.device)
        return self.text_encoder(**tok).last_hidden_state

    def encode_rows(self, batch_text):
        """One tokeniser / text-encoder call PER string -> a list of [1, L_b, 512] features without padding: what the reference's bs = 1
        exploration loop feeds the sampler for each task (the text branch of the UNet takes no attention mask, so the pad-token states of a
        padded batch would leak into the shorter strings' conditioning)."""
        return [self.encode_batch_text([t]) for t in batch_text]

    def forward(self, x_conds, tasks, row_seeds=None):
        """row_seeds (extension): one sampler seed per row, see GoalGaussianDiffusion.sample."""
        n = x_conds.shape[0]
        if n != len(tasks):
            raise ValueError(f"{n} conditioning images for {len(tasks)} tasks")
        if torch.is_tensor(tasks):
            tokens = tasks.to(self.device)
        elif len(tasks) and torch.is_tensor(tasks[0]):
            tokens = [t.to(self.device) for t in tasks]          # per-row token features of different lengths (encode_rows): never padded
        else:
            tokens = self.encode_batch_text(_spaced(tasks))
        kw = {} if row_seeds is None else {"row_seeds": row_seeds}
        frames = self.ema.ema_model.sample(batch_size=n, x_cond=x_conds.to(self.device), task_embed=tokens, **kw)
        b, c, h, w = frames.shape
        return frames.view(b, c // self.single_img_channels, self.single_img_channels, h, w).detach()

    sample = forward
