"""Video_PredModel with the reference's surface (diffuser/models/video_model.py:9-85): holds an EMA container around the
diffusion model (so released `model-{milestone}.pt['ema']` state dicts with ema_pytorch's key layout load with strict=True),
CLIP tokenizer / text encoder handles, and forward(x_conds [B,3,H,W] in [0,1], tasks: list[str]) -> [B,7,3,H,W] in [0,1]."""
import copy
from pathlib import Path
import torch
import torch.nn as nn


class _EMAContainer(nn.Module):
    """ema_pytorch.EMA key layout: online_model.*, ema_model.*, initted, step.  Used when ema_pytorch is not installed."""

    def __init__(self, model, beta=0.995, update_every=10, **kw):
        super().__init__()
        self.online_model = model
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.register_buffer("initted", torch.Tensor([False]))
        self.register_buffer("step", torch.tensor([0]))


def _make_ema(model, **kw):
    try:
        from ema_pytorch import EMA
        return EMA(model, **kw)
    except ImportError:
        return _EMAContainer(model, **kw)


def _no_dash(tasks):
    return [" ".join(t.split("-")) for t in tasks]


def _no_underscore(tasks):
    return [" ".join(t.split("_")) for t in tasks]


class Video_PredModel(nn.Module):
    def __init__(self, diffusion_model, tokenizer, text_encoder, single_img_channels=3, results_folder="./results"):
        super().__init__()
        assert isinstance(diffusion_model, nn.Module)
        self.ema = _make_ema(diffusion_model, beta=0.995, update_every=10)
        self.tokenizer = tokenizer
        self.text_encoder = text_encoder
        self.single_img_channels = single_img_channels
        self.image_size = diffusion_model.image_size
        self.results_folder = Path(results_folder)
        self.video_future_horizon = round(diffusion_model.channels / single_img_channels)

    def load_trained_model(self, milestone):
        data = torch.load(str(self.results_folder / f"model-{milestone}.pt"), map_location="cpu")
        self.ema.load_state_dict(data["ema"], strict=True)
        if "version" in data:
            print(f"loading from version {data['version']}")

    def encode_batch_text(self, batch_text):
        ids = self.tokenizer(batch_text, return_tensors="pt", padding=True, truncation=True, max_length=128).to(self.device)
        return self.text_encoder(**ids).last_hidden_state

    def sample(self, x_conds, tasks):
        assert x_conds.shape[0] == len(tasks)
        bs = x_conds.shape[0]
        x_conds = x_conds.to(self.device)
        if torch.is_tensor(tasks):
            emb = tasks.to(self.device)            # pre-encoded task tokens [B,L,512] (benchmarks: CLIP-free)
        else:
            emb = self.encode_batch_text(_no_underscore(_no_dash(tasks))).to(self.device)
        out = self.ema.ema_model.sample(batch_size=bs, x_cond=x_conds, task_embed=emb)
        B, C, H, W = out.shape
        return out.view(B, C // self.single_img_channels, self.single_img_channels, H, W).detach()

    def forward(self, x_conds, tasks):
        return self.sample(x_conds, tasks)

    @property
    def device(self):
        return next(self.ema.parameters()).device
