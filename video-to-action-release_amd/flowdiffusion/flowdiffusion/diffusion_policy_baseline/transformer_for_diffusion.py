"""TransformerForDiffusion with the reference's surface (diffusion_policy_baseline/transformer_for_diffusion.py:23-358): same
constructor keywords, parameter / buffer names (so its checkpoints load with strict=True), `forward(sample, timestep, cond)`,
`get_optim_groups`, `configure_optimizers`.  The torch.nn Transformer layers below are *parameter containers only*: forward and
backward run on HIP kernels (v2a_hip.transformer_engine) behind one autograd Function, so `loss.backward()` fills `.grad` as usual.
HIP device only (no CPU fallback).  Dropout (training mode): the same sites and rates as the torch layers (embeddings p_drop_emb;
attention probabilities, attention / feed-forward outputs and the feed-forward activation p_drop_attn), masks from a stateless
counter-based hash (`.dropout_seed`, default torch.initial_seed()) instead of torch's generator stream."""
from typing import Optional, Tuple, Union
import torch
import torch.nn as nn


def _additive(allowed: torch.Tensor) -> torch.Tensor:
    """bool [T,S] -> float mask with 0 where attention is allowed and -inf elsewhere (what nn.Transformer adds to the logits)."""
    out = torch.zeros(allowed.shape, dtype=torch.float32)
    return out.masked_fill(~allowed, float("-inf"))


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, sample, t, cond, *params):
        eng = module._engine()
        drop = None
        if module.training and max(module._p_emb, module._p_attn) > 0:
            if module.dropout_seed is None:
                module.dropout_seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
            drop = (module._p_emb, module._p_attn, module.dropout_seed)
        out, tape = eng.forward(sample, t, cond, drop)
        ctx.pack = (module, eng, tape, cond is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        module, eng, tape, has_cond = ctx.pack
        ctx.pack = None
        names = [n for n, _ in module.named_parameters()]
        P = dict(module.named_parameters())
        grads = {n: torch.zeros_like(P[n]) for n in names}
        ds, dc = eng.backward(tape, dout, grads, need_dsample=ctx.needs_input_grad[1], need_dcond=has_cond and ctx.needs_input_grad[3])
        return (None, ds, None, dc) + tuple(grads[n] for n in names)


class TransformerForDiffusion(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, horizon: int, n_obs_steps: int = None, cond_dim: int = 0, n_layer: int = 12,
                 n_head: int = 12, n_emb: int = 768, p_drop_emb: float = 0.1, p_drop_attn: float = 0.1, causal_attn: bool = False,
                 time_as_cond: bool = True, obs_as_cond: bool = False, n_cond_layers: int = 0) -> None:
        super().__init__()
        n_obs_steps = horizon if n_obs_steps is None else n_obs_steps
        obs_as_cond = cond_dim > 0                       # (the reference overrides the argument the same way, :51)
        T = horizon + (0 if time_as_cond else 1)         # the diffusion step is a trunk token when it is not a condition token
        T_cond = (1 if time_as_cond else 0)
        if obs_as_cond:
            assert time_as_cond
            T_cond += n_obs_steps
        self.input_emb = nn.Linear(input_dim, n_emb)
        self.pos_emb = nn.Parameter(torch.zeros(1, T, n_emb))
        self.drop = nn.Dropout(p_drop_emb)
        self.cond_obs_emb = nn.Linear(cond_dim, n_emb) if obs_as_cond else None
        self.cond_pos_emb, self.encoder, self.decoder = None, None, None
        layer_kw = dict(d_model=n_emb, nhead=n_head, dim_feedforward=4 * n_emb, dropout=p_drop_attn, activation="gelu", batch_first=True,
                        norm_first=True)
        self.encoder_only = T_cond == 0
        if self.encoder_only:
            self.encoder = nn.TransformerEncoder(nn.TransformerEncoderLayer(**layer_kw), num_layers=n_layer, enable_nested_tensor=False)
        else:
            self.cond_pos_emb = nn.Parameter(torch.zeros(1, T_cond, n_emb))
            if n_cond_layers > 0:
                self.encoder = nn.TransformerEncoder(nn.TransformerEncoderLayer(**layer_kw), num_layers=n_cond_layers,
                                                     enable_nested_tensor=False)
            else:
                self.encoder = nn.Sequential(nn.Linear(n_emb, 4 * n_emb), nn.Mish(), nn.Linear(4 * n_emb, n_emb))
            self.decoder = nn.TransformerDecoder(nn.TransformerDecoderLayer(**layer_kw), num_layers=n_layer)
        if causal_attn:
            i = torch.arange(T)
            self.register_buffer("mask", _additive(i[:, None] >= i[None, :]))           # token t sees tokens <= t
            if time_as_cond and obs_as_cond:
                s = torch.arange(T_cond)
                self.register_buffer("memory_mask", _additive(i[:, None] >= (s[None, :] - 1)))   # condition token 0 is the time step
            else:
                self.memory_mask = None
        else:
            self.mask, self.memory_mask = None, None
        self.ln_f = nn.LayerNorm(n_emb)
        self.head = nn.Linear(n_emb, output_dim)
        self.T, self.T_cond, self.horizon = T, T_cond, horizon
        self.time_as_cond, self.obs_as_cond = time_as_cond, obs_as_cond
        self._cfg = dict(n_emb=n_emb, n_head=n_head, n_layer=n_layer, n_cond_layers=n_cond_layers, encoder_only=self.encoder_only)
        self._p_emb, self._p_attn = float(p_drop_emb), float(p_drop_attn)
        self.dropout_seed = None
        self._init_all()

    def _init_all(self):
        """The reference's `_init_weights` (:152-195): N(0, 0.02) for Linear / attention weights and the position tables
        (`cond_pos_emb` only when observations are condition tokens), zero biases, unit LayerNorm."""
        for name, p in self.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if name == "pos_emb" or (name == "cond_pos_emb" and self.cond_obs_emb is not None):
                nn.init.normal_(p, mean=0.0, std=0.02)
            elif name == "cond_pos_emb":
                continue
            elif "norm" in name.split(".")[-2] or name.startswith("ln_f."):
                (nn.init.ones_ if leaf == "weight" else nn.init.zeros_)(p)
            elif p.dim() >= 2:
                nn.init.normal_(p, mean=0.0, std=0.02)
            else:
                nn.init.zeros_(p)

    # ------------------------------------------------------------------ optimiser groups (reference :197-263)
    def get_optim_groups(self, weight_decay: float = 1e-3):
        """Matrices of Linear / attention projections decay; biases, LayerNorm parameters and the position tables do not."""
        named = dict(self.named_parameters())
        decay = sorted(n for n, p in named.items() if p.dim() == 2)
        rest = sorted(n for n in named if n not in set(decay))
        return [{"params": [named[n] for n in decay], "weight_decay": weight_decay},
                {"params": [named[n] for n in rest], "weight_decay": 0.0}]

    def configure_optimizers(self, learning_rate: float = 1e-4, weight_decay: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.95)):
        return torch.optim.AdamW(self.get_optim_groups(weight_decay=weight_decay), lr=learning_rate, betas=betas)

    # ------------------------------------------------------------------ HIP execution
    def _engine(self):
        from v2a_hip.transformer_engine import TransformerEngine
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("TransformerForDiffusion runs on a HIP device only: call .to('cuda') first (no CPU fallback)")
        eng = self.__dict__.get("_eng")
        if eng is None or eng.device != dev or any(eng.P[n] is not p for n, p in params.items()):
            eng = TransformerEngine(self._cfg, params, {n: b.detach() for n, b in self.named_buffers()})
            eng._versions = None
            self.__dict__["_eng"] = eng
        ver = sum(p._version for p in params.values())
        if eng._versions != ver:
            eng.refresh_packs()
            eng._versions = ver
        return eng

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_eng" else copy.deepcopy(v, memo)
        return new

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], cond: Optional[torch.Tensor] = None, **kwargs):
        """sample (B,T,input_dim), timestep (B,) or scalar, cond (B,T',cond_dim) -> (B,T,output_dim)."""
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        elif t.dim() == 0:
            t = t[None]
        t = t.to(sample.device).long().expand(sample.shape[0]).contiguous()
        if self.obs_as_cond and cond is None:
            raise ValueError("this model was built with observation condition tokens: `cond` is required")
        params = [p for _, p in self.named_parameters()]
        return _Fn.apply(self, sample, t, cond if self.obs_as_cond else None, *params)
