"""CPU restatement of TransformerForDiffusion.forward.  TEST INFRASTRUCTURE (only tests/ may import it).

Follows /root/reference/flowdiffusion/flowdiffusion/diffusion_policy_baseline/transformer_for_diffusion.py:283-358 (forward) over the
torch.nn.TransformerEncoderLayer / TransformerDecoderLayer arithmetic it instantiates at :75-110 (norm_first=True, activation='gelu'
(erf form), batch_first=True, additive float masks :128-150; dropout off), SinusoidalPosEmb positional_embedding.py:10-17.
Functional over a state dict, differentiable by torch autograd; pinned on fixtures produced by the reference class itself
(tests/golden/transformer.npz, tools/make_golden.py g_transformer).
`masks`: optional callable(site_tensor_shape) -> multiplier tensor (0 or 1/(1-p)), called once per dropout site in forward order
(embedding dropouts, then per layer: attention probabilities, attention output, [cross-attention probabilities, output,]
feed-forward activation, feed-forward output) -- the order torch's layers apply them in; with masks=None dropout is off."""
import math
import torch
import torch.nn.functional as F


def sinusoidal(t, dim):
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    a = t.float()[:, None] * f[None, :]
    return torch.cat([a.sin(), a.cos()], dim=-1)


def _lin(P, pre, x):
    return x @ P[pre + ".weight"].T + P[pre + ".bias"]


def _ln(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[pre + ".weight"], P[pre + ".bias"], 1e-5)


def _drop(x, masks, kind):
    return x if masks is None else x * masks(tuple(x.shape), kind)


def mha(P, pre, xq, xkv, mask, H, masks=None):
    E = xq.shape[-1]
    W, b = P[pre + ".in_proj_weight"], P[pre + ".in_proj_bias"]
    q = xq @ W[:E].T + b[:E]
    k = xkv @ W[E:2 * E].T + b[E:2 * E]
    v = xkv @ W[2 * E:].T + b[2 * E:]
    B, T, _ = q.shape
    S, D = k.shape[1], E // H
    q, k, v = (z.view(B, -1, H, D).transpose(1, 2) for z in (q, k, v))
    att = q @ k.transpose(-1, -2) / math.sqrt(D)
    if mask is not None:
        att = att + mask
    o = (_drop(att.softmax(dim=-1), masks, "attn") @ v).transpose(1, 2).reshape(B, T, E)
    return _lin(P, pre + ".out_proj", o)


def _ff(P, pre, x, masks=None):
    return _lin(P, pre + ".linear2", _drop(F.gelu(_lin(P, pre + ".linear1", x)), masks, "layer"))


def enc_layer(P, pre, x, mask, H, masks=None):
    h = _ln(P, pre + ".norm1", x)
    x = x + _drop(mha(P, pre + ".self_attn", h, h, mask, H, masks), masks, "layer")
    return x + _drop(_ff(P, pre, _ln(P, pre + ".norm2", x), masks), masks, "layer")


def dec_layer(P, pre, x, mem, mask, mem_mask, H, masks=None):
    h = _ln(P, pre + ".norm1", x)
    x = x + _drop(mha(P, pre + ".self_attn", h, h, mask, H, masks), masks, "layer")
    x = x + _drop(mha(P, pre + ".multihead_attn", _ln(P, pre + ".norm2", x), mem, mem_mask, H, masks), masks, "layer")
    return x + _drop(_ff(P, pre, _ln(P, pre + ".norm3", x), masks), masks, "layer")


def forward(P, sample, t, cond, n_head, n_layer, n_cond_layers, encoder_only, masks=None):
    """P: state dict (parameters + `mask` / `memory_mask` buffers when causal)."""
    E = P["pos_emb"].shape[-1]
    temb = sinusoidal(t, E).to(sample.dtype)[:, None]
    inp = _lin(P, "input_emb", sample)
    mask, mem_mask = P.get("mask"), P.get("memory_mask")
    if encoder_only:
        x = torch.cat([temb, inp], dim=1)
        x = _drop(x + P["pos_emb"][:, :x.shape[1]], masks, "emb")
        for li in range(n_layer):
            x = enc_layer(P, f"encoder.layers.{li}", x, mask, n_head, masks)
        x = x[:, 1:]
    else:
        ce = temb
        if cond is not None:
            ce = torch.cat([ce, _lin(P, "cond_obs_emb", cond)], dim=1)
        m = _drop(ce + P["cond_pos_emb"][:, :ce.shape[1]], masks, "emb")
        if n_cond_layers > 0:
            for li in range(n_cond_layers):
                m = enc_layer(P, f"encoder.layers.{li}", m, None, n_head, masks)
        else:
            m = _lin(P, "encoder.2", F.mish(_lin(P, "encoder.0", m)))
        x = _drop(inp + P["pos_emb"][:, :inp.shape[1]], masks, "emb")
        for li in range(n_layer):
            x = dec_layer(P, f"decoder.layers.{li}", x, m, mask, mem_mask, n_head, masks)
    return _lin(P, "head", _ln(P, "ln_f", x))
