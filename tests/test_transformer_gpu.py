"""Transformer policy backbone on HIP kernels vs fixtures produced by the reference's TransformerForDiffusion and
vs the CPU oracle: forward, input gradients, every parameter gradient through `loss.backward()`.  fp32, 1e-4 relative."""
import numpy as np
import pytest
import torch

from test_oracle_golden import TRANSFORMER_CFGS

pytestmark = pytest.mark.gpu


def rel(a, b):
    from conftest import parity_record
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return parity_record("rel", ((a - b).abs().max() / (b.abs().max() + 1e-30)).item())


def _sample_idx(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (k,), generator=g)


@pytest.mark.parametrize("tag", list(TRANSFORMER_CFGS))
def test_transformer_forward_backward_vs_reference(golden_dir, tag):
    from flowdiffusion.flowdiffusion.diffusion_policy_baseline.transformer_for_diffusion import TransformerForDiffusion
    from oracle.param_fill import fill_module
    g = np.load(f"{golden_dir}/transformer.npz", allow_pickle=True)
    cfg = TRANSFORMER_CFGS[tag]
    torch.manual_seed(0)
    m = TransformerForDiffusion(**cfg)
    fill_module(m, seed=21)
    m = m.to("cuda:0").train()                                  # p = 0: training mode is allowed
    x = torch.from_numpy(g[f"{tag}_x"]).cuda().requires_grad_(True)
    cond = torch.from_numpy(g[f"{tag}_cond"]).cuda().requires_grad_(True) if f"{tag}_cond" in g else None
    t = torch.from_numpy(g[f"{tag}_t"]).cuda()
    y = m(x, t, cond)
    assert y.requires_grad and rel(y, g[f"{tag}_y"]) <= 1e-4, rel(y, g[f"{tag}_y"])
    (y * torch.from_numpy(g[f"{tag}_R"]).cuda()).sum().backward()
    assert rel(x.grad, g[f"{tag}_dx"]) <= 1e-4, rel(x.grad, g[f"{tag}_dx"])
    if cond is not None:
        assert rel(cond.grad, g[f"{tag}_dcond"]) <= 1e-4
    names = [str(n) for n in g[f"{tag}_names"]]
    P = dict(m.named_parameters())
    gmax = float(g[f"{tag}_grad_norms"].max())
    for i, n in enumerate(names):
        gr = P[n].grad
        assert gr is not None, n
        rn = float(g[f"{tag}_grad_norms"][i])
        assert abs(float(gr.double().norm()) - rn) <= 1e-4 * max(rn, 1e-3 * gmax), (n, float(gr.double().norm()), rn)
        smp = gr.flatten().cpu()[_sample_idx(gr.numel(), 6, 9)].numpy()
        ref = g[f"{tag}_grad_samples"][i]
        assert np.max(np.abs(smp - ref)) <= 1e-4 * max(np.abs(ref).max(), rn / np.sqrt(gr.numel()), 1e-3 * gmax / np.sqrt(gr.numel())), n
    # a second call after an optimiser step sees the new weights (packed operands are refreshed)
    opt = m.configure_optimizers(learning_rate=1e-2)
    opt.step()
    y2 = m(x.detach(), t, None if cond is None else cond.detach())
    assert rel(y2, y) > 1e-3


@pytest.mark.parametrize("tag", ["dec_causal", "bert_causal"])
def test_transformer_training_dropout_vs_oracle_with_same_masks(golden_dir, tag):
    """Training mode with the reference's default-style rates (p_drop_emb 0.1, p_drop_attn 0.2): the HIP masks are stateless hashes of
    (seed, site, element), so the oracle can be run with exactly the same masks (extracted by applying the dropout op to ones, site by
    site in forward order) -- outputs and every gradient must then agree to 1e-4; keep rates are checked statistically."""
    from flowdiffusion.flowdiffusion.diffusion_policy_baseline.transformer_for_diffusion import TransformerForDiffusion
    from oracle import transformer as OT
    from oracle.param_fill import fill_module
    from v2a_hip import ops
    g = np.load(f"{golden_dir}/transformer.npz", allow_pickle=True)
    cfg = dict(TRANSFORMER_CFGS[tag], p_drop_emb=0.1, p_drop_attn=0.2)
    torch.manual_seed(0)
    m = TransformerForDiffusion(**cfg)
    sd = fill_module(m, seed=21)
    m = m.to("cuda:0").train()
    m.dropout_seed = 1234567
    x = torch.from_numpy(g[f"{tag}_x"]).cuda().requires_grad_(True)
    cond = torch.from_numpy(g[f"{tag}_cond"]).cuda().requires_grad_(True) if f"{tag}_cond" in g else None
    t = torch.from_numpy(g[f"{tag}_t"]).cuda()
    R = torch.from_numpy(g[f"{tag}_R"])
    sid0 = m._engine()._sid
    y = m(x, t, cond)
    assert rel(y, g[f"{tag}_y"]) > 1e-2                           # dropout is really active
    (y * R.cuda()).sum().backward()
    n_sites = m._engine()._sid - sid0
    # ---- oracle with the same masks
    state = {"sid": sid0, "kept": 0.0, "total": 0}

    def masks(shape, kind):
        state["sid"] += 1
        p = cfg["p_drop_emb"] if kind == "emb" else cfg["p_drop_attn"]
        mk = ops.dropout(torch.ones(shape, device="cuda:0"), p, m.dropout_seed, state["sid"]).cpu()
        if kind != "emb":
            state["kept"] += float((mk > 0).sum())
            state["total"] += mk.numel()
            assert torch.all((mk == 0) | ((mk - 1.0 / (1.0 - p)).abs() < 1e-6))
        return mk

    names = [n for n, _ in m.named_parameters()]
    P = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    xo = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    co = torch.from_numpy(g[f"{tag}_cond"]).requires_grad_(True) if cond is not None else None
    yo = OT.forward(P, xo, torch.from_numpy(g[f"{tag}_t"]), co, cfg["n_head"], cfg["n_layer"], cfg.get("n_cond_layers", 0), m.encoder_only, masks)
    assert state["sid"] - sid0 == n_sites                         # same number of dropout sites, same order
    assert abs(state["kept"] / state["total"] - 0.8) < 0.02
    assert rel(y, yo) <= 1e-4, rel(y, yo)
    (yo * R).sum().backward()
    assert rel(x.grad, xo.grad) <= 1e-4
    if cond is not None:
        assert rel(cond.grad, co.grad) <= 1e-4
    Pm = dict(m.named_parameters())
    gmax = max(float(P[n].grad.abs().max()) for n in names)
    for n in names:
        err = float((Pm[n].grad.cpu().double() - P[n].grad.double()).abs().max())
        assert err <= 1e-4 * max(float(P[n].grad.abs().max()), 1e-3 * gmax), (n, err)
    y2 = m(x.detach(), t, None if cond is None else cond.detach())          # a second call draws new masks
    assert rel(y2, y) > 1e-3
    m.eval()
    assert rel(m(x.detach(), t, None if cond is None else cond.detach()), g[f"{tag}_y"]) <= 1e-4


def test_transformer_eval_scalar_timestep_and_default_dropout():
    from flowdiffusion.flowdiffusion.diffusion_policy_baseline.transformer_for_diffusion import TransformerForDiffusion
    torch.manual_seed(0)
    m = TransformerForDiffusion(input_dim=4, output_dim=4, horizon=10, n_obs_steps=3, cond_dim=512, n_cond_layers=2, n_layer=8, n_head=8,
                                n_emb=384, causal_attn=True, time_as_cond=True, obs_as_cond=True).to("cuda:0")   # TransformerNet's trunk
    x, c = torch.randn(5, 10, 4, device="cuda:0"), torch.randn(5, 3, 512, device="cuda:0")
    tr = m(x, 3, c)                                             # default p_drop = 0.1 in training mode
    assert torch.isfinite(tr).all()
    m.eval()
    with torch.no_grad():
        a = m(x, 3, c)
        b = m(x, torch.full((5,), 3, device="cuda:0"), c)
    assert a.shape == (5, 10, 4) and torch.equal(a, b) and torch.isfinite(a).all()


def test_transformer_net_sized_trunk_backward_vs_oracle():
    """The trunk TransformerNet instantiates (diffusion_policy_baseline/unet.py:57-70: 8 decoder + 2 encoder layers, 384 wide, 8 heads of
    48, 10 action tokens, 3 + 1 condition tokens of 512): forward and every gradient against the (reference-pinned) CPU oracle."""
    from flowdiffusion.flowdiffusion.diffusion_policy_baseline.transformer_for_diffusion import TransformerForDiffusion
    from oracle import transformer as OT
    from oracle.param_fill import fill_module
    cfg = dict(input_dim=4, output_dim=4, horizon=10, n_obs_steps=3, cond_dim=512, n_cond_layers=2, n_layer=8, n_head=8, n_emb=384,
               p_drop_emb=0.0, p_drop_attn=0.0, causal_attn=True, time_as_cond=True, obs_as_cond=True)
    torch.manual_seed(0)
    m = TransformerForDiffusion(**cfg)
    sd = fill_module(m, seed=41)
    m = m.to("cuda:0").train()
    g = torch.Generator().manual_seed(9)
    B = 6
    x, c = torch.randn(B, 10, 4, generator=g), torch.randn(B, 3, 512, generator=g)
    t, R = torch.randint(0, 100, (B,), generator=g), torch.randn(B, 10, 4, generator=g)
    xg, cg = x.cuda().requires_grad_(True), c.cuda().requires_grad_(True)
    y = m(xg, t.cuda(), cg)
    (y * R.cuda()).sum().backward()
    names = [n for n, _ in m.named_parameters()]
    P = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    xo, co = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
    yo = OT.forward(P, xo, t, co, 8, 8, 2, False)
    (yo * R).sum().backward()
    assert rel(y, yo) <= 1e-4, rel(y, yo)
    assert rel(xg.grad, xo.grad) <= 2e-4 and rel(cg.grad, co.grad) <= 2e-4
    Pm = dict(m.named_parameters())
    gmax = max(float(P[n].grad.abs().max()) for n in names)
    for n in names:
        err = float((Pm[n].grad.cpu().double() - P[n].grad.double()).abs().max())
        assert err <= 2e-4 * max(float(P[n].grad.abs().max()), 1e-3 * gmax), (n, err)
