"""Video-model training path: hand-written UNet backward vs torch autograd through the CPU oracle
(oracle/video_unet.py, itself pinned bit-exact against the reference's UNetModel forward)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny():
    from flowdiffusion.flowdiffusion.unet import Unet_Tiny
    from oracle.param_fill import fill_module
    torch.manual_seed(0)
    m = Unet_Tiny()
    sd = fill_module(m, seed=11)
    return m.to("cuda:0"), sd


def test_unet_backward_matches_oracle_autograd():
    import oracle.video_unet as OV
    from v2a_hip.unet_train import UNetTrainEngine
    m, sd = _tiny()
    cfg = OV.UNetCfg(in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
                     num_head_channels=16)
    g = torch.Generator().manual_seed(3)
    B, Fr, H, W = 2, 3, 32, 32
    x = torch.randn(B, 6, Fr, H, W, generator=g)
    t = torch.tensor([5, 77])
    y = torch.randn(B, 4, 512, generator=g)
    R = torch.randn(B, 3, Fr, H, W, generator=g)                      # d(loss)/d(out): loss = <out, R>

    # ---- oracle: autograd over the functional restatement, capturing the label embedding
    P = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}     # fp64 ground truth
    cap = {}
    orig = OV.label_embedding

    def spy(*a, **k):
        out = orig(*a, **k)
        out.retain_grad()
        cap["lab"] = out
        return out

    OV.label_embedding = spy
    orig_te = OV.timestep_embedding
    OV.timestep_embedding = lambda *a, **k: orig_te(*a, **k).double()
    keep_float = torch.Tensor.float
    torch.Tensor.float = lambda self, *a, **k: self if self.dtype == torch.float64 else keep_float(self, *a, **k)   # the oracle's
    try:                                                               # fp32 casts (GroupNorm32, softmax) would cap the truth at fp32
        out_ref = OV.unet_forward(P, x.double(), t, y.double(), cfg, pre="unet.")
        (out_ref * R.double()).sum().backward()
    finally:
        OV.label_embedding = orig
        OV.timestep_embedding = orig_te
        torch.Tensor.float = keep_float

    # ---- the same autograd in fp32 (what the reference itself computes): its distance from the fp64 truth is the yardstick for
    # gradients that are sums of thousands of cancelling terms (the emb_layers weights see a 3072-row column sum per sample)
    P32 = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    out32 = OV.unet_forward(P32, x, t, y, cfg, pre="unet.")
    (out32 * R).sum().backward()

    # ---- HIP: forward with tape + hand-written backward
    params = dict(m.named_parameters())
    eng = UNetTrainEngine(m.unet.engine_cfg(), params, prefix="unet.")
    xin = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
    out, tape = eng.forward_train_tokens(xin, t.cuda(), y.cuda())
    ref_cl = out_ref.detach().permute(0, 2, 3, 4, 1)
    assert ((out.cpu().double() - ref_cl).abs().max() / ref_cl.abs().max()).item() < 1e-5
    grads = {n: torch.zeros_like(p) for n, p in params.items()}
    dlab = eng.backward(tape, R.permute(0, 2, 3, 4, 1).contiguous().cuda(), grads)
    assert ((dlab.cpu().double() - cap["lab"].grad).abs().max() / cap["lab"].grad.abs().max()).item() < 1e-4
    text = lambda n: n.startswith("unet.task_attnpool")
    gmax = {tb: max(P[n].grad.abs().max().item() for n in P if P[n].grad is not None and text(n) == tb) for tb in (False, True)}
    worst, worst_name, checked = 0.0, None, 0
    for n in params:
        rg = P[n].grad
        assert rg is not None, n
        scale = max(rg.abs().max().item(), 1e-3 * gmax[text(n)])
        err = ((grads[n].cpu().double() - rg).abs().max() / scale).item()
        err32 = ((P32[n].grad.double() - rg).abs().max() / scale).item()
        checked += 1
        assert err <= max(1e-4, 4.0 * err32), (n, err, err32)          # 1e-4, or within 4x of the reference's own fp32 rounding
        if err > worst:
            worst, worst_name = err, n
    print("worst gradient error vs fp64 truth", worst, worst_name)
    assert checked > 100
