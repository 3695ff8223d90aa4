"""Per-kernel parity: every HIP op (through the C ABI) against a plain torch-CPU fp32 restatement of the same op.
Tolerance: 1e-4 relative to the tensor's max magnitude (north_star's fp32 budget), typically met at ~1e-6."""
import math
import os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def close(a, b, tol=1e-4, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    from conftest import parity_record
    parity_record(what or "close", err / scale, tol)
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def nhwc(x):  # NCHW cpu -> NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().to(dev())


def nchw(y):
    return y.permute(0, 3, 1, 2).contiguous().cpu()


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 16, 16, 64, 3, 1, 1),
    (2, 64, 16, 16, 128, 3, 2, 1),
    (3, 128, 9, 7, 96, 1, 1, 0),
    (2, 3, 32, 32, 64, 7, 2, 3),       # scalar gather path (Cin % 16 != 0)
    (1, 6, 12, 12, 32, 3, 1, 1),
    (2, 256, 8, 8, 512, 3, 1, 1),
    (4, 32, 64, 64, 3, 3, 1, 1),       # Cout = 3 (final video conv)
    (1, 512, 4, 4, 512, 3, 1, 1),      # small M, large K -> split-K
    (40, 48, 20, 20, 80, 3, 1, 1),     # M > 4096 -> 128-row tiles, ragged N
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_and_grads(case):
    from v2a_hip import ops
    N, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wp = ops.pack_weight(w.detach().to(dev()), 0)
    yd = ops.conv2d(nhwc(x.detach()), wp, b.to(dev()), Cout, k, k, (s, s), (p, p))
    close(nchw(yd), y, what="conv fwd")
    # weight gradient
    dw = ops.conv2d_wgrad(nhwc(x.detach()), nhwc(dy), tuple(w.shape), k, k, (s, s), (p, p))
    close(dw, w.grad, what="conv wgrad")
    # data gradient = same kernel, flipped/transposed pack, input dilation = stride
    wd = ops.pack_weight(w.detach().to(dev()), 1)
    dx = ops.conv2d(nhwc(dy), wd, None, Cin, k, k, (1, 1), (k - 1 - p, k - 1 - p), idil=s, out_hw=(H, W))
    close(nchw(dx), x.grad, what="conv dgrad")


def test_conv_epilogue_concat_upsample_split():
    from v2a_hip import ops
    g = torch.Generator().manual_seed(7)
    N, C1, C2, H, W, Cout = 4, 32, 48, 6, 6, 64
    x1 = torch.randn(N, C1, H, W, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / 20
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(2, Cout, generator=g)          # per-batch vector, 2 "batches" of 2 images
    xin = torch.cat([x1, x2], 1)
    up = F.interpolate(xin, scale_factor=2, mode="nearest")
    res = torch.randn(N, Cout, 2 * H, 2 * W, generator=g)
    ref = F.conv2d(up, w, b, padding=1) + rv.repeat_interleave(2, 0)[:, :, None, None] + res
    wp = ops.pack_weight(w.to(dev()), 0)
    y = ops.conv2d(nhwc(x1), wp, b.to(dev()), Cout, 3, 3, (1, 1), (1, 1), x2=nhwc(x2), rowvec=rv.to(dev()),
                   rows_per_batch=2 * (2 * H) * (2 * W), residual=nhwc(res), ups=True)
    close(nchw(y), ref, what="concat+upsample+epilogue")
    # split output channels into two buffers
    ya = torch.empty(N, 2 * H, 2 * W, 24, device=dev())
    yb = torch.empty(N, 2 * H, 2 * W, Cout - 24, device=dev())
    ops.conv2d(nhwc(x1), wp, b.to(dev()), Cout, 3, 3, (1, 1), (1, 1), x2=nhwc(x2), ups=True, y=ya, y2=yb, csplit=24)
    ref2 = F.conv2d(up, w, b, padding=1)
    close(nchw(ya), ref2[:, :24], what="csplit a")
    close(nchw(yb), ref2[:, 24:], what="csplit b")


def test_temporal_conv_as_3x1_view():
    """Conv3d temporal part: Conv1d(k=3) over frames with zero pad 1+1 == (3x1) conv on [B, F, (H W), C]."""
    from v2a_hip import ops
    g = torch.Generator().manual_seed(3)
    B, C, Fr, H, W = 2, 32, 5, 4, 6
    x = torch.randn(B, C, Fr, H, W, generator=g)
    w = torch.randn(C, C, 3, generator=g) / 10
    b = torch.randn(C, generator=g)
    z = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, Fr)
    ref = F.conv1d(F.pad(z, (1, 1)), w, b).reshape(B, H, W, C, Fr).permute(0, 3, 4, 1, 2)
    xcl = x.permute(0, 2, 3, 4, 1).contiguous().to(dev())            # [B,F,H,W,C]
    wp = ops.pack_weight(w.to(dev()), 0)                              # [Cout][1][3][Cin] == [Cout][3][1][Cin]
    y = ops.conv2d(xcl.view(B, Fr, H * W, C), wp, b.to(dev()), C, 3, 1, (1, 1), (1, 0))
    y = y.view(B, Fr, H, W, C).permute(0, 4, 1, 2, 3).cpu()
    close(y, ref, what="temporal conv")


def test_conv1d_and_transposed():
    from v2a_hip import ops
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, T = 8, 64, 96, 16
    x = torch.randn(B, Cin, T, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 5, generator=g) / 18).requires_grad_(True)
    b = torch.randn(Cout, generator=g)
    y = F.conv1d(x, w, b, padding=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = x.detach().permute(0, 2, 1).contiguous().to(dev()).view(B, 1, T, Cin)
    yd = ops.conv2d(xd, ops.pack_weight(w.detach().to(dev()), 0), b.to(dev()), Cout, 1, 5, (1, 1), (0, 2))
    close(yd.view(B, T, Cout).permute(0, 2, 1), y, what="conv1d")
    dyd = dy.permute(0, 2, 1).contiguous().to(dev()).view(B, 1, T, Cout)
    dw = ops.conv2d_wgrad(xd, dyd, tuple(w.shape), 1, 5, (1, 1), (0, 2))
    close(dw, w.grad, what="conv1d wgrad")
    # ConvTranspose1d(k4, s2, p1): forward = dgrad-style conv (pack mode 1, idil 2, pad k-1-p)
    wt = (torch.randn(Cin, Cout, 4, generator=g) / 16).requires_grad_(True)     # torch layout [Cin][Cout][K]
    bt = torch.randn(Cout, generator=g)
    x2 = torch.randn(B, Cin, 8, generator=g, requires_grad=True)
    yt = F.conv_transpose1d(x2, wt, bt, stride=2, padding=1)
    dyt = torch.randn(yt.shape, generator=g)
    yt.backward(dyt)
    x2d = x2.detach().permute(0, 2, 1).contiguous().to(dev()).view(B, 1, 8, Cin)
    wtp = ops.pack_weight(wt.detach().to(dev()), 1)      # regular-conv view: [Cout'=Cin][Cin'=Cout][K] -> dgrad pack
    ytd = ops.conv2d(x2d, wtp, bt.to(dev()), Cout, 1, 4, (1, 1), (0, 2), idil=2, out_hw=(1, 16))
    close(ytd.view(B, 16, Cout).permute(0, 2, 1), yt, what="convtranspose1d fwd")
    # its data gradient: regular stride-2 conv over dy with the forward pack of the same tensor
    dytd = dyt.permute(0, 2, 1).contiguous().to(dev()).view(B, 1, 16, Cout)
    dx2 = ops.conv2d(dytd, ops.pack_weight(wt.detach().to(dev()), 0), None, Cin, 1, 4, (1, 2), (0, 1))
    close(dx2.view(B, 8, Cin).permute(0, 2, 1), x2.grad, what="convtranspose1d dgrad")
    # its weight gradient: wgrad of the regular conv with roles swapped (input = dy, output-grad = x)
    dwt = ops.conv2d_wgrad(dytd, x2d, tuple(wt.shape), 1, 4, (1, 2), (0, 1))
    close(dwt, wt.grad, what="convtranspose1d wgrad")


def test_linear_and_colsum():
    from v2a_hip import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(64, 256, generator=g)
    w = torch.randn(2048, 256, generator=g) / 16
    b = torch.randn(2048, generator=g)
    y = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()))
    close(y, F.linear(x, w, b), what="linear")
    close(ops.colsum(x.to(dev())), x.sum(0), what="colsum")


GN_CASES = [
    # N, S, C, G, act, residual, film
    (2, 7 * 32 * 32, 128, 32, "silu", False, False),     # large path (video resblock)
    (3, 64 * 64, 64, 4, "relu", True, False),            # large path + residual (resnet)
    (4, 257, 384, 32, "silu", False, False),             # large path, ragged rows, L4 not dividing 256
    (8, 16, 256, 8, "mish", False, True),                # small path + FiLM (conv1d block)
    (8, 4, 1024, 8, "mish", False, False),
    (6, 16, 512, 32, "relu", True, False),               # resnet layer4 small path + residual (float4 wave kernel, one row per lane)
    (5, 64, 256, 16, "relu", True, False),               # resnet layer3: float4 wave kernel, four rows per lane
    (3, 64, 256, 16, "relu", False, False),
    (4, 8, 512, 8, "mish", False, True),                 # Conv1dBlock widths: 64 channels per group, two passes, FiLM
    (3, 16, 512, 8, "mish", True, True),                 # ... four passes
    (5, 8, 1024, 8, "mish", False, True),                # 128 channels per group, four passes
    (5, 64, 640, 32, "none", False, False),              # attention norm (per frame)
]


@pytest.mark.parametrize("case", GN_CASES)
def test_groupnorm_fwd_bwd(case):
    from v2a_hip import ops
    N, S, C, G, act, use_res, use_film = case
    g = torch.Generator().manual_seed(N * S + C)
    x = (torch.randn(N, C, S, generator=g) * 1.7 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    res = torch.randn(N, C, S, generator=g).requires_grad_(True) if use_res else None
    film = torch.randn(N, 2, C, generator=g).requires_grad_(True) if use_film else None
    z = F.group_norm(x, G, gamma, beta, eps=1e-5)
    if use_res:
        z = z + res
    a = {"silu": F.silu, "relu": F.relu, "mish": F.mish, "none": lambda v: v}[act](z)
    if use_film:
        a = film[:, 0, :, None] * a + film[:, 1, :, None]
    dout = torch.randn(a.shape, generator=g)
    a.backward(dout)
    cl = lambda t: t.detach().permute(0, 2, 1).contiguous().to(dev())
    xd = cl(x)
    yd, mean, rstd = ops.groupnorm_fwd(xd, gamma.detach().to(dev()), beta.detach().to(dev()), G, act,
                                       residual=cl(res) if use_res else None,
                                       film=film.detach().to(dev()) if use_film else None)
    close(yd.permute(0, 2, 1), a, what="gn fwd")
    dx, dgam, dbet, dres, dfilm = ops.groupnorm_bwd(xd, gamma.detach().to(dev()), beta.detach().to(dev()), G, cl(dout), mean, rstd,
                                                    act, residual=cl(res) if use_res else None,
                                                    film=film.detach().to(dev()) if use_film else None,
                                                    want_dres=use_res, want_dfilm=use_film)
    close(dx.permute(0, 2, 1), x.grad, what="gn dx")
    close(dgam, gamma.grad, what="gn dgamma")
    close(dbet, beta.grad, what="gn dbeta")
    if use_res:
        close(dres.permute(0, 2, 1), res.grad, what="gn dres")
    if use_film:
        close(dfilm, film.grad, what="gn dfilm")


@pytest.mark.parametrize("L,heads,ch,n", [(256, 4, 32, 3), (64, 5, 32, 7), (16, 2, 16, 2), (1024, 2, 32, 1)])
def test_attention(L, heads, ch, n):
    from v2a_hip import ops
    g = torch.Generator().manual_seed(L + heads)
    C = heads * ch
    qkv = torch.randn(n, 3 * C, L, generator=g)          # reference layout [N, 3C, L]
    q, k, v = qkv.reshape(n * heads, 3 * ch, L).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    ref = torch.einsum("bts,bcs->bct", w, v).reshape(n, C, L)
    out = ops.attention(qkv.permute(0, 2, 1).contiguous().to(dev()).view(n * L, 3 * C), n, L, heads, ch)
    close(out.view(n, L, C).permute(0, 2, 1), ref, what="attention")


@pytest.mark.parametrize("L,heads,n", [(256, 16, 14), (64, 20, 7), (32, 1, 1), (96, 3, 2)])
def test_attention_three_plane_mfma_vs_fp64(L, heads, n):
    """attn_mfma_x3_kernel (fp32 storage, head_ch 32, L <= 256): QK^T and PV as six bf16 plane products each on the matrix pipe, softmax
    state in fp32.  Against an fp64 evaluation of QKVAttentionLegacy (unet.py:341-358) -- the fp32 budget of the plane products (~1e-6 of
    max |out|) -- on inputs with a wide dynamic range (large logits: the softmax is close to one-hot for some queries); bitwise repeatable."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1:
        pytest.skip("three-plane mode only")
    ch = 32
    g = torch.Generator().manual_seed(L * heads + n)
    C = heads * ch
    qkv = torch.randn(n, 3 * C, L, generator=g) * torch.rand(n, 3 * C, 1, generator=g).mul(2).exp2()
    q, k, v = qkv.double().reshape(n * heads, 3 * ch, L).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    ref = torch.einsum("bts,bcs->bct", w, v).reshape(n, C, L)
    x = qkv.permute(0, 2, 1).contiguous().to(dev()).view(n * L, 3 * C)
    out = ops.attention(x, n, L, heads, ch)
    got = out.view(n, L, C).permute(0, 2, 1).cpu().double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    assert torch.equal(out, ops.attention(x, n, L, heads, ch))


def test_maxpool_and_spatial_softmax():
    from v2a_hip import ops
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 16, 20, 20, generator=g, requires_grad=True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    yd, idx = ops.maxpool_fwd(nhwc(x.detach()))
    close(nchw(yd), y, what="maxpool fwd")
    dx = ops.maxpool_bwd(nhwc(dy), idx, (3, 20, 20, 16))
    close(nchw(dx), x.grad, what="maxpool bwd")
    import numpy as np
    B, K, H, W = 4, 32, 4, 4
    f = torch.randn(B, K, H, W, generator=g, requires_grad=True)
    px, py = np.meshgrid(np.linspace(-1., 1., W), np.linspace(-1., 1., H))
    px = torch.from_numpy(px.reshape(1, H * W)).float()
    py = torch.from_numpy(py.reshape(1, H * W)).float()
    att = F.softmax(f.reshape(-1, H * W), dim=-1)
    kp = torch.cat([(px * att).sum(1, keepdim=True), (py * att).sum(1, keepdim=True)], 1).view(B, K * 2)
    dkp = torch.randn(kp.shape, generator=g)
    kp.backward(dkp)
    kpd, attd = ops.spatial_softmax_fwd(nhwc(f.detach()))
    close(kpd, kp, what="spatial softmax fwd")
    df = ops.spatial_softmax_bwd(attd, kpd, dkp.to(dev()))
    close(nchw(df), f.grad, what="spatial softmax bwd")


def test_elementwise_bits():
    from v2a_hip import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1000, generator=g) * 3
    for act, fn in [("silu", F.silu), ("mish", F.mish), ("relu", F.relu), ("gelu", F.gelu)]:
        close(ops.act_fwd(x.to(dev()), act), fn(x), what=act)
    xr = x.clone().requires_grad_(True)
    F.mish(xr).backward(torch.ones_like(xr))
    close(ops.act_bwd(x.to(dev()), torch.ones(1000, device=dev()), "mish"), xr.grad, what="mish bwd")
    t = torch.tensor([0, 7, 50, 99])
    half = 64
    e = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    ref = torch.cat(((t[:, None] * e[None]).sin(), (t[:, None] * e[None]).cos()), -1)
    close(ops.sincos_embed(t.to(dev()), 128, 0), ref, what="SinusoidalPosEmb")
    fr = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * fr[None]
    close(ops.sincos_embed(t.to(dev()), 128, 1), torch.cat([torch.cos(a), torch.sin(a)], -1), what="timestep_embedding")
    img = torch.rand(2, 3, 8, 8, generator=g)
    close(ops.nchw_to_nhwc(img.to(dev()), normalize=True).permute(0, 3, 1, 2), 2 * img - 1, what="nchw->nhwc")
    u8 = (img * 255).to(torch.uint8)
    close(ops.nchw_to_nhwc(u8.to(dev()), normalize=False).permute(0, 3, 1, 2), u8.float() / 255, what="u8 convert")
    ln_x = torch.randn(10, 512, generator=g)
    gg, bb = torch.randn(512, generator=g), torch.randn(512, generator=g)
    close(ops.layernorm(ln_x.to(dev()), gg.to(dev()), bb.to(dev())), F.layer_norm(ln_x, (512,), gg, bb), what="layernorm")


@pytest.mark.parametrize("case", [(2, 64, 16, 16, 128, 3, 2, 1), (2, 256, 8, 8, 512, 3, 1, 1), (3, 128, 9, 7, 96, 1, 1, 0), (64, 32, 1, 16, 64, (1, 5), 1, (0, 2))])
def test_dgrad_from_forward_pack(case):
    """bmode=1: the data gradient read straight from the forward pack (N-major loader) equals autograd's."""
    from v2a_hip import ops
    N, Cin, H, W, Cout, k, s, p = case
    kh, kw = (k, k) if isinstance(k, int) else k
    ph, pw = (p, p) if isinstance(p, int) else p
    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, kh, kw, generator=g) / math.sqrt(Cin * kh * kw)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=(ph, pw))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wp = ops.pack_weight(w.detach().to(dev()), 0)
    dx = ops.conv2d(nhwc(dy), wp, None, Cin, kh, kw, (1, 1), (kh - 1 - ph, kw - 1 - pw), idil=s, out_hw=(H, W), bmode=1)
    close(nchw(dx), x.grad, what="dgrad via forward pack")


def test_bf16_mfma_mode_matches_bf16_rounded_reference():
    """precision='bf16': operands rounded to bf16 (RNE), fp32 accumulation -> equals torch conv on bf16-rounded inputs to ~1e-5,
    and stays within ~1e-2 of the fp32 result."""
    import v2a_hip
    from v2a_hip import ops
    g = torch.Generator().manual_seed(77)
    N, Cin, H, W, Cout = 4, 64, 24, 24, 96
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    old = v2a_hip.set_precision("bf16")
    try:
        y = ops.conv2d(nhwc(x), ops.pack_weight(w.to(dev()), 0), b.to(dev()), Cout, 3, 3, (1, 1), (1, 1))
    finally:
        v2a_hip.set_precision(old)
    xr, wr = x.bfloat16().float(), w.bfloat16().float()
    close(nchw(y), F.conv2d(xr, wr, b, padding=1), tol=2e-5, what="bf16 mode vs bf16-rounded reference")
    close(nchw(y), F.conv2d(x, w, b, padding=1), tol=2e-2, what="bf16 mode vs fp32")
    assert v2a_hip.get_precision() == "fp32"


def test_bf16_wgrad_matches_bf16_rounded_reference():
    import v2a_hip
    from v2a_hip import ops
    g = torch.Generator().manual_seed(78)
    for (N, Cin, H, W, Cout, k, s_) in [(4, 64, 24, 24, 96, 3, 1), (64, 256, 1, 8, 128, (1, 5), 1), (3, 128, 16, 16, 128, 3, 2)]:
        kh, kw = (k, k) if isinstance(k, int) else k
        x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float().requires_grad_(False)
        w = (torch.randn(Cout, Cin, kh, kw, generator=g) / math.sqrt(Cin * kh * kw)).requires_grad_(True)
        b = torch.zeros(Cout, requires_grad=True)
        y = F.conv2d(x, w, b, stride=s_, padding=(kh // 2, kw // 2))
        dy = torch.randn(y.shape, generator=g).bfloat16().float()
        y.backward(dy)
        old = v2a_hip.set_precision("bf16")
        try:
            db = torch.empty(Cout, device=dev())
            dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), tuple(w.shape), kh, kw, (s_, s_), (kh // 2, kw // 2), dbias=db)
        finally:
            v2a_hip.set_precision(old)
        close(dw, w.grad, tol=2e-5, what="bf16 wgrad (inputs exactly representable in bf16)")
        close(db, b.grad, tol=2e-5, what="fused bias grad")


def test_groupnorm_large_path_odd_group_width():
    """cg % 4 != 0 on the HBM-bound path (UnetBridge: 160 channels / 32 groups = 5): float4 columns straddle groups."""
    from v2a_hip import ops
    g = torch.Generator().manual_seed(4)
    N, S, C, G = 2, 9000, 160, 32
    x = (torch.randn(N, C, S, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    a = F.silu(F.group_norm(x, G, gamma, beta, eps=1e-5))
    dout = torch.randn(a.shape, generator=g)
    a.backward(dout)
    cl = lambda t: t.detach().permute(0, 2, 1).contiguous().to(dev())
    y, mean, rstd = ops.groupnorm_fwd(cl(x), gamma.detach().to(dev()), beta.detach().to(dev()), G, "silu")
    close(y.permute(0, 2, 1), a, what="gn fwd cg=5")
    dx, dg, db, _, _ = ops.groupnorm_bwd(cl(x), gamma.detach().to(dev()), beta.detach().to(dev()), G, cl(dout), mean, rstd, "silu")
    close(dx.permute(0, 2, 1), x.grad, what="gn dx cg=5")
    close(dg, gamma.grad, what="gn dgamma cg=5")
    close(db, beta.grad, what="gn dbeta cg=5")


# ------------------------------------------------------------------------------------------------ bf16-storage family
def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    dict(N=3, H=20, W=24, C1=128, C2=0, Co=192, k=3, s=1, ups=False),          # ragged M (1440 rows), Cout not a tile multiple
    dict(N=2, H=16, W=16, C1=64, C2=128, Co=128, k=3, s=1, ups=False),         # two-source concat
    dict(N=2, H=17, W=19, C1=64, C2=0, Co=64, k=3, s=2, ups=False),            # stride 2, odd extents
    dict(N=2, H=8, W=8, C1=128, C2=0, Co=128, k=3, s=1, ups=True),             # folded nearest x2 upsample
    dict(N=1, H=1, W=300, C1=256, C2=0, Co=3, k=1, s=1, ups=False, f32=True),   # 1x1 / linear, tiny Cout, fp32 output
    dict(N=4, H=8, W=8, C1=1280, C2=0, Co=640, k=3, s=1, ups=False),           # deep K -> split-K path
    dict(N=2, H=7, W=256, C1=128, C2=0, Co=128, k=(3, 1), s=1, ups=False),     # temporal (3x1) view
])
def test_conv2d_h_matches_fp32_reference_on_bf16_inputs(case):
    from v2a_hip import ops
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(5)
    N, H, W, C1, C2, Co = case["N"], case["H"], case["W"], case["C1"], case["C2"], case["Co"]
    kh, kw = (case["k"], case["k"]) if isinstance(case["k"], int) else case["k"]
    s, ups = case["s"], case["ups"]
    x = _bf(torch.randn(N, H, W, C1, generator=g)).to(dev)
    x2 = _bf(torch.randn(N, H, W, C2, generator=g)).to(dev) if C2 else None
    w = (torch.randn(Co, C1 + C2, kh, kw, generator=g) * 0.05).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    wp = ops.pack_weight_h(w)
    # fp32 reference on exactly the bf16-rounded operands
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    wq = w.to(torch.bfloat16).float()
    ref = torch.nn.functional.conv2d(xin.double(), wq.double(), b.double(), stride=s, padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1], ref.shape[2]
    rowvec = torch.randn(N, Co, generator=g).to(dev)
    res = _bf(torch.randn(N, OH, OW, Co, generator=g)).to(dev)
    ref = ref + rowvec.double()[:, None, None, :] + res.double()
    y = ops.conv2d_h(x, wp, b, Co, kh, kw, (s, s), (kh // 2, kw // 2), x2=x2, rowvec=rowvec, rows_per_batch=OH * OW, residual=res,
                     ups=ups, out_f32=case.get("f32", False))
    assert y.shape == (N, OH, OW, Co)
    scale = ref.abs().max().item()
    if case.get("f32", False):
        assert y.dtype == torch.float32
        assert (y.double() - ref).abs().max().item() <= 2e-5 * scale            # fp32 accumulation order only
    else:
        assert y.dtype == torch.bfloat16
        # one bf16 rounding of the exact result: |err| <= 2^-9 |ref| (+ accumulation noise)
        err = (y.double() - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -8 + 2e-5 * scale).all()


@pytest.mark.gpu
def test_cast_roundtrip_and_pack_weight_h():
    from v2a_hip import ops
    dev = "cuda:0"
    x = torch.randn(4099 * 4, device=dev)
    h = ops.cast_h(x)
    assert torch.equal(h, x.to(torch.bfloat16))                                   # round-to-nearest-even like torch
    assert torch.equal(ops.cast_f(h), h.float())
    w = torch.randn(5, 64, 3, 3, device=dev)
    p = ops.pack_weight_h(w).view(5, 3, 3, 64)
    assert torch.equal(p, w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize("N,S,C1,C2,act", [(3, 1000, 128, 0, "silu"), (2, 777, 256, 128, "silu"), (4, 64, 640, 640, "silu"),
                                            (5, 256, 512, 0, "none"), (2, 4099, 384, 0, "silu")])
def test_groupnorm_fwd_h_matches_fp32_groupnorm_on_bf16_inputs(N, S, C1, C2, act):
    from v2a_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(N * 1000 + S)
    x = (torch.randn(N, S, C1, generator=g) * 2 + 0.5).to(torch.bfloat16).to(dev)
    x2 = (torch.randn(N, S, C2, generator=g) - 0.3).to(torch.bfloat16).to(dev) if C2 else None
    C = C1 + C2
    gamma, beta = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    y = ops.groupnorm_fwd_h(x, gamma, beta, 32, act, x2=x2)
    full = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    ref = torch.nn.functional.group_norm(full.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), eps=1e-5).permute(0, 2, 1)
    if act == "silu":
        ref = ref * torch.sigmoid(ref)
    assert y.dtype == torch.bfloat16 and y.shape == (N, S, C)
    err = (y.double() - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -8 + 1e-4 * ref.abs().max()).all(), float(err.max())


@pytest.mark.gpu
def test_conv_epilogue_statistics_feed_groupnorm():
    """conv2d_h(want_stats=True) returns per-64-row sum / sum-of-squares slabs of its rounded outputs; GroupNorm consuming them
    (two-source concat, ragged last tile) equals GroupNorm that computes its own statistics."""
    from v2a_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(2)
    N, H, W = 3, 96, 98                       # 9408 rows per sample = 147 blocks of 64; M = 28224 = 220.5 tiles of 128 (ragged)
    xa = torch.randn(N, H, W, 64, generator=g).to(torch.bfloat16).to(dev)       # K = 576: 9 k tiles, below the split-K threshold
    xb = torch.randn(N, H, W, 64, generator=g).to(torch.bfloat16).to(dev)
    wa = ops.pack_weight_h((torch.randn(192, 64, 3, 3, generator=g) * 0.05).to(dev))
    wb = ops.pack_weight_h((torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev))        # 64-wide: the BN=64 tile variant
    ya, sa = ops.conv2d_h(xa, wa, None, 192, 3, 3, (1, 1), (1, 1), want_stats=True)
    yb, sb = ops.conv2d_h(xb, wb, None, 64, 3, 3, (1, 1), (1, 1), want_stats=True)
    assert sa is not None and sb is not None and sa.shape == (441, 2, 192) and sb.shape == (441, 2, 64)
    rows = ya.float().view(441, 64, 192)
    assert torch.allclose(sa[:, 0], rows.sum(1), rtol=1e-4, atol=1e-3) and torch.allclose(sa[:, 1], (rows * rows).sum(1), rtol=1e-4, atol=1e-3)
    C = 256
    gamma, beta = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    own = ops.groupnorm_fwd_h(ya.view(N, H * W, 192), gamma, beta, 32, "silu", x2=yb.view(N, H * W, 64))
    fused = ops.groupnorm_fwd_h(ya.view(N, H * W, 192), gamma, beta, 32, "silu", x2=yb.view(N, H * W, 64), stats=sa, stats2=sb)
    d = (own.float() - fused.float()).abs()
    assert (d <= own.float().abs() * 2.0 ** -7 + 1e-5).all(), float(d.max())            # at most one bf16 ulp apart
    assert (d > 0).float().mean().item() < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("N,L,heads", [(3, 256, 16), (5, 64, 20), (2, 1024, 4), (2, 96, 3)])
def test_attention_h_mfma_matches_fp32_attention_on_bf16_inputs(N, L, heads):
    """QKVAttentionLegacy (unet.py:341-358) on bf16 tensors: MFMA kernel (swapped QK^T, in-register P) vs an fp64 softmax-attention of
    the same bf16-rounded q, k, v; P and the output are rounded to bf16, so the bound is a few bf16 ulps of the value range."""
    from v2a_hip import ops
    dev = "cuda:0"
    hc = 32
    g = torch.Generator().manual_seed(L + heads)
    qkv = (torch.randn(N * L, heads * 3 * hc, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    out = ops.attention(qkv, N, L, heads, hc)
    assert out.dtype == torch.bfloat16 and out.shape == (N * L, heads * hc)
    x = qkv.double().view(N, L, heads, 3, hc)
    q, k, v = x[..., 0, :], x[..., 1, :], x[..., 2, :]
    s = torch.einsum("nqhc,nkhc->nhqk", q, k) / math.sqrt(hc)
    ref = torch.einsum("nhqk,nkhc->nqhc", torch.softmax(s, -1), v).reshape(N * L, heads * hc)
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item(), err
    # and it agrees with the VALU bf16 kernel's contract on a transposition-sensitive input (distinct q / k / v statistics per channel)
    assert torch.isfinite(out).all()


@pytest.mark.gpu
def test_conv2d_h2_multistage_kernel_matches_fp64_reference():
    """The 8-wave, 4-stage 256x256 kernel (csrc/igemm_h2.hip; counted-vmcnt pipeline) on a problem large enough to be routed to it:
    two-source concat, bias + per-sample row vector + residual, ragged last tile, statistics; result within one bf16 ulp of an
    fp64 conv of the same rounded operands -- and equal to the 128x128 kernel's result up to summation order."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    dev = "cuda:0"
    g = torch.Generator().manual_seed(9)
    N, H, W, C1, C2, Co = 5, 168, 168, 64, 64, 256                  # M = 141120 = 551.25 tiles of 256 rows
    assert lib.v2a_conv2d_h2_eligible(N * H * W, Co, 9 * (C1 + C2), C1, C2) == 1
    x = torch.randn(N, H, W, C1, generator=g).to(torch.bfloat16).to(dev)
    x2 = torch.randn(N, H, W, C2, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    rowvec = torch.randn(N, Co, generator=g).to(dev)
    res = torch.randn(N, H, W, Co, generator=g).to(torch.bfloat16).to(dev)
    wp = ops.pack_weight_h(w)
    y, st = ops.conv2d_h(x, wp, b, Co, 3, 3, (1, 1), (1, 1), x2=x2, rowvec=rowvec, rows_per_batch=H * W, residual=res, want_stats=True)
    assert ops.last_kernel[0].startswith("conv_igemm_h2") and st is not None
    ops.CONV_H2[0] = False           # test hook: the same layer on the 128 x 128 kernel
    try:
        y1 = ops.conv2d_h(x, wp, b, Co, 3, 3, (1, 1), (1, 1), x2=x2, rowvec=rowvec, rows_per_batch=H * W, residual=res)
    finally:
        ops.CONV_H2[0] = True
    d = (y.float() - y1.float()).abs()
    assert (d <= y1.float().abs() * 2.0 ** -7 + 1e-5).all() and (d > 0).float().mean().item() < 0.01
    # fp64 reference on a slab of rows (full conv on the host would take minutes): samples 0 and 4, all channels
    xin = torch.cat([x.float(), x2.float()], -1).permute(0, 3, 1, 2).cpu().double()
    wq = w.to(torch.bfloat16).float().cpu().double()
    for n in (0, 4):
        ref = torch.nn.functional.conv2d(xin[n:n + 1], wq, b.cpu().double(), padding=1).permute(0, 2, 3, 1)[0]
        ref = ref + rowvec[n].cpu().double() + res[n].cpu().double()
        err = (y[n].cpu().double() - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -8 + 2e-5 * ref.abs().max()).all(), float(err.max())
    rows = y.float().view(-1, 64, Co)
    assert torch.allclose(st[:, 0], rows.sum(1), rtol=1e-4, atol=2e-3) and torch.allclose(st[:, 1], (rows * rows).sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("epi", ["residual", "rowvec"])
@pytest.mark.parametrize("N,H,W,C,Co", [(34, 64, 64, 96, 256), (36, 48, 80, 64, 128), (44, 32, 32, 128, 384), (56, 64, 32, 64, 128)])
def test_conv2d_halo_h3_kernel_matches_fp64_reference(N, H, W, C, Co, epi):
    """The halo-tile 3x3 kernel (csrc/igemm_h3.hip: 16 x 16 output patches, the 18 x 18 input halo DMA-ed once per 32-channel chunk and
    read by the nine taps through shifted LDS windows, counted-vmcnt weight ring): both instances (256x256 / 256x128), image borders on
    every side of the patch grid, non-square frames, three chunk counts, bias + per-sample row vector + residual + statistics.  Within
    one bf16 ulp of an fp64 conv of the same rounded operands, and equal to the tap-by-tap kernel up to summation order.  Both epilogues:
    fp32 staging (with a residual) and the register path (pack_subtile3, without one)."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    dev = "cuda:0"
    g = torch.Generator().manual_seed(31 + N)
    assert lib.v2a_conv2d_h3_eligible(N, H, W, C, Co, 3, 3, 1, 1, 1, 1, 0, 0) == 1
    x = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Co, C, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    rowvec = torch.randn(N, Co, generator=g).to(dev)
    res = torch.randn(N, H, W, Co, generator=g).to(torch.bfloat16).to(dev) if epi == "residual" else None
    wp = ops.pack_weight_h(w)
    y, st = ops.conv2d_h(x, wp, b, Co, 3, 3, (1, 1), (1, 1), rowvec=rowvec, rows_per_batch=H * W, residual=res, want_stats=True)
    assert ops.last_kernel[0].startswith("conv_halo_h3") and st is not None
    ops.CONV_H3[0] = False          # test hook: the same layer on the tap-by-tap kernel
    try:
        y1 = ops.conv2d_h(x, wp, b, Co, 3, 3, (1, 1), (1, 1), rowvec=rowvec, rows_per_batch=H * W, residual=res)
    finally:
        ops.CONV_H3[0] = True
    assert not ops.last_kernel[0].startswith("conv_halo_h3")
    d = (y.float() - y1.float()).abs()
    assert (d <= y1.float().abs() * 2.0 ** -7 + 1e-5).all() and (d > 0).float().mean().item() < 0.01
    xin = x.float().permute(0, 3, 1, 2).cpu().double()
    wq = w.to(torch.bfloat16).float().cpu().double()
    for n in (0, N - 1):
        ref = torch.nn.functional.conv2d(xin[n:n + 1], wq, b.cpu().double(), padding=1).permute(0, 2, 3, 1)[0]
        ref = ref + rowvec[n].cpu().double() + (res[n].cpu().double() if res is not None else 0.0)
        err = (y[n].cpu().double() - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -8 + 2e-5 * ref.abs().max()).all(), float(err.max())
    # statistics: blocks are numbered per (patch, 64-row group), so compare per FRAME (what GroupNorm reduces over)
    yf = y.float().view(N, H * W, Co)
    nb = H * W // 64
    assert torch.allclose(st[:, 0].view(N, nb, Co).sum(1), yf.sum(1), rtol=1e-4, atol=5e-2)
    assert torch.allclose(st[:, 1].view(N, nb, Co).sum(1), (yf * yf).sum(1), rtol=1e-4, atol=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,C1,C2,Co,act", [(112, 32, 32, 128, 0, 128, "silu"), (56, 32, 32, 128, 64, 256, "silu"), (35, 16, 48, 64, 32, 384, "none"),
                                                 (28, 64, 64, 32, 0, 128, "silu")])
def test_conv2d_halo_h3_applies_groupnorm_in_the_loader(N, H, W, C1, C2, Co, act):
    """GroupNorm + SiLU folded into the 3x3 halo conv (v2a_groupnorm_prep_h -> v2a_conv2d_fwd_h3_gn): the conv normalises its input
    halo in LDS -- y = act(x * a[n, c] + b[n, c]), rounded to bf16, padding ring left at zero -- instead of reading a tensor the
    apply pass wrote.  Same arithmetic as gn_apply_h, so the result must equal apply-then-conv on the same kernel bit for bit; also
    for the decoder's two-source input [x | x2], per-sample statistics over 7 frames, 1 .. 6 channel chunks, and bitwise repeatable."""
    from v2a_hip import ops
    dev, F = "cuda:0", 7
    g = torch.Generator().manual_seed(N + C1)
    C = C1 + C2
    x = (torch.randn(N, H, W, C1, generator=g) * 1.5 + 0.3).to(torch.bfloat16).to(dev)
    x2 = (torch.randn(N, H, W, C2, generator=g) * 0.7 - 0.2).to(torch.bfloat16).to(dev) if C2 else None
    gamma, beta = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    w = ops.pack_weight_h((torch.randn(Co, C, 3, 3, generator=g) * 0.05).to(dev))
    b = torch.randn(Co, generator=g).to(dev)
    rowvec = torch.randn(N // F, Co, generator=g).to(dev)
    assert ops.gn_fusable(N, H, W, C, Co, 3, 3, (1, 1), (1, 1), False)
    S = F * H * W
    pg = ops.groupnorm_prep_h(x.view(N // F, S, C1), gamma, beta, 32 if C % 32 == 0 else 8, act, x2=None if x2 is None else x2.view(N // F, S, C2))
    y, st = ops.conv2d_h(x, w, b, Co, 3, 3, (1, 1), (1, 1), x2=x2, rowvec=rowvec, rows_per_batch=S, want_stats=True, pre_gn=pg)
    assert ops.last_kernel[0].startswith("conv_halo_h3_gn") and st is not None
    a = pg.apply().view(N, H, W, C)                                   # the stand-alone apply pass on the same table
    ref_fwd = ops.groupnorm_fwd_h(x.view(N // F, S, C1), gamma, beta, 32 if C % 32 == 0 else 8, act, x2=None if x2 is None else x2.view(N // F, S, C2))
    assert torch.equal(a.view(N // F, S, C), ref_fwd)                 # prep + apply == the one-call GroupNorm
    y1, st1 = ops.conv2d_h(a, w, b, Co, 3, 3, (1, 1), (1, 1), rowvec=rowvec, rows_per_batch=S, want_stats=True)
    assert ops.last_kernel[0].startswith("conv_halo_h3<")
    assert torch.equal(y, y1) and torch.equal(st, st1)
    y2, _ = ops.conv2d_h(x, w, b, Co, 3, 3, (1, 1), (1, 1), x2=x2, rowvec=rowvec, rows_per_batch=S, want_stats=True, pre_gn=pg)
    assert torch.equal(y, y2)


@pytest.mark.gpu
def test_conv2d_halo_h3_folds_the_nearest_upsample():
    """Upsample (nearest x2 on H, W) + 3x3 conv (unet.py:105-115) on the halo kernel: the gather reads source pixel (ih >> 1, iw >> 1)."""
    from v2a_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(77)
    N, H, W, C, Co = 30, 16, 16, 64, 256                              # runs over 32 x 32 frames: 30 x 4 = 120 patches x 1 column tile ... x
    N = 60                                                            # 240 tiles
    x = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Co, C, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    wp = ops.pack_weight_h(w)
    y = ops.conv2d_h(x, wp, b, Co, 3, 3, (1, 1), (1, 1), ups=True)
    assert ops.last_kernel[0].startswith("conv_halo_h3") and y.shape == (N, 2 * H, 2 * W, Co)
    xin = x.float().permute(0, 3, 1, 2).cpu().double()
    xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    wq = w.to(torch.bfloat16).float().cpu().double()
    for n in (0, N - 1):
        ref = torch.nn.functional.conv2d(xin[n:n + 1], wq, b.cpu().double(), padding=1).permute(0, 2, 3, 1)[0]
        err = (y[n].cpu().double() - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -8 + 2e-5 * ref.abs().max()).all(), float(err.max())


@pytest.mark.gpu
@pytest.mark.parametrize("two", [False, True])
def test_fp32_groupnorm_takes_statistics_from_the_conv_epilogue(two):
    """fp32 sampler: the LDS-DMA fp32 conv leaves per-64-row (sum, sum of squares) blocks in its epilogue and
    v2a_groupnorm_fwd_st reduces those (in double, fixed order) instead of re-reading the tensor with gn_colreduce.  Same mean / rstd /
    output as the statistics pass to fp32 round-off, also for the virtual channel concat [x | x2] of the decoder blocks; bitwise
    repeatable."""
    from v2a_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(123)
    N, H, W, C, Co = 112, 32, 32, 64, 128                               # M = 114688: a single-pass plan on 128-row tiles
    x = torch.randn(N, H, W, C, generator=g).to(dev)
    w = ops.pack_weight((torch.randn(Co, C, 3, 3, generator=g) * 0.05).to(dev))
    b = torch.randn(Co, generator=g).to(dev)
    y, st = ops.conv2d(x, w, b, Co, 3, 3, (1, 1), (1, 1), want_stats=True)
    assert st is not None and ops.last_kernel[0].startswith(("conv_igemm_f32x3<128", "conv_igemm_f32p<128")) and st.shape == (N * H * W // 64, 2, Co)
    rows = y.view(-1, 64, Co)
    assert torch.allclose(st[:, 0], rows.sum(1), rtol=1e-5, atol=1e-4) and torch.allclose(st[:, 1], (rows * rows).sum(1), rtol=1e-5, atol=1e-4)
    y2 = st2 = None
    if two:
        w2 = ops.pack_weight((torch.randn(64, C, 3, 3, generator=g) * 0.05).to(dev))
        y2, st2 = ops.conv2d(x, w2, None, 64, 3, 3, (1, 1), (1, 1), want_stats=True)
        assert st2 is not None
    Ct = Co + (64 if two else 0)
    gamma, beta = torch.randn(Ct, generator=g).to(dev), torch.randn(Ct, generator=g).to(dev)
    S = 7 * H * W                                                       # seven frames per sample: 16 samples
    a = y.view(N // 7, S, Co)
    a2 = None if y2 is None else y2.view(N // 7, S, 64)
    ref, m0, r0 = ops.groupnorm_fwd(a, gamma, beta, 32, "silu", x2=a2)
    assert ops.lib.v2a_groupnorm_takes_slabs(S, Ct, 32) == 0            # the large (three-launch) path
    out, m1, r1 = ops.groupnorm_fwd(a, gamma, beta, 32, "silu", x2=a2, stats=st, stats2=st2)
    assert torch.allclose(m1, m0, rtol=0, atol=2e-6) and torch.allclose(r1, r0, rtol=2e-6, atol=0)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    out_b, m2, r2 = ops.groupnorm_fwd(a, gamma, beta, 32, "silu", x2=a2, stats=st, stats2=st2)
    assert torch.equal(out, out_b) and torch.equal(m1, m2) and torch.equal(r1, r2)


@pytest.mark.gpu
@pytest.mark.parametrize("epi", ["residual", "rowvec", "bias"])
@pytest.mark.parametrize("B,HW,C,Co", [(16, 1024, 128, 128), (4, 4096, 192, 256), (30, 512, 64, 128), (18, 256, 320, 384)])
def test_conv2d_temporal_frames_kernel_matches_fp64_reference(B, HW, C, Co, epi):
    """The frame-stack temporal kernel (csrc/igemm_h3.hip conv_frames_h3: all 7 frames of 64 pixels per workgroup, operand fragments
    kept in registers across the three taps, products against frames -1 / 7 not issued): first / last frame borders, several
    pixel tiles per sample, 2 .. 5 channel chunks (fewer than, equal to and more than the three LDS stages), bias + per-sample
    row vector + bf16 residual + statistics in natural 64-row blocks.  Within one bf16 ulp of an fp64 conv of the same rounded
    operands, and equal to the tap-by-tap kernel up to summation order.  `epi` selects the epilogue: with a residual the tile is staged
    in fp32; without one ("rowvec": bias + per-sample row vector, "bias": bias only) it is rounded and paired in registers
    (pack_subtile3) and the statistics never touch LDS."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    dev, F = "cuda:0", 7
    g = torch.Generator().manual_seed(5 + B)
    assert lib.v2a_conv2d_t3_eligible(B, F, HW, C, Co, 3, 1, 1, 1, 1, 0, 0, 0) == 1
    x = torch.randn(B, F, HW, C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Co, C, 3, 1, generator=g) * 0.08).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    rowvec = torch.randn(B, Co, generator=g).to(dev) if epi != "bias" else None
    res = torch.randn(B, F, HW, Co, generator=g).to(torch.bfloat16).to(dev) if epi == "residual" else None
    wp = ops.pack_weight_h(w)
    y, st = ops.conv2d_h(x, wp, b, Co, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=F * HW, residual=res, want_stats=True)
    assert ops.last_kernel[0].startswith("conv_frames_h3") and st is not None
    ops.CONV_H3[0] = False          # test hook: the same layer on the tap-by-tap kernel
    try:
        y1 = ops.conv2d_h(x, wp, b, Co, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=F * HW, residual=res)
    finally:
        ops.CONV_H3[0] = True
    assert not ops.last_kernel[0].startswith("conv_frames_h3")
    d = (y.float() - y1.float()).abs()
    assert (d <= y1.float().abs() * 2.0 ** -7 + 1e-5).all() and (d > 0).float().mean().item() < 0.01
    xin = x.float().permute(0, 3, 1, 2).cpu().double()
    wq = w.to(torch.bfloat16).float().cpu().double()
    for n in (0, B - 1):
        ref = torch.nn.functional.conv2d(xin[n:n + 1], wq, b.cpu().double(), padding=(1, 0)).permute(0, 2, 3, 1)[0]
        if rowvec is not None:
            ref = ref + rowvec[n].cpu().double()
        if res is not None:
            ref = ref + res[n].cpu().double()
        err = (y[n].cpu().double() - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -8 + 2e-5 * ref.abs().max()).all(), float(err.max())
    rows = y.float().view(-1, 64, Co)                                  # natural block numbering: rows / 64
    assert torch.allclose(st[:, 0], rows.sum(1), rtol=1e-4, atol=2e-3) and torch.allclose(st[:, 1], (rows * rows).sum(1), rtol=1e-4, atol=2e-3)
    y2, st2 = ops.conv2d_h(x, wp, b, Co, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=F * HW, residual=res, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(st, st2)                 # fixed summation order


@pytest.mark.parametrize("N,Cin,H,W,Cout", [(28, 128, 32, 32, 128), (56, 128, 16, 16, 256), (256, 128, 8, 8, 256), (14, 128, 32, 64, 128),
                                            (27, 128, 33, 32, 128), (4, 64, 32, 32, 64)])
def test_wgrad_halo_kernel_vs_torch(N, Cin, H, W, Cout):
    """3x3 / stride 1 / pad 1 weight gradient on the halo-tile kernel (conv_wgrad_halo_f32: 32-pixel patches, input halo shared by the
    nine taps in LDS) -- including image borders, several patches per row, a split reduction with a ragged last slice and the fused
    bias gradient; (27,128,33,32,128) has an odd height (still eligible: TW = 32 needs nothing of OH); the last case is below the
    size threshold (8 GFLOP, 4 channel blocks) and runs on the generic kernel."""
    from v2a_hip import ops
    g = torch.Generator().manual_seed(1000 + N + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv2d(x, w, b, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    db = torch.empty(Cout, device=dev())
    dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), tuple(w.shape), 3, 3, (1, 1), (1, 1), dbias=db)
    close(dw, w.grad, tol=2e-5, what="halo wgrad")
    close(db, b.grad, tol=2e-5, what="halo wgrad fused bias grad")
    acc = torch.ones_like(dw)
    ops.conv2d_wgrad(nhwc(x), nhwc(dy), tuple(w.shape), 3, 3, (1, 1), (1, 1), dw=acc, accumulate=True)
    close(acc - 1.0, w.grad, tol=2e-5, what="halo wgrad accumulate")


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s_", [(4, 64, 24, 24, 96, 3, 1), (64, 256, 1, 8, 128, (1, 5), 1), (3, 128, 16, 16, 128, 3, 2),
                                                  (8, 128, 32, 32, 256, 3, 1), (2, 72, 9, 11, 136, 3, 1), (6, 64, 32, 32, 64, 3, 1),
                                                  (3, 128, 16, 16, 64, 3, 2)])
def test_twin_fed_bf16_wgrad_lds_dma_tr_read(N, Cin, H, W, Cout, k, s_):
    """conv_wgrad_tr_h (bf16 twins of x / dy staged by LDS-DMA, MFMA operands through ds_read_b64_tr_b16): inputs exactly representable
    in bf16 make the bf16 products exact, so the result must match the fp32 reference to accumulation-order noise -- a transposed,
    swizzle-mismatched or mis-ordered operand fragment cannot pass (random data, ragged Cout / K tiles, image borders, stride 2,
    split reduction with a ragged tail, fused bias gradient, accumulate)."""
    import v2a_hip
    from v2a_hip import ops
    g = torch.Generator().manual_seed(500 + N + Cout)
    kh, kw = (k, k) if isinstance(k, int) else k
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, kh, kw, generator=g) / math.sqrt(Cin * kh * kw)).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv2d(x, w, b, stride=s_, padding=(kh // 2, kw // 2))
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    old = v2a_hip.set_precision("bf16")
    try:
        xd, dyd = nhwc(x), nhwc(dy)
        xh, dyh = ops.cast_h(xd), ops.cast_h(dyd)
        db = torch.empty(Cout, device=dev())
        dw = ops.conv2d_wgrad(xd, dyd, tuple(w.shape), kh, kw, (s_, s_), (kh // 2, kw // 2), dbias=db, x_h=xh, dy_h=dyh)
        assert ops.last_kernel[0].startswith("conv_wgrad_bf16h")
        acc = torch.full_like(dw, 0.5)
        ops.conv2d_wgrad(xd, dyd, tuple(w.shape), kh, kw, (s_, s_), (kh // 2, kw // 2), dw=acc, accumulate=True, x_h=xh, dy_h=dyh)
    finally:
        v2a_hip.set_precision(old)
    close(dw, w.grad, tol=2e-5, what="twin-fed bf16 wgrad")
    close(db, b.grad, tol=2e-5, what="twin-fed bf16 wgrad: fused bias grad")
    close(acc - 0.5, w.grad, tol=2e-5, what="twin-fed bf16 wgrad: accumulate")


@pytest.mark.parametrize("N,S,C,G", [(3, 40, 128, 8), (2, 9000, 128, 32)])
def test_groupnorm_twin_outputs_are_the_rounded_outputs(N, S, C, G):
    """groupnorm_fwd / groupnorm_bwd with twin_out: the bf16 twin is bit-for-bit the round-to-nearest-even cast of the fp32 output
    (small LDS-resident path and large HBM path), so a conv fed with it computes exactly what it would after a cast launch."""
    from v2a_hip import ops
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(N, S, C, generator=g) * 1.5 + 0.3).to(dev())
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev()), (0.1 * torch.randn(C, generator=g)).to(dev())
    dout = torch.randn(N, S, C, generator=g).to(dev())
    tw = []
    y, mean, rstd = ops.groupnorm_fwd(x, gamma, beta, G, "silu", twin_out=tw)
    y0, _, _ = ops.groupnorm_fwd(x, gamma, beta, G, "silu")
    assert torch.allclose(y, y0, rtol=0, atol=1e-5) and torch.equal(tw[0].view(torch.int16), ops.cast_h(y).view(torch.int16))
    tb = []
    dx, dg, db, _, _ = ops.groupnorm_bwd(x, gamma, beta, G, dout, mean, rstd, "silu", twin_out=tb)
    assert torch.equal(tb[0].view(torch.int16), ops.cast_h(dx).view(torch.int16))


def test_twin_fed_bf16_wgrad_two_source_concat():
    """conv_wgrad_tr_h over a channel concat [x | x2] read from both sources in place (ConditionalUnet1D decoder blocks: (1 x 5) conv over
    [h | skip]): exact-product inputs, fused bias gradient."""
    import v2a_hip
    from v2a_hip import ops
    g = torch.Generator().manual_seed(77)
    N, C1, C2, T, Cout = 64, 128, 64, 16, 256
    xa = torch.randn(N, C1, 1, T, generator=g).bfloat16().float()
    xb = torch.randn(N, C2, 1, T, generator=g).bfloat16().float()
    w = (torch.randn(Cout, C1 + C2, 1, 5, generator=g) / math.sqrt((C1 + C2) * 5)).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv2d(torch.cat([xa, xb], 1), w, b, padding=(0, 2))
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    old = v2a_hip.set_precision("bf16")
    try:
        xad, xbd, dyd = nhwc(xa), nhwc(xb), nhwc(dy)
        db = torch.empty(Cout, device=dev())
        dw = ops.conv2d_wgrad(xad, dyd, tuple(w.shape), 1, 5, (1, 1), (0, 2), x2=xbd, dbias=db, x_h=ops.cast_h(xad), x2_h=ops.cast_h(xbd),
                              dy_h=ops.cast_h(dyd))
        assert ops.last_kernel[0].startswith("conv_wgrad_bf16h")
    finally:
        v2a_hip.set_precision(old)
    close(dw, w.grad, tol=2e-5, what="two-source twin-fed wgrad")
    close(db, b.grad, tol=2e-5, what="two-source twin-fed wgrad: bias")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_grouped_weight_gradients_one_launch_vs_torch(mode):
    """conv_wgrad_multi_kernel: several layers' weight gradients (ResNet 3x3 at two widths incl. stride 2, a 1x1 downsample, a
    (1 x 5) ConditionalUnet1D conv with a channel concat, fused bias gradients, one accumulate) described into ONE launch, the split
    slabs finished by the collector's multi-tensor reduce.  fp32: the exact-f32 64x64 LDS-DMA body; bf16: the twin-fed bodies (inputs
    exactly representable in bf16, so the products are exact and only the summation order differs from torch).  An ineligible
    gradient (3 input channels) must be refused by `add`.  Run twice: bitwise reproducible."""
    import v2a_hip
    from v2a_hip import ops
    g = torch.Generator().manual_seed(77)
    rnd = (lambda *s: torch.randn(*s, generator=g).bfloat16().float()) if mode == "bf16" else (lambda *s: torch.randn(*s, generator=g))
    # N, Cin, H, W, Cout, (kh, kw), stride, Cin2, bias
    cases = [(6, 64, 32, 32, 64, (3, 3), 1, 0, False), (5, 64, 16, 16, 128, (3, 3), 2, 0, False), (5, 64, 16, 16, 128, (1, 1), 2, 0, False),
             (16, 256, 1, 8, 512, (1, 5), 1, 256, True), (3, 128, 16, 16, 128, (3, 3), 1, 0, True), (9, 512, 4, 4, 512, (3, 3), 1, 0, False),
             (4, 256, 8, 8, 256, (3, 3), 1, 0, True)]     # fp32: cases 0 / 4 / 6 take the halo-tile body (patch width 32 / 16 / 8)
    old = v2a_hip.set_precision(mode)
    try:
        col = ops.WgradCollector(dev())
        prob = []
        for (N, Cin, H, W, Cout, (kh, kw), s_, C2, bias) in cases:
            x = rnd(N, Cin + C2, H, W)
            w = (torch.randn(Cout, Cin + C2, kh, kw, generator=g) / math.sqrt((Cin + C2) * kh * kw)).requires_grad_(True)
            b = torch.zeros(Cout, requires_grad=True)
            y = F.conv2d(x, w, b, stride=s_, padding=(kh // 2, kw // 2))
            dy = rnd(*y.shape)
            y.backward(dy)
            xd = nhwc(x[:, :Cin])
            x2d = nhwc(x[:, Cin:]) if C2 else None
            dyd = nhwc(dy)
            tw = mode == "bf16"
            prob.append(dict(args=(xd, dyd, tuple(w.shape), kh, kw, (s_, s_), (kh // 2, kw // 2)),
                             kw=dict(x2=x2d, x_h=ops.cast_h(xd) if tw else None, dy_h=ops.cast_h(dyd) if tw else None,
                                     x2_h=ops.cast_h(x2d) if (tw and C2) else None),
                             ref_w=w.grad, ref_b=b.grad if bias else None, Cout=Cout, shape=tuple(w.shape)))
        outs = []
        for rep in range(2):
            batch = ops.WgradBatch(col)
            res = []
            for i, pr in enumerate(prob):
                dw = torch.full(pr["shape"], 0.25 if i == 1 else float("nan"), device=dev())
                db = torch.full((pr["Cout"],), float("nan"), device=dev()) if pr["ref_b"] is not None else None
                assert batch.add(*pr["args"], dw=dw, dbias=db, accumulate=(i == 1), slab_key=("t", i), **pr["kw"])
                res.append((dw, db))
            x3 = torch.randn(2, 16, 16, 3, device=dev())
            assert not batch.add(x3, torch.randn(2, 16, 16, 64, device=dev()), (64, 3, 3, 3), 3, 3, (1, 1), (1, 1),
                                 dw=torch.empty(64, 3, 3, 3, device=dev()))
            batch.launch()
            assert ops.last_kernel[0] == "conv_wgrad_multi"
            col.flush()
            torch.cuda.synchronize()
            outs.append(res)
        for i, (pr, (dw, db)) in enumerate(zip(prob, outs[0])):
            close(dw - (0.25 if i == 1 else 0.0), pr["ref_w"], tol=2e-5, what=f"grouped wgrad {i}")
            if db is not None:
                close(db, pr["ref_b"], tol=2e-5, what=f"grouped wgrad {i}: bias grad")
        for (a, ab), (b2, bb) in zip(outs[0], outs[1]):
            assert torch.equal(a, b2) and (ab is None or torch.equal(ab, bb))
    finally:
        v2a_hip.set_precision(old)


# ------------------------------------------------------------------------------------------------ fp16 instances of the 16-bit family
def _conv_ref64(x16, w, b, k, pad, n, x2=None, rowvec=None, res=None, hdt=torch.float16):
    """fp64 conv of sample n on exactly the 16-bit-rounded operands."""
    xin = x16.float() if x2 is None else torch.cat([x16.float(), x2.float()], -1)
    xin = xin.permute(0, 3, 1, 2).cpu().double()
    wq = w.to(hdt).float().cpu().double()
    ref = torch.nn.functional.conv2d(xin[n:n + 1], wq, b.cpu().double(), padding=pad).permute(0, 2, 3, 1)[0]
    if rowvec is not None:
        ref = ref + rowvec[n].cpu().double()
    if res is not None:
        ref = ref + res[n].cpu().double()
    return ref


@pytest.mark.gpu
def test_fp16_instances_of_the_16bit_kernels_match_fp64_on_fp16_rounded_operands():
    """BASELINE configs[4] says fp16 and the reference's GPU path is fp16 autocast (lb_online_trainer_v7.py:72-76,889): every kernel of
    the 16-bit video-storage family is instantiated for IEEE half as well (template flag F16: v_mfma_f32_32x32x16_f16, v_cvt_f16_f32),
    selected by the tensors' dtype.  Each one against an fp64 reference on exactly the fp16-rounded operands: one fp16 rounding of the
    exact result (2^-11 relative) plus accumulation noise -- 8x tighter than a bf16 result could be, so a kernel that silently ran its
    bf16 instance on fp16 bits (or mixed the two conversions) cannot pass.  Covers casts / packs, the tap-by-tap conv (two sources,
    split-K, fp32 output), the multi-stage 256-row conv, the halo-tile 3x3 conv plain and with GroupNorm + SiLU applied in its loader
    (== apply-then-conv, bitwise), the frame-stack temporal conv, GroupNorm with own / conv-epilogue statistics, MFMA attention."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    dev, H16 = "cuda:0", torch.float16
    g = torch.Generator().manual_seed(123)
    r16 = lambda *s: torch.randn(*s, generator=g).to(H16).to(dev)
    ulp = 2.0 ** -10                                                     # bound on one fp16 rounding (2^-11) with slack for subnormal steps
    # casts / packs
    x = torch.randn(4099 * 4, device=dev)
    h = ops.cast_h(x, H16)
    assert h.dtype == H16 and torch.equal(h, x.to(H16)) and torch.equal(ops.cast_f(h), h.float())
    assert torch.equal(ops.cast_h(x), x.to(torch.bfloat16))              # the bf16 instance is still what a bf16 request gets
    w5 = torch.randn(5, 64, 3, 3, device=dev)
    assert torch.equal(ops.pack_weight_h(w5, dtype=H16).view(5, 3, 3, 64), w5.permute(0, 2, 3, 1).contiguous().to(H16))
    xp = torch.randn(3, 5, 7, 6, device=dev)
    pp = ops.pad_cast_h(xp, 32, H16)
    assert pp.dtype == H16 and torch.equal(pp[..., :6], xp.to(H16)) and float(pp[..., 6:].abs().max()) == 0.0

    def check(y, ref, what):
        err = (y.cpu().double() - ref).abs()
        assert (err <= ref.abs() * ulp + 2e-5 * ref.abs().max()).all(), (what, float(err.max()))

    # tap-by-tap conv: two sources, bias / row vector / residual; split-K + fp32 output
    N, Hh, W, C1, C2, Co = 2, 16, 16, 64, 128, 128
    xa, xb = r16(N, Hh, W, C1), r16(N, Hh, W, C2)
    w = (torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.05).to(dev)
    b, rv, res = torch.randn(Co, generator=g).to(dev), torch.randn(N, Co, generator=g).to(dev), r16(N, Hh, W, Co)
    wp = ops.pack_weight_h(w, dtype=H16)
    y = ops.conv2d_h(xa, wp, b, Co, 3, 3, (1, 1), (1, 1), x2=xb, rowvec=rv, rows_per_batch=Hh * W, residual=res)
    assert y.dtype == H16 and ops.last_kernel[0].startswith("conv_igemm_h<")
    for n in range(N):
        check(y[n], _conv_ref64(xa, w, b, 3, 1, n, x2=xb, rowvec=rv, res=res), "conv_igemm_h fp16")
    xd = r16(4, 8, 8, 1280)
    wd = (torch.randn(640, 1280, 3, 3, generator=g) * 0.02).to(dev)
    bd = torch.randn(640, generator=g).to(dev)
    yd = ops.conv2d_h(xd, ops.pack_weight_h(wd, dtype=H16), bd, 640, 3, 3, (1, 1), (1, 1), out_f32=True)
    refd = _conv_ref64(xd, wd, bd, 3, 1, 1)
    assert yd.dtype == torch.float32 and float((yd[1].cpu().double() - refd).abs().max()) <= 3e-5 * float(refd.abs().max())
    # halo-tile 3x3 (+ statistics), frame-stack temporal, multi-stage 256-row kernels
    N, Hh, W, C, Co = 44, 32, 32, 128, 384
    assert lib.v2a_conv2d_h3_eligible(N, Hh, W, C, Co, 3, 3, 1, 1, 1, 1, 0, 0) == 1
    x = r16(N, Hh, W, C)
    w = (torch.randn(Co, C, 3, 3, generator=g) * 0.05).to(dev)
    b, rv, res = torch.randn(Co, generator=g).to(dev), torch.randn(N, Co, generator=g).to(dev), r16(N, Hh, W, Co)
    wp = ops.pack_weight_h(w, dtype=H16)
    y, st = ops.conv2d_h(x, wp, b, Co, 3, 3, (1, 1), (1, 1), rowvec=rv, rows_per_batch=Hh * W, residual=res, want_stats=True)
    assert ops.last_kernel[0].startswith("conv_halo_h3") and y.dtype == H16
    for n in (0, N - 1):
        check(y[n], _conv_ref64(x, w, b, 3, 1, n, rowvec=rv, res=res), "conv_halo_h3 fp16")
    yf = y.float().view(N, Hh * W, Co)                        # blocks are numbered per (patch, 64-row group): compare per frame
    nb = Hh * W // 64
    assert torch.allclose(st[:, 0].view(N, nb, Co).sum(1), yf.sum(1), rtol=1e-4, atol=5e-2)
    assert torch.allclose(st[:, 1].view(N, nb, Co).sum(1), (yf * yf).sum(1), rtol=1e-4, atol=5e-2)
    # GroupNorm + SiLU applied inside the halo conv's loader == apply-then-conv, bitwise; statistics from the conv epilogue == own pass to 1 ulp
    gamma, beta = torch.randn(Co, generator=g).to(dev), torch.randn(Co, generator=g).to(dev)
    y3 = y.view(N, Hh * W, Co)
    pg = ops.groupnorm_prep_h(y3, gamma, beta, 32, "silu", stats=st)
    w2 = (torch.randn(384, Co, 3, 3, generator=g) * 0.03).to(dev)
    wp2 = ops.pack_weight_h(w2, dtype=H16)
    assert ops.gn_fusable(N, Hh, W, Co, 384, 3, 3, (1, 1), (1, 1), False)
    fused = ops.conv2d_h(y, wp2, None, 384, 3, 3, (1, 1), (1, 1), pre_gn=pg)
    assert ops.last_kernel[0].startswith("conv_halo_h3_gn")
    applied = pg.apply().view(N, Hh, W, Co)
    unfused = ops.conv2d_h(applied, wp2, None, 384, 3, 3, (1, 1), (1, 1))
    assert fused.dtype == H16 and torch.equal(fused, unfused)
    gn_ref = torch.nn.functional.group_norm(y3.float().double().permute(0, 2, 1).cpu(), 32, gamma.double().cpu(), beta.double().cpu(), eps=1e-5).permute(0, 2, 1)
    gn_ref = gn_ref * torch.sigmoid(gn_ref)
    own = ops.groupnorm_fwd_h(y3, gamma, beta, 32, "silu")
    err = (own.cpu().double() - gn_ref).abs()
    assert own.dtype == H16 and (err <= gn_ref.abs() * ulp + 1e-4 * gn_ref.abs().max()).all(), float(err.max())
    d = (applied.view(N, Hh * W, Co).float() - own.float()).abs()
    assert (d <= own.float().abs() * 2.0 ** -9 + 1e-5).all()
    B, F, HW, C, Co = 16, 7, 1024, 128, 128
    assert lib.v2a_conv2d_t3_eligible(B, F, HW, C, Co, 3, 1, 1, 1, 1, 0, 0, 0) == 1
    x = r16(B, F, HW, C)
    w = (torch.randn(Co, C, 3, 1, generator=g) * 0.08).to(dev)
    b, rv, res = torch.randn(Co, generator=g).to(dev), torch.randn(B, Co, generator=g).to(dev), r16(B, F, HW, Co)
    y = ops.conv2d_h(x, ops.pack_weight_h(w, dtype=H16), b, Co, 3, 1, (1, 1), (1, 0), rowvec=rv, rows_per_batch=F * HW, residual=res)
    assert ops.last_kernel[0].startswith("conv_frames_h3") and y.dtype == H16
    for n in (0, B - 1):
        xin = x.float().permute(0, 3, 1, 2).cpu().double()
        ref = torch.nn.functional.conv2d(xin[n:n + 1], w.to(H16).float().cpu().double(), b.cpu().double(), padding=(1, 0)).permute(0, 2, 3, 1)[0]
        check(y[n], ref + rv[n].cpu().double() + res[n].cpu().double(), "conv_frames_h3 fp16")
    N, Hh, W, C1, C2, Co = 5, 168, 168, 64, 64, 256
    assert lib.v2a_conv2d_h2_eligible(N * Hh * W, Co, 9 * (C1 + C2), C1, C2) == 1
    xa, xb = r16(N, Hh, W, C1), r16(N, Hh, W, C2)
    w = (torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    y = ops.conv2d_h(xa, ops.pack_weight_h(w, dtype=H16), b, Co, 3, 3, (1, 1), (1, 1), x2=xb)
    assert ops.last_kernel[0].startswith("conv_igemm_h2") and y.dtype == H16
    check(y[4], _conv_ref64(xa, w, b, 3, 1, 4, x2=xb), "conv_igemm_h2 fp16")
    # MFMA attention (head_ch 32), QKVAttentionLegacy semantics (guided_diffusion/unet.py:341-358)
    n_fr, L, heads, ch = 3, 256, 16, 32
    qkv = (torch.randn(n_fr * L, heads * 3 * ch, generator=g)).to(H16).to(dev)
    out = ops.attention(qkv, n_fr, L, heads, ch)
    assert out.dtype == H16
    q, k, v = qkv.float().double().cpu().view(n_fr, L, heads, 3, ch).permute(3, 0, 2, 1, 4)      # [n, heads, L, ch]
    sc = ch ** -0.25
    wgt = torch.softmax((q * sc) @ (k * sc).transpose(-1, -2), dim=-1)
    ref = (wgt @ v).permute(0, 2, 1, 3).reshape(n_fr * L, heads * ch)
    err = (out.cpu().double() - ref).abs()
    assert float(err.max()) <= 2.0 ** -8 * float(ref.abs().max()), float(err.max())      # P is rounded to fp16 before P V (as the bf16 instance rounds to bf16)


@pytest.mark.gpu
def test_pack_weights_multi_forward_and_flipped_packs_vs_torch():
    """v2a_pack_weights_multi: every pack of a model in two launches.  Mode 0 ([Cout][Cin][taps] -> [Cout][taps][Cin]) and mode 1
    (tap-reversed transpose for the data gradient, through LDS tiles), fp32 + bf16 twins, for filters smaller and larger than a
    chunk, odd and even tap counts, operands that do not fill their last chunk or tile.  Bit-exact against torch permutes."""
    from v2a_hip._lib import lib, check
    from v2a_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(9)
    ce = lib.v2a_pack_chunk_elems()
    shapes = [(64, 3, 49), (128, 64, 9), (96, 40, 5), (8, 2560, 9), (33, 17, 4), (256, 512, 1), (5, 7, 3)]
    assert any(ci * t > ce for _, ci, t in shapes) and any(co * ci * t % ce for co, ci, t in shapes)
    ws, rows, ch0, ch1, outs = [], [], [], [], []
    for co, ci, taps in shapes:
        w = torch.randn(co, ci, taps, generator=g).to(dev)
        f32a, f32b = torch.zeros(w.numel(), device=dev), torch.zeros(w.numel(), device=dev)
        ha, hb = torch.zeros(w.numel(), dtype=torch.bfloat16, device=dev), torch.zeros(w.numel(), dtype=torch.bfloat16, device=dev)
        rows.append([w.data_ptr(), f32a.data_ptr(), co, ci, taps, 0, ha.data_ptr()])
        ch0 += [[len(rows) - 1, s0] for s0 in range(0, w.numel(), ce)]
        rows.append([w.data_ptr(), f32b.data_ptr(), co, ci, taps, 1, hb.data_ptr()])
        ch1 += [[len(rows) - 1, t] for t in range(-(-co // 64) * -(-(ci * taps) // 64))]
        ws.append(w)
        outs.append((f32a, ha, f32b, hb))
    tab = torch.tensor(rows, dtype=torch.int64).to(dev)
    c0, c1 = torch.tensor(ch0, dtype=torch.int32).to(dev), torch.tensor(ch1, dtype=torch.int32).to(dev)
    check(lib.v2a_pack_weights_multi(tab.data_ptr(), c0.data_ptr(), len(ch0), 0, ops._stream()), "pack0")
    check(lib.v2a_pack_weights_multi(tab.data_ptr(), c1.data_ptr(), len(ch1), 1, ops._stream()), "pack1")
    torch.cuda.synchronize()
    for w, (f32a, ha, f32b, hb) in zip(ws, outs):
        fwd = w.permute(0, 2, 1).contiguous().view(-1)                          # [Cout][taps][Cin]
        flip = w.flip(2).permute(1, 2, 0).contiguous().view(-1)                 # [Cin][taps reversed][Cout]
        assert torch.equal(f32a, fwd) and torch.equal(ha, fwd.to(torch.bfloat16))
        assert torch.equal(f32b, flip) and torch.equal(hb, flip.to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,u8", [(3, 32, 48, False), (64, 128, 128, True), (5, 18, 30, True)])
def test_stem_channel_window_conv_and_weight_gradient_vs_torch(N, H, W, u8):
    """RGB stem 7x7 / stride 2 / pad 3 (torchvision resnet18.conv1 behind vision_nets.py:29-39) as a channel-window conv over the
    zero-bordered 4-channel image (v2a_nchw_to_nhwc4p + pack mode 2 + v2a_conv2d_fwd_window_f32), and its weight gradient on the
    three-plane body fed from the same buffer: against torch fp32 on the CPU (fp64 yard-stick for the long reduction)."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("channel-window conv: fp32 three-plane mode only")
    torch.manual_seed(5)
    img = torch.randint(0, 256, (N, 3, H, W), dtype=torch.uint8) if u8 else torch.rand(N, 3, H, W)
    w = torch.randn(64, 3, 7, 7) * 0.1
    xn = (img.float() / 255.0 if u8 else img) * 2.0 - 1.0
    ref = F.conv2d(xn, w, stride=2, padding=3)
    xp = torch.zeros((N, H + 6, W + 6, 4), dtype=torch.float32, device=dev())
    ops.nchw_to_nhwc4p(img.to(dev()), xp, 3, normalize=True)
    close(xp[:, 3:-3, 3:-3, :3].permute(0, 3, 1, 2), xn, 1e-6, "padded NHWC4 interior")
    assert float(xp[..., 3].abs().max()) == 0.0 and float(xp[:, :3].abs().max()) == 0.0 and float(xp[:, :, -3:].abs().max()) == 0.0
    pw = torch.zeros(64 * 7 * 8 * 4, dtype=torch.float32, device=dev())
    ops.pack_weight(w.to(dev()), 2, pw)
    pw4 = pw.view(64, 7, 8, 4).cpu()
    assert torch.equal(pw4[:, :, :7, :3], w.permute(0, 2, 3, 1)) and float(pw4[:, :, 7].abs().max()) == 0.0 and float(pw4[..., 3].abs().max()) == 0.0
    y = ops.conv2d_window(xp, pw, 64, 7, 1, (2, 1), (H // 2, W // 2), xpitch=8, C=32)
    close(nchw(y), ref, 1e-4, "window conv")
    # weight gradient from the padded buffer (pad 0): the 4-channel filter's gradient, first three input channels kept
    dy = torch.randn(N, H // 2, W // 2, 64)
    dw4 = ops.conv2d_wgrad(xp, dy.to(dev()), (64, 4, 7, 7), 7, 7, (2, 2), (0, 0))
    wd = w.double().requires_grad_(True)
    F.conv2d(xn.double(), wd, stride=2, padding=3).backward(dy.permute(0, 3, 1, 2).double())
    close(dw4[:, :3], wd.grad.float(), 1e-4, "stem weight gradient")
    assert float(dw4[:, 3].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("N,HW,Ci,Co,res", [(8, 32, 64, 64, False), (3, 32, 96, 128, True), (8, 16, 128, 128, True), (6, 8, 256, 256, False),
                                             (16, 4, 512, 512, True), (40, 4, 64, 192, False), (2, 8, 32, 64, True), (3, 64, 64, 128, True)])
def test_three_plane_halo_conv_vs_fp64(N, HW, Ci, Co, res):
    """conv_halo_x3 (3x3 / stride 1 / pad 1 over square 32 / 16 / 8 / 4 maps, 128 output pixels x 64 channels per workgroup, halo split
    once per 32-channel chunk): tiles of four map rows, eight map rows, two whole maps and eight whole maps; odd chunk counts; the fp32
    residual of the data-gradient launches; split-K slabs.  Against fp64 torch, bound = the fp32 budget of the path (1e-4 of max |y|,
    measured ~3e-7)."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    assert (N * HW * HW) % 128 == 0
    g = torch.Generator().manual_seed(N * HW + Ci)
    x = torch.randn(N, Ci, HW, HW, generator=g) * torch.rand(N, Ci, HW, HW, generator=g).mul(4).exp2()
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    r = torch.randn(N, Co, HW, HW, generator=g) if res else None
    ref = F.conv2d(x.double(), w.double(), padding=1)
    if res:
        ref = ref + r.double()
    wp = ops.pack_weight(w.to(dev()), 0)
    y = ops.conv2d(nhwc(x), wp, None, Co, 3, 3, (1, 1), (1, 1), residual=nhwc(r) if res else None)
    assert ops.last_kernel[0] is not None
    close(nchw(y), ref.float(), 1e-4, "halo conv")
    err = (nchw(y).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 5e-6, err


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Ci,Co,kh,kw,st,pad,res", [(64, 1, 16, 7, 256, 1, 5, (1, 1), (0, 2), False), (64, 1, 16, 7, 256, 1, 1, (1, 1), (0, 0), True),
                                                          (3, 1, 16, 7, 300, 1, 5, (1, 1), (0, 2), True), (2, 9, 11, 3, 40, 3, 3, (2, 1), (1, 1), False),
                                                          (1, 5, 5, 1, 8, 5, 5, (1, 1), (2, 2), False), (5, 1, 8, 12, 16, 1, 5, (1, 2), (0, 2), True)])
def test_direct_small_reduction_conv_vs_fp64(N, H, W, Ci, Co, kh, kw, st, pad, res):
    """conv_smallk (csrc/igemm.hip): reductions of at most 64 values whose channel count the vector loaders cannot take -- the
    ConditionalUnet1D layers over the 7 action channels (Conv1d k = 5 -> K = 35, 1 x 1 residual conv and the final conv's data gradient ->
    K = 7) at batch 64, a ragged channel count above one 256-thread block, a strided 2-d case, a single input channel, K = 60.  Exact fp32 FMA
    chains: against fp64 torch at 1e-6 of max |y|, and against the tile kernel on the same inputs."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_precision() != 0:
        pytest.skip("fp32 mode only")
    g = torch.Generator().manual_seed(N * W + Ci)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, kh, kw, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=pad)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    wp = ops.pack_weight(w.to(dev()), 0)
    xd, rd, bd = nhwc(x), (nhwc(r) if res else None), b.to(dev())
    y = ops.conv2d(xd, wp, bd, Co, kh, kw, st, pad, residual=rd)
    assert ops.last_kernel[0].startswith("conv_smallk"), ops.last_kernel[0]
    err = (nchw(y).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-6, err
    old = lib.v2a_debug_set_smallk(0)
    try:
        y0 = ops.conv2d(xd, wp, bd, Co, kh, kw, st, pad, residual=rd)
        assert not ops.last_kernel[0].startswith("conv_smallk"), ops.last_kernel[0]
    finally:
        lib.v2a_debug_set_smallk(old)
    d = (y.double() - y0.double()).abs().max().item() / ref.abs().max().item()
    assert d < 1e-6, d
    from conftest import parity_record
    parity_record(f"conv_smallk {N}x{H}x{W} {Ci}->{Co} k{kh}x{kw} vs fp64", err, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N,HW,Ci,Co,res", [(64, 32, 64, 64, True), (64, 16, 128, 128, False), (64, 8, 256, 256, True), (64, 4, 512, 512, False),
                                             (48, 32, 64, 64, False), (64, 8, 96, 256, True), (128, 4, 160, 512, True), (192, 16, 32, 64, False)])
def test_three_plane_small_map_conv_vs_fp64(N, HW, Ci, Co, res):
    """conv_maps_x3 (csrc/igemm_x3m.hip): the 3x3 / stride 1 / pad 1 convs of the policy's ResNet-18 encoders at batch 64 (64 ch x 32^2,
    128 x 16^2, 256 x 8^2, 512 x 4^2: 1 / 2 / 4 / 8 slabs over 32-channel chunks), plus an uneven slab count (5 chunks in 3 slabs), one chunk
    per workgroup, a single-chunk layer and a tile count that is not a power of two.  Against fp64 torch at the fp32 budget of the path, and
    against conv_halo_x3 on the same inputs (same six plane products per block; the reduction is cut differently)."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    assert lib.v2a_conv2d_x3m_eligible(N, HW, Ci, Co) == 1
    g = torch.Generator().manual_seed(N * HW + Ci)
    x = torch.randn(N, Ci, HW, HW, generator=g) * torch.rand(N, Ci, HW, HW, generator=g).mul(4).exp2()
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    r = torch.randn(N, Co, HW, HW, generator=g) if res else None
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    wp = ops.pack_weight(w.to(dev()), 0)
    xd, rd, bd = nhwc(x), (nhwc(r) if res else None), b.to(dev())
    y = ops.conv2d(xd, wp, bd, Co, 3, 3, (1, 1), (1, 1), residual=rd)
    assert ops.last_kernel[0] == f"conv_maps_x3<{HW}>", ops.last_kernel[0]
    err = (nchw(y).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 5e-6, err
    old = lib.v2a_debug_set_maps_kernel(0)
    try:
        y0 = ops.conv2d(xd, wp, bd, Co, 3, 3, (1, 1), (1, 1), residual=rd)
        assert ops.last_kernel[0].startswith("conv_halo_x3"), ops.last_kernel[0]
    finally:
        lib.v2a_debug_set_maps_kernel(old)
    d = (y.double() - y0.double()).abs().max().item() / ref.abs().max().item()
    assert d < 2e-6, d
    from conftest import parity_record
    parity_record(f"conv_maps_x3 {N}x{HW}x{HW} {Ci}->{Co} vs fp64", err, 5e-6)
    parity_record("conv_maps_x3 vs conv_halo_x3", d, 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Ci,Co,ups,res", [(1, 128, 128, 32, 64, False, False), (2, 24, 48, 96, 128, False, True), (1, 8, 16, 64, 64, False, False),
                                                 (2, 16, 16, 64, 128, True, True), (1, 64, 64, 32, 64, True, False), (3, 12, 24, 160, 64, True, False),
                                                 (4, 4, 4, 128, 64, True, False)])
def test_three_plane_halo_conv_patches_and_upsample_vs_fp64(N, H, W, Ci, Co, ups, res):
    """conv_halo_x3<0>: 8 x 16 pixel patches of maps that are not one of the square 4 ... 64 sizes (the video UNet's 128 x 128 level,
    non-square maps, a map that is exactly one patch), and the nearest x2 upsample of unet.py:105-115 folded into the halo gather of every
    instance (H, W are the SOURCE sizes when ups; the 4 x 4 source lands on the whole-row 8 x 8 instance).  Against fp64 torch."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    g = torch.Generator().manual_seed(N * H + W + Ci)
    x = torch.randn(N, Ci, H, W, generator=g) * torch.rand(N, Ci, H, W, generator=g).mul(4).exp2()
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    b = torch.randn(Co, generator=g)
    OH, OW = (2 * H, 2 * W) if ups else (H, W)
    r = torch.randn(N, Co, OH, OW, generator=g) if res else None
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if ups else x.double()
    ref = F.conv2d(xin, w.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    wp = ops.pack_weight(w.to(dev()), 0)
    y = ops.conv2d(nhwc(x), wp, b.to(dev()), Co, 3, 3, (1, 1), (1, 1), residual=nhwc(r) if res else None, ups=ups)
    assert ops.last_kernel[0].startswith("conv_halo_x3"), ops.last_kernel[0]
    assert y.shape == (N, OH, OW, Co)
    err = (nchw(y).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 5e-6, err


@pytest.mark.gpu
@pytest.mark.parametrize("epi", ["residual", "rowvec", "bias"])
@pytest.mark.parametrize("B,HW,C,Co", [(16, 1024, 128, 128), (4, 4096, 96, 256), (30, 512, 32, 128), (18, 256, 160, 384)])
def test_three_plane_temporal_frames_kernel_vs_fp64(B, HW, C, Co, epi):
    """conv_frames_x3 (csrc/igemm_x3t.hip): the (3 x 1) temporal conv of the fp32 video UNet with all 7 frames of 64 pixels in one
    workgroup and the reduction in 16-channel phases -- first / last frame borders (products against frames -1 and 7 are not issued),
    several pixel tiles per sample and several tiles per persistent workgroup, 1 .. 5 channel chunks (2 .. 10 phases), bias + per-sample
    row vector + fp32 residual, statistics in natural 64-row blocks.  Against fp64 torch at the fp32 budget of the three-plane products
    (measured ~3e-7 of max |y|); bitwise repeatable."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    F = 7
    g = torch.Generator().manual_seed(11 + B)
    x = (torch.randn(B, F, HW, C, generator=g) * torch.rand(B, F, HW, C, generator=g).mul(4).exp2()).to(dev())
    w = (torch.randn(Co, C, 3, 1, generator=g) * 0.08).to(dev())
    b = torch.randn(Co, generator=g).to(dev())
    rowvec = torch.randn(B, Co, generator=g).to(dev()) if epi != "bias" else None
    res = torch.randn(B, F, HW, Co, generator=g).to(dev()) if epi == "residual" else None
    wp = ops.pack_weight(w, 0)
    y, st = ops.conv2d(x, wp, b, Co, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=F * HW, residual=res, want_stats=True)
    assert ops.last_kernel[0].startswith("conv_frames_x3") and st is not None, ops.last_kernel[0]
    xin = x.permute(0, 3, 1, 2).cpu().double()
    for n in (0, B // 2, B - 1):
        ref = F_conv(xin[n:n + 1], w.cpu().double(), b.cpu().double()).permute(0, 2, 3, 1)[0]
        if rowvec is not None:
            ref = ref + rowvec[n].cpu().double()
        if res is not None:
            ref = ref + res[n].cpu().double()
        err = (y[n].cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 5e-6, (n, err)
    rows = y.view(-1, 64, Co)                                          # natural block numbering: rows / 64
    assert torch.allclose(st[:, 0], rows.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(st[:, 1], (rows * rows).sum(1), rtol=1e-4, atol=1e-1)
    y2, st2 = ops.conv2d(x, wp, b, Co, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=F * HW, residual=res, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(st, st2)                 # fixed summation order
    # a row vector that is not one row per sample is outside the kernel's contract: the tap-by-tap kernel takes the launch
    if epi == "rowvec":
        rv2 = torch.randn(B * F, Co, generator=g).to(dev())
        y3 = ops.conv2d(x, wp, b, Co, 3, 1, (1, 1), (1, 0), rowvec=rv2, rows_per_batch=HW)
        assert not ops.last_kernel[0].startswith("conv_frames_x3")
        ref = F_conv(xin[:1], w.cpu().double(), b.cpu().double()).permute(0, 2, 3, 1)[0] + rv2[:F].cpu().double()[:, None, :]
        assert (y3[0].cpu().double() - ref).abs().max().item() / ref.abs().max().item() < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Ci,Co,ups,res", [(7, 64, 64, 64, 256, False, False), (56, 32, 32, 32, 128, False, True), (28, 16, 16, 96, 256, True, False),
                                                 (4, 48, 64, 160, 640, False, True), (7, 64, 64, 128, 128, True, False)])
def test_three_plane_patch_conv_vs_fp64(N, H, W, Ci, Co, ups, res):
    """conv_patch_x3 (csrc/igemm_x3p.hip): 3x3 / stride 1 / pad 1 on 16 x 16 pixel patches x 128 output channels per persistent
    workgroup, reduction in (chunk, 16-channel half, filter row) phases -- one to five chunks, one and several output-channel tiles per
    patch, image borders on every side of a patch, a non-square map, fewer tiles than workgroup slots and several tiles per workgroup,
    the folded nearest x2 upsample (H, W = SOURCE sizes then), bias + fp32 residual.  Against fp64 torch; bitwise repeatable."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    g = torch.Generator().manual_seed(N * H + W + Ci)
    x = torch.randn(N, Ci, H, W, generator=g) * torch.rand(N, Ci, H, W, generator=g).mul(4).exp2()
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    b = torch.randn(Co, generator=g)
    OH, OW = (2 * H, 2 * W) if ups else (H, W)
    r = torch.randn(N, Co, OH, OW, generator=g) if res else None
    wp = ops.pack_weight(w.to(dev()), 0)
    xd, rd = nhwc(x), (nhwc(r) if res else None)
    y = ops.conv2d(xd, wp, b.to(dev()), Co, 3, 3, (1, 1), (1, 1), residual=rd, ups=ups)
    assert ops.last_kernel[0].startswith("conv_patch_x3"), ops.last_kernel[0]
    assert y.shape == (N, OH, OW, Co)
    for n in sorted({0, N // 2, N - 1}):
        xin = F.interpolate(x[n:n + 1].double(), scale_factor=2, mode="nearest") if ups else x[n:n + 1].double()
        ref = F.conv2d(xin, w.double(), b.double(), padding=1)
        if res:
            ref = ref + r[n:n + 1].double()
        err = (nchw(y[n:n + 1]).double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 5e-6, (n, err)
    y2 = ops.conv2d(xd, wp, b.to(dev()), Co, 3, 3, (1, 1), (1, 1), residual=rd, ups=ups)
    assert torch.equal(y, y2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Fr,H,W,C,Co,act", [(8, 7, 32, 32, 128, 128, "silu"), (2, 7, 64, 64, 256, 128, "silu"), (32, 7, 16, 16, 128, 128, "none")])
def test_three_plane_patch_conv_applies_groupnorm_in_its_loader(B, Fr, H, W, C, Co, act):
    """GroupNorm32 + SiLU folded into conv_patch_x3 (v2a_groupnorm_stats_f32 -> v2a_conv2d_fwd_x3p_gn): statistics from the producing
    conv's 64-row blocks, the affine + activation applied to the halo in registers, padding ring left at zero.  Against fp64
    GroupNorm -> SiLU -> conv2d (the fp32 budget: the normalised values are fp32 before they are split into planes), and within fp32
    round-off of the unfused pair (materialised GroupNorm, then the same conv kernel)."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    N, S = B * Fr, Fr * H * W
    assert ops.conv2d_x3p_gn_ok(N, H, W, C, Co)
    g = torch.Generator().manual_seed(B + C)
    x = (torch.randn(B, S, C, generator=g) * 2 + torch.randn(1, 1, C, generator=g)).to(dev())
    gamma, beta = torch.randn(C, generator=g).to(dev()), torch.randn(C, generator=g).to(dev())
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.05
    b = torch.randn(Co, generator=g)
    rows = x.view(-1, 64, C)
    stats = torch.stack([rows.sum(1), (rows * rows).sum(1)], dim=1).contiguous()          # what a conv epilogue leaves: [M/64][2][C]
    pg = ops.groupnorm_prep_f32(x, gamma, beta, 32, act, stats=stats)
    assert pg is not None
    wp = ops.pack_weight(w.to(dev()), 0)
    y = ops.conv2d_x3p_gn(pg, x.view(N, H, W, C), wp, b.to(dev()), Co, Fr)
    # unfused pair on the same kernels
    y1 = ops.conv2d(pg.apply().view(N, H, W, C), wp, b.to(dev()), Co, 3, 3, (1, 1), (1, 1))
    assert ops.last_kernel[0].startswith("conv_patch_x3")
    d = (y - y1).abs().max().item() / y1.abs().max().item()
    assert d < 2e-6, d
    # fp64 reference on two samples
    for n in (0, B - 1):
        xs = x[n].double().cpu().view(Fr, H, W, C).permute(3, 0, 1, 2)[None]              # [1, C, F, H, W]: GroupNorm over (C/32, F, H, W)
        a = F.group_norm(xs, 32, gamma.double().cpu(), beta.double().cpu(), 1e-5)
        if act == "silu":
            a = F.silu(a)
        a = a[0].permute(1, 0, 2, 3)                                                       # [F, C, H, W]
        ref = F.conv2d(a, w.double(), b.double(), padding=1)
        got = y.view(B, Fr, H, W, Co)[n].permute(0, 3, 1, 2).double().cpu()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, (n, err)


@pytest.mark.gpu
@pytest.mark.parametrize("N,Ci,Co,H,k,res", [(64, 128, 64, 16, 3, True), (64, 256, 128, 8, 3, False), (64, 512, 256, 4, 3, True), (64, 128, 64, 16, 1, False),
                                             (8, 96, 64, 12, 3, False)])
def test_stride2_data_gradient_walks_live_taps_only(N, Ci, Co, H, k, res):
    """Data gradient of a stride-2 conv = a conv over the zero-interleaved output gradient (idil = 2).  Round 6: tile rows are numbered
    by parity class of the output pixel and a tile's K loop visits only the taps that meet stored pixels (1 / 2 / 2 / 4 of 9; 1 / 0 / 0 / 0
    for the 1 x 1 downsample) -- against torch's conv_transpose2d in fp64, and against the all-taps form of round 5 (the skipped products
    are exact zeros: equal to fp32 reassociation across the different split-K partitions).  dy [N, H, H, Ci] -> dx [N, 2H, 2H, Co]."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    g = torch.Generator().manual_seed(N + Ci + k)
    pad = k // 2
    w = torch.randn(Ci, Co, k, k, generator=g) * 0.05                    # forward conv: Co -> Ci channels, stride 2 (torch layout [out, in, kh, kw])
    dy = torch.randn(N, Ci, H, H, generator=g)
    r = torch.randn(N, Co, 2 * H, 2 * H, generator=g) if res else None
    ref = F.conv_transpose2d(dy.double(), w.double(), stride=2, padding=pad, output_padding=2 * H - ((H - 1) * 2 - 2 * pad + k))
    if res:
        ref = ref + r.double()
    # the data-gradient operand: flipped taps, in / out channels swapped -> an ordinary forward pack [Co][k][k][Ci]
    wd = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()
    wp = ops.pack_weight(wd.to(dev()), 0) if k > 1 else wd.reshape(Co, Ci).contiguous().to(dev())
    outs = {}
    for on in (1, 0):
        old = lib.v2a_debug_set_parity_classes(on)
        try:
            y = ops.conv2d(nhwc(dy), wp, None, Co, k, k, (1, 1), (k - 1 - pad, k - 1 - pad), idil=2, out_hw=(2 * H, 2 * H),
                           residual=nhwc(r) if res else None)
        finally:
            lib.v2a_debug_set_parity_classes(old)
        outs[on] = y
        err = (nchw(y).double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 5e-6, (on, err)
    d = (outs[1] - outs[0]).abs().max().item() / outs[0].abs().max().item()
    assert d < 2e-6, d


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Ci,Co", [(7, 64, 64, 64, 128), (28, 16, 16, 96, 256), (10, 32, 48, 32, 128)])
def test_upsample_conv_as_four_class_convs_vs_fp64(N, H, W, Ci, Co):
    """Upsample (nearest x2) + 3x3 conv as four 2x2 convs over the source map (conv_patch_x3<.., 2>, v2a_pack_weight_ups4): the filter
    taps that read the same source pixel are summed in advance, every output pixel of parity class (oh & 1, ow & 1) gets its own 2 x 2
    filter and window origin.  Borders on every side (the up-sampled map's zero padding falls on source pixels outside the map),
    several tiles per workgroup, a non-square map; against fp64 interpolate -> conv2d and against the gather form (H, W = source sizes)."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    assert ops.conv2d_x3p_ups4_ok(N, 2 * H, 2 * W, Ci, Co)
    g = torch.Generator().manual_seed(N * H + W + Ci)
    x = torch.randn(N, Ci, H, W, generator=g) * torch.rand(N, Ci, H, W, generator=g).mul(4).exp2()
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    b = torch.randn(Co, generator=g)
    wp = ops.pack_weight(w.to(dev()), 0)
    w4 = ops.pack_weight_ups4(wp, Co, Ci)
    # the class filters are sums of the 3x3 taps: class (1, 0), tap (0, 1) = w[kh 0..1][kw 1..2] summed
    want = (w[:, :, 0:2, 1:3].double().sum((2, 3))).float()
    assert torch.allclose(w4[2, :, 0, 1, :].cpu(), want, rtol=1e-6, atol=1e-6)
    xd = nhwc(x)
    y = ops.conv2d_x3p_ups4(xd, w4, b.to(dev()), Co)
    assert y.shape == (N, 2 * H, 2 * W, Co)
    y1 = ops.conv2d(xd, wp, b.to(dev()), Co, 3, 3, (1, 1), (1, 1), ups=True)
    d = (y - y1).abs().max().item() / y1.abs().max().item()
    assert d < 3e-6, d
    for n in sorted({0, N - 1}):
        ref = F.conv2d(F.interpolate(x[n:n + 1].double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
        err = (nchw(y[n:n + 1]).double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 5e-6, (n, err)
    assert torch.equal(y, ops.conv2d_x3p_ups4(xd, w4, b.to(dev()), Co))


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,Ci,Co,k,two", [(64, 4, 1024, 1024, 5, False), (64, 8, 512, 512, 5, False), (64, 16, 256, 256, 5, True), (128, 4, 64, 128, 3, False)])
def test_short_conv1d_skips_the_taps_that_meet_the_padding(B, T, Ci, Co, k, two):
    """ConditionalUnet1D Conv1d (k = 5, pad 2 over T = 4 / 8 / 16 positions; conv1d_components.py:23-40) on conv_igemm_f32x3 with one row
    class per output position: a tile's K loop visits only the taps that land inside the sequence (3, 4, 4, 3 of 5 at T = 4).  The skipped
    products were exact zeros: against fp64 conv1d, and against the all-taps form up to the different split-K partitions; deferred
    split-K slabs (summed here by hand) and a two-source (concat) input included."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    if lib.v2a_get_f32_conv_mode() != 1 or lib.v2a_get_precision() != 0:
        pytest.skip("fp32 three-plane mode only")
    g = torch.Generator().manual_seed(B + T + Ci)
    x = torch.randn(B, Ci, T, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) * 0.03
    b = torch.randn(Co, generator=g)
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=k // 2).permute(0, 2, 1)          # [B, T, Co]
    xd = x.permute(0, 2, 1).contiguous().to(dev())                                               # [B, T, Ci]
    wp = ops.pack_weight(w.view(Co, Ci, 1, k).to(dev()), 0)
    C1 = Ci // 2 if two else Ci
    xa = xd[..., :C1].contiguous().view(B, 1, T, C1)
    xb = xd[..., C1:].contiguous().view(B, 1, T, Ci - C1) if two else None
    outs = {}
    for on in (1, 0):
        old = lib.v2a_debug_set_parity_classes(on)
        try:
            y = ops.conv2d(xa, wp, b.to(dev()), Co, 1, k, (1, 1), (0, k // 2), x2=xb)
            yd, sl = ops.conv2d(xa, wp, b.to(dev()), Co, 1, k, (1, 1), (0, k // 2), x2=xb, defer=True)
        finally:
            lib.v2a_debug_set_parity_classes(old)
        if sl is not None:                                    # unreduced slabs: their sum + bias is the tensor
            yd = sl.ws.view(torch.float32)[:sl.n * sl.stride].view(sl.n, -1).sum(0).view(B, 1, T, Co) + b.to(dev())
        outs[on] = y
        for name, t in (("finished", y), ("deferred", yd)):
            err = (t.view(B, T, Co).cpu().double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 5e-6, (on, name, err)
    d = (outs[1] - outs[0]).abs().max().item() / outs[0].abs().max().item()
    assert d < 2e-6, d


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W,Ci,Co", [(14, 64, 64, 64, 128), (28, 32, 32, 96, 256), (20, 32, 48, 32, 128)])
def test_16bit_upsample_conv_as_four_class_convs(N, H, W, Ci, Co, dt):
    """conv_patch_h_ups4 (csrc/igemm_hp.hip): Upsample + 3x3 of the 16-bit video UNet as four 2x2 class convs over the source map.  Against
    an fp64 conv of the same rounded inputs with the PRE-SUMMED, rounded class filters (what the kernel multiplies: within one 16-bit ulp
    of the stored result), and against the nine-tap gather form on conv_halo_h3 (differs by the second rounding of the summed weights:
    a few 16-bit ulps); borders on every side, several tiles per workgroup, a non-square map, both 16-bit formats; bitwise repeatable."""
    from v2a_hip import ops
    assert ops.conv2d_hp_ups4_ok(N, 2 * H, 2 * W, Ci, Co)
    g = torch.Generator().manual_seed(N * H + W + Ci)
    x = torch.randn(N, H, W, Ci, generator=g).to(dt).to(dev())
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    b = torch.randn(Co, generator=g).to(dev())
    wp = ops.pack_weight(w.to(dev()), 0)
    w4 = ops.cast_h(ops.pack_weight_ups4(wp, Co, Ci), dt)
    y = ops.conv2d_hp_ups4(x, w4, b, Co)
    assert y.shape == (N, 2 * H, 2 * W, Co) and y.dtype == dt
    assert torch.equal(y, ops.conv2d_hp_ups4(x, w4, b, Co))
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    # reference per class from the rounded class filters
    w4d = w4.float().cpu().double()                                     # [4, Co, 2, 2, Ci]
    for n in sorted({0, N - 1}):
        xs = x[n].float().cpu().double().permute(2, 0, 1)[None]        # [1, Ci, H, W]
        got = y[n].float().cpu().double()                               # [2H, 2W, Co]
        for cls in range(4):
            ph, pw = cls >> 1, cls & 1
            xp = F.pad(xs, (1 - pw, pw, 1 - ph, ph))                    # window origin (a - 1 + ph, b - 1 + pw)
            ref = F.conv2d(xp, w4d[cls].permute(0, 3, 1, 2).contiguous(), b.cpu().double())[0].permute(1, 2, 0)      # [H, W, Co]
            err = (got[ph::2, pw::2] - ref).abs()
            assert (err <= ref.abs() * eps + 2e-5 * ref.abs().max()).all(), (n, cls, float(err.max()))
    yh = ops.conv2d_h(x, ops.pack_weight_h(w.to(dev()), dtype=dt), b, Co, 3, 3, (1, 1), (1, 1), ups=True)
    d = (y.float() - yh.float()).abs()
    assert float(d.max()) <= 16 * eps * float(yh.float().abs().max()), float(d.max())


def F_conv(x, w, b):
    return F.conv2d(x, w, b, padding=(1, 0))


def test_two_host_threads_two_streams_do_not_disturb_each_other():
    """SURVEY 8b / VERDICT r4 next #5: the C ABI is "thread-safe for distinct streams ... no hidden global state".  Two host threads, each
    on a stream and a scratch lane of its own, run DIFFERENT operands through the entry points that used to hand operands over in process
    globals (fp32 conv with deferred split-K slabs -> GroupNorm that sums them and adds a post-activation addend) and through the 16-bit
    convs in DIFFERENT formats (bf16 in one thread, IEEE fp16 in the other: the format flag is per host thread).  ctypes releases the GIL
    inside every call, so the launches interleave at the C level.  Every result must be bit-equal to the same sequence run alone."""
    import threading
    from v2a_hip import ops
    dev = "cuda:0"

    def work(seed, half, out):
        g = torch.Generator().manual_seed(seed)
        B, T, C = 64, 4, 1024                                        # a deep small-M Conv1d: the plan splits K, the GroupNorm sums the slabs
        x = torch.randn(B, 1, T, C, generator=g).to(dev)
        w = (torch.randn(C, 5 * C, generator=g) * 0.02).to(dev)
        bias = torch.randn(C, generator=g).to(dev)
        gamma, beta = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
        post = torch.randn(B, T, C, generator=g).to(dev)
        xh = torch.randn(2, 16, 16, 128, generator=g).to(half).to(dev)
        wh = ops.pack_weight_h((torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(dev), dtype=half)
        res = []
        for it in range(12):
            y, sl = ops.conv2d(x, w, bias, C, 1, 5, (1, 1), (0, 2), defer=True)
            assert sl is not None and sl.n > 1
            a, mean, rstd = ops.groupnorm_fwd(y.view(B, T, C), gamma, beta, 8, "mish", slabs=sl, post=post)
            yh = ops.conv2d_h(xh, wh, None, 128, 3, 3, (1, 1), (1, 1))
            assert yh.dtype == half
            if it in (0, 11):
                res.append((a.clone(), mean.clone(), yh.clone()))
        out.append(res)

    def run(seed, half, lane, out, err):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st), ops.ws_lane(lane):
                work(seed, half, out)
            st.synchronize()
        except BaseException as e:      # noqa: a failure in a thread must fail the test
            err.append(e)

    alone = {}
    for seed, half, lane in ((1, torch.bfloat16, 31), (2, torch.float16, 32)):
        out, err = [], []
        run(seed, half, lane, out, err)
        assert not err, err
        alone[seed] = out[0]
    outs, errs = {1: [], 2: []}, []
    th = [threading.Thread(target=run, args=(1, torch.bfloat16, 31, outs[1], errs)),
          threading.Thread(target=run, args=(2, torch.float16, 32, outs[2], errs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for seed in (1, 2):
        for (a0, m0, h0), (a1, m1, h1) in zip(alone[seed], outs[seed][0]):
            assert torch.equal(a0, a1) and torch.equal(m0, m1) and torch.equal(h0.view(torch.int16), h1.view(torch.int16)), seed
    assert not torch.equal(alone[1][0][0], alone[2][0][0])
