"""Video path parity on the GPU: HIP UNet / sampler (through Unet_Libero / GoalGaussianDiffusion surfaces) vs golden vectors
produced by the reference itself and vs the CPU oracle.  fp32, tolerance 1e-4 relative (sampler: per final frame)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _tiny():
    from flowdiffusion.flowdiffusion.unet import Unet_Tiny
    from oracle.param_fill import fill_module
    from tools_wsum import wsum
    torch.manual_seed(0)
    m = Unet_Tiny()
    sd = fill_module(m, seed=11)
    return m.to("cuda:0").eval(), sd, wsum(sd)


def test_tiny_unet_forward_vs_golden_and_oracle(golden_dir):
    from oracle.video_unet import UNetCfg, unet_libero_forward
    g = np.load(f"{golden_dir}/unet_tiny.npz", allow_pickle=True)
    m, sd, ws = _tiny()
    assert abs(ws - float(g["weights_abs_sum"])) < 1e-6 * ws
    x, t, te = torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_t"]), torch.from_numpy(g["fwd_te"])
    y = m(x.cuda(), t.cuda(), te.cuda())
    assert rel(y, g["fwd_y"]) <= TOL, rel(y, g["fwd_y"])
    cfg = UNetCfg(in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
                  num_head_channels=16)
    with torch.no_grad():
        yo = unet_libero_forward(sd, x, t, te, cfg)
    assert rel(yo, g["fwd_y"]) < 1e-6          # oracle == reference
    assert rel(y, yo) <= TOL


@pytest.mark.parametrize("name,steps,gw", [("ddpm100", 100, 0.0), ("ddim50", 50, 0.0), ("ddim10_cfg", 10, 1.5)])
def test_sampler_vs_golden(golden_dir, name, steps, gw):
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    g = np.load(f"{golden_dir}/unet_tiny.npz", allow_pickle=True)
    m, _, _ = _tiny()
    d = GoalGaussianDiffusion(m, image_size=(32, 32), channels=9, timesteps=100, sampling_timesteps=steps, loss_type="l2",
                              objective="pred_v", beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=gw).to("cuda:0")
    torch.manual_seed(1234)          # the reference's CPU stream: randn(shape) then one randn_like per step
    d.__dict__["_noise_hook"] = lambda shape: torch.randn(shape)
    out = d.sample(torch.from_numpy(g["x_cond"]).cuda(), torch.from_numpy(g["fwd_te"]).cuda(), batch_size=2)
    ref = g[f"sample_{name}"]
    assert out.shape == ref.shape and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    err = rel(out, ref)
    assert err <= (5e-4 if steps == 100 else 2e-4), err     # 100 sequential fp32 UNet calls: budget 5e-4


def test_full_unet_libero_forward_vs_golden(golden_dir):
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    from oracle.param_fill import fill_module
    from tools_wsum import wsum
    g = np.load(f"{golden_dir}/unet_libero_full.npz")
    torch.manual_seed(0)
    m = Unet_Libero()
    sd = fill_module(m, seed=12)
    assert abs(wsum(sd) - float(g["weights_abs_sum"])) < 1e-6 * float(g["weights_abs_sum"])
    m = m.to("cuda:0").eval()
    gen = torch.Generator().manual_seed(101)
    x = torch.randn(1, 24, 128, 128, generator=gen)
    t = torch.tensor([41])
    te = torch.randn(1, 10, 512, generator=gen)
    y = m(x.cuda(), t.cuda(), te.cuda())
    yf = y.flatten().cpu()
    got = yf[torch.from_numpy(g["idx"])]
    scale = np.abs(g["y_sampled"]).max()
    assert np.abs(got.numpy() - g["y_sampled"]).max() <= TOL * scale
    assert abs(float(yf.double().abs().sum()) - float(g["y_abs_sum"])) <= 1e-5 * float(g["y_abs_sum"])


def test_cpu_unet_raises():
    from flowdiffusion.flowdiffusion.unet import Unet_Tiny
    with pytest.raises(RuntimeError):
        Unet_Tiny()(torch.zeros(1, 12, 32, 32), torch.zeros(1, dtype=torch.long), torch.zeros(1, 4, 512))


@pytest.mark.parametrize("cls_name,ci,res,frames", [("UnetThor", 3, 16, 2), ("UnetMWFlow", 2, 32, 2), ("UnetBridge", 3, 16, 2)])
def test_other_unet_wrappers_vs_oracle(cls_name, ci, res, frames):
    """the other AVDC wrappers (same kernels, other hyper-parameters; UnetBridge has 160 base channels -> GroupNorm
    groups of 5/10/20 channels, UnetMWFlow packs 2 flow channels per frame) against the CPU oracle with the same parameters."""
    import flowdiffusion.flowdiffusion.unet as U
    from oracle.param_fill import fill_module
    from oracle.video_unet import UNetCfg, unet_forward
    torch.manual_seed(0)
    m = getattr(U, cls_name)()
    sd = fill_module(m, seed=21)
    cfgm = m.unet
    cfg = UNetCfg(in_channels=cfgm.in_channels, model_channels=cfgm.model_channels, out_channels=cfgm.out_channels,
                  num_res_blocks=cfgm.num_res_blocks, attention_resolutions=cfgm.attention_resolutions, channel_mult=cfgm.channel_mult,
                  num_head_channels=32)
    g = torch.Generator().manual_seed(5)
    B, H, W = 1, res, res
    x = torch.randn(B, frames * ci + 3, H, W, generator=g)
    t = torch.tensor([17])
    te = torch.randn(B, 4, 512, generator=g)
    y = m.to("cuda:0").eval()(x.cuda(), t.cuda(), te.cuda()).cpu()
    # oracle: the reference's own pack (unet.py:28-35 / :87-92) then UNetModel.forward
    cond = x[:, -3:, None].expand(B, 3, frames, H, W)
    xx = x[:, :-3].reshape(B, frames, ci, H, W).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        yo = unet_forward(sd, torch.cat([xx, cond], 1), t, te, cfg, pre="unet.")
    yo = yo.permute(0, 2, 1, 3, 4).reshape(B, frames * cfgm.out_channels, H, W)
    assert rel(y, yo) <= TOL, rel(y, yo)


@pytest.mark.gpu
def test_bf16_storage_full_unet_tracks_fp32(golden_dir):
    """bf16-storage configuration (bf16 activations / weights in HBM, fp32 accumulation): the full-size Unet_Libero forward stays
    within bf16 rounding noise of the fp32 parity path -- and its own golden samples stay within 3 % of the reference's."""
    import v2a_hip
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    from oracle.param_fill import fill_module
    torch.manual_seed(0)
    m = Unet_Libero()
    fill_module(m, seed=11)
    m = m.to("cuda:0").eval()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 24, 128, 128, generator=g).to("cuda:0")
    t = torch.tensor([7, 93], device="cuda:0")
    emb = torch.randn(2, 5, 512, generator=g).to("cuda:0")
    ref = m(x, t, task_embed=emb)
    m.storage = "bf16"
    out = m(x, t, task_embed=emb)
    m.storage = "f32"
    again = m(x, t, task_embed=emb)
    assert rel(again, ref) < 1e-5                                   # switching back restores the parity path
    assert out.shape == ref.shape and out.dtype == torch.float32 and torch.isfinite(out).all()
    rl2 = ((out - ref).norm() / ref.norm()).item()
    print('bf16-storage relative L2 error', rl2, 'max', rel(out, ref))
    assert rl2 < 3e-2, rl2
    assert (out - ref).abs().max().item() < 0.1 * ref.abs().max().item()
    # process-wide default: models with narrow widths silently stay fp32, wide ones switch
    old = v2a_hip.set_video_storage("bf16")
    try:
        m2 = Unet_Libero().to("cuda:0")
        assert m2._engine().storage == "bf16"
    finally:
        v2a_hip.set_video_storage(old)


@pytest.mark.gpu
def test_highres_256x256x16_frames_config_properties():
    """BASELINE.json configs[4] shape (256x256, 16-frame sequence = 1 conditioning + 15 predicted frames): too large for the CPU
    oracle in a test, so it is covered by size-independent properties -- rows of a batch are computed independently (no
    cross-sample statistic: GroupNorm is per sample, attention per frame), the result equals the B=1 result row by row, and the
    bf16-storage configuration tracks the fp32 one."""
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    from oracle.param_fill import fill_module
    torch.manual_seed(0)
    m = Unet_Libero()
    fill_module(m, seed=11)
    m = m.to("cuda:0").eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 45 + 3, 256, 256, generator=g).to("cuda:0")
    t = torch.tensor([3, 77], device="cuda:0")
    emb = torch.randn(2, 6, 512, generator=g).to("cuda:0")
    both = m(x, t, task_embed=emb)
    assert both.shape == (2, 45, 256, 256) and torch.isfinite(both).all()
    for b in range(2):
        one = m(x[b:b + 1], t[b:b + 1], task_embed=emb[b:b + 1])
        assert rel(one[0], both[b]) < 2e-5, b
    m.storage = "bf16"
    half = m(x, t, task_embed=emb)
    assert ((half - both).norm() / both.norm()).item() < 3e-2
