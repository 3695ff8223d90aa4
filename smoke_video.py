"""Tiny video-UNet denoise step on cuda:0 checked against the CPU oracle.  Checker-side code (it imports oracle/): lives next to
__graft_entry__.py, not inside the product package."""
import torch


def video_smoke():
    from flowdiffusion.flowdiffusion.unet import Unet_Tiny
    from oracle.param_fill import fill_module
    from oracle.video_unet import UNetCfg, unet_libero_forward
    torch.manual_seed(0)
    m = Unet_Tiny()
    sd = fill_module(m, seed=5)
    m = m.to("cuda:0").eval()
    g = torch.Generator().manual_seed(2)
    x, t, te = torch.randn(1, 12, 32, 32, generator=g), torch.tensor([33]), torch.randn(1, 5, 512, generator=g)
    y = m(x.cuda(), t.cuda(), te.cuda()).cpu()
    cfg = UNetCfg(in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
                  num_head_channels=16)
    with torch.no_grad():
        yo = unet_libero_forward(sd, x, t, te, cfg)
    err = ((y - yo).abs().max() / yo.abs().max()).item()
    assert err <= 1e-4, err
    print(f"[smoke] video UNet step rel err {err:.2e}")
