"""Generate tests/golden/*.npz from the REFERENCE ITSELF (imported from /root/reference under
tools/ref_shims).  Build-container only: the GPU box has no /root/reference and only reads the
committed .npz files.  Fixtures hold inputs/seeds and expected outputs -- never reference source.

    python -m tools.make_golden            # all
    python -m tools.make_golden policy     # one group

Weights are not stored: oracle/param_fill.fill_module regenerates them bit-identically from
parameter NAMES on both sides; each fixture records a checksum of the weights it was made with.
"""
import sys
import random
import numpy as np
import torch

from tools.ref_build import build_ref_policy, build_ref_unet
from oracle.param_fill import fill_module

OUT = "tests/golden"
torch.set_num_threads(8)


def wsum(sd):
    """Order-independent weight checksum (float64 sum of |w| over canonical float tensors)."""
    return float(sum(v.double().abs().sum() for k, v in sorted(sd.items()) if torch.is_floating_point(v)))


def sample_idx(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (k,), generator=g).numpy()


def g_tables():
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    d = GoalGaussianDiffusion(torch.nn.Identity(), image_size=(128, 128), channels=21, timesteps=100,
                              sampling_timesteps=100, loss_type="l2", objective="pred_v", beta_schedule="cosine",
                              min_snr_loss_weight=True, guidance_weight=0)
    out = {k: v.numpy() for k, v in d.state_dict().items() if v.dim() == 1 and v.numel() == 100}
    assert len(out) == 13
    times = torch.linspace(-1, 99, steps=51)
    times = list(reversed(times.int().tolist()))
    out["ddim_pairs_100_50"] = np.array(list(zip(times[:-1], times[1:])), dtype=np.int64)
    # third-party restated (diffusers, via tools/ref_shims): tagged, parity unpinned
    from diffusers.schedulers.scheduling_ddpm import DDPMScheduler
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler
    s = DDPMScheduler(num_train_timesteps=100, beta_schedule="squaredcos_cap_v2")
    out["thirdparty_squaredcos_alphas_cumprod"] = s.alphas_cumprod.numpy()
    s2 = DDIMScheduler(num_train_timesteps=100, beta_schedule="squaredcos_cap_v2")
    s2.set_timesteps(8)
    out["thirdparty_ddim8_timesteps"] = s2.timesteps.numpy()
    np.savez_compressed(f"{OUT}/tables.npz", **out)
    print("tables", {k: v.shape for k, v in out.items()})


def g_unet_tiny():
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    torch.manual_seed(0)
    m = build_ref_unet(tiny=True).eval()
    sd = fill_module(m, seed=11)
    out = {"weights_abs_sum": wsum(sd), "param_names": np.array(sorted(n for n, _ in m.named_parameters()))}
    g = torch.Generator().manual_seed(100)
    B, f, H, W = 2, 3, 32, 32
    x = torch.randn(B, (f + 1) * 3, H, W, generator=g)
    t = torch.tensor([7, 93])
    te = torch.randn(B, 6, 512, generator=g)
    with torch.no_grad():
        y = m(x, t, te)
    out.update(fwd_x=x.numpy(), fwd_t=t.numpy(), fwd_te=te.numpy(), fwd_y=y.numpy())
    x_cond = torch.rand(B, 3, H, W, generator=g)
    out["x_cond"] = x_cond.numpy()
    for name, steps, gw in [("ddpm100", 100, 0.0), ("ddim50", 50, 0.0), ("ddim10_cfg", 10, 1.5)]:
        d = GoalGaussianDiffusion(m, image_size=(H, W), channels=3 * f, timesteps=100, sampling_timesteps=steps,
                                  loss_type="l2", objective="pred_v", beta_schedule="cosine",
                                  min_snr_loss_weight=True, guidance_weight=gw).eval()
        torch.manual_seed(1234)   # noise stream: randn(shape) then one randn_like per step (Appendix A item 7)
        with torch.no_grad():
            img = d.sample(x_cond, te, batch_size=B)
        out[f"sample_{name}"] = img.numpy()
    # round 3: the other two objectives of model_predictions (:521-532), var_temp (:365,:578) -- all by the reference's own sampler
    for name, steps, gw, obj, vt in [("ddpm100_pred_noise", 100, 0.0, "pred_noise", 1.0), ("ddim50_pred_noise", 50, 0.0, "pred_noise", 1.0),
                                     ("ddim50_pred_x0", 50, 0.0, "pred_x0", 1.0), ("ddim10_cfg_pred_x0", 10, 1.5, "pred_x0", 1.0),
                                     ("ddim10_cfg_pred_noise", 10, 1.5, "pred_noise", 1.0), ("ddpm100_vt06", 100, 0.0, "pred_v", 0.6)]:
        d = GoalGaussianDiffusion(m, image_size=(H, W), channels=3 * f, timesteps=100, sampling_timesteps=steps,
                                  loss_type="l2", objective=obj, beta_schedule="cosine",
                                  min_snr_loss_weight=True, guidance_weight=gw, var_temp=vt).eval()
        torch.manual_seed(1234)
        with torch.no_grad():
            img = d.sample(x_cond, te, batch_size=B)
        out[f"sample_{name}"] = img.numpy()
    np.savez_compressed(f"{OUT}/unet_tiny.npz", **out)
    print("unet_tiny ok", out["weights_abs_sum"])


def g_video_train():
    """GoalGaussianDiffusion.forward (training loss) + autograd gradients + 3 Trainer-style steps (clip 1.0 -> Adam -> EMA) of the tiny UNet."""
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    torch.manual_seed(0)
    m = build_ref_unet(tiny=True).train()
    sd = fill_module(m, seed=11)
    sd0 = {n: p.detach().clone() for n, p in m.named_parameters()}
    B, f, H, W = 2, 3, 32, 32
    g = torch.Generator().manual_seed(300)
    img = torch.rand(B, 3 * f, H, W, generator=g)
    cond = torch.rand(B, 3, H, W, generator=g)
    te = torch.randn(B, 5, 512, generator=g)
    out = {"weights_abs_sum": wsum(sd), "img": img.numpy(), "cond": cond.numpy(), "te": te.numpy()}
    names = [n for n, _ in m.named_parameters()]
    out["param_names"] = np.array(names)
    for tag, lt, obj in (("l2_v", "l2", "pred_v"), ("l1_noise", "l1", "pred_noise")):
        d = GoalGaussianDiffusion(m, image_size=(H, W), channels=3 * f, timesteps=100, sampling_timesteps=100, loss_type=lt, objective=obj,
                                  beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0)
        m.zero_grad()
        torch.manual_seed(77)
        loss = d(img, cond, te)
        loss.backward()
        torch.manual_seed(77)                         # replay the draws: t = randint (:718), noise = randn_like (:692)
        t = torch.randint(0, 100, (B,)).long()
        noise = torch.randn(B, 3 * f, H, W)
        P = dict(m.named_parameters())
        gn, gs = [], []
        for n in names:
            gr = P[n].grad
            gn.append(float(gr.double().norm()))
            gs.append(gr.flatten()[sample_idx(gr.numel(), 8, 9)].numpy())
        out.update({f"{tag}_loss": loss.item(), f"{tag}_t": t.numpy(), f"{tag}_noise": noise.numpy(), f"{tag}_grad_norms": np.array(gn),
                    f"{tag}_grad_samples": np.stack(gs)})
    # Trainer.train's arithmetic (:962-996) on a fixed batch, ema_pytorch restated (third party): update_every=2 here so that the
    # copy / lerp branches are both visited in a few steps (update_after_step=2)
    from ema_pytorch import EMA
    d = GoalGaussianDiffusion(m, image_size=(H, W), channels=3 * f, timesteps=100, sampling_timesteps=100, loss_type="l2", objective="pred_v",
                              beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0)
    opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.9, 0.99))
    ema = EMA(d, beta=0.995, update_every=2, update_after_step=2)
    m.zero_grad()
    losses, norms, ts, nzs = [], [], [], []
    for it in range(6):
        torch.manual_seed(80 + it)
        l = d(img, cond, te)
        l.backward()
        torch.manual_seed(80 + it)
        ts.append(torch.randint(0, 100, (B,)).long().numpy())
        nzs.append(torch.randn(B, 3 * f, H, W).numpy())
        norms.append(float(torch.nn.utils.clip_grad_norm_(d.parameters(), 1.0)))
        opt.step(); opt.zero_grad(); ema.update()
        losses.append(l.item())
    P = dict(m.named_parameters())
    E = dict(ema.ema_model.model.named_parameters())
    out.update(train_losses=np.array(losses), train_gnorms=np.array(norms), train_t=np.stack(ts), train_noise=np.stack(nzs),
               train_param_norms=np.array([float(P[n].double().norm()) for n in names]),
               train_ema_norms=np.array([float(E[n].double().norm()) for n in names]),
               train_param_delta=np.array([float((P[n].detach().double() - sd0[n].double()).norm()) for n in names]),
               thirdparty_note=np.array("Adam / clip_grad_norm_ from torch (present); EMA via restated ema_pytorch 0.2.3: parity unpinned"))
    np.savez_compressed(f"{OUT}/video_train.npz", **out)
    print("video_train ok", out["l2_v_loss"], out["l1_noise_loss"], losses, norms)


TRANSFORMER_CFGS = {
    # the shape family of TransformerNet (diffusion_policy_baseline/unet.py:57-70), scaled down
    "dec_causal": dict(input_dim=7, output_dim=7, horizon=16, n_obs_steps=3, cond_dim=64, n_layer=2, n_head=4, n_emb=64, p_drop_emb=0.0,
                       p_drop_attn=0.0, causal_attn=True, time_as_cond=True, obs_as_cond=True, n_cond_layers=2),
    "dec_mlp": dict(input_dim=4, output_dim=4, horizon=10, n_obs_steps=2, cond_dim=32, n_layer=1, n_head=2, n_emb=96, p_drop_emb=0.0,
                    p_drop_attn=0.0, causal_attn=False, time_as_cond=True, obs_as_cond=True, n_cond_layers=0),
    "bert_causal": dict(input_dim=7, output_dim=7, horizon=12, n_layer=2, n_head=4, n_emb=64, p_drop_emb=0.0, p_drop_attn=0.0,
                        causal_attn=True, time_as_cond=False),
}


def g_transformer():
    """TransformerForDiffusion forward + autograd gradients (loss = <out, R>) from the reference class, three structural variants."""
    from flowdiffusion.flowdiffusion.diffusion_policy_baseline.transformer_for_diffusion import TransformerForDiffusion as Ref
    assert "/root/reference" in sys.modules[Ref.__module__].__file__
    out = {}
    for tag, cfg in TRANSFORMER_CFGS.items():
        torch.manual_seed(0)
        m = Ref(**cfg).train()
        sd = fill_module(m, seed=21)
        g = torch.Generator().manual_seed(400)
        B, T = 3, cfg["horizon"]
        x = torch.randn(B, T, cfg["input_dim"], generator=g, requires_grad=True)
        t = torch.tensor([3, 50, 99])
        cond = torch.randn(B, cfg["n_obs_steps"], cfg["cond_dim"], generator=g, requires_grad=True) if cfg.get("cond_dim", 0) > 0 else None
        R = torch.randn(B, T, cfg["output_dim"], generator=g)
        y = m(x, t, cond)
        (y * R).sum().backward()
        names = [n for n, _ in m.named_parameters()]
        P = dict(m.named_parameters())
        out.update({f"{tag}_x": x.detach().numpy(), f"{tag}_t": t.numpy(), f"{tag}_R": R.numpy(), f"{tag}_y": y.detach().numpy(),
                    f"{tag}_dx": x.grad.numpy(), f"{tag}_names": np.array(names), f"{tag}_wsum": wsum({k: v for k, v in sd.items() if "mask" not in k}),
                    f"{tag}_grad_norms": np.array([float(P[n].grad.double().norm()) for n in names]),
                    f"{tag}_grad_samples": np.stack([P[n].grad.flatten()[sample_idx(P[n].numel(), 6, 9)].numpy() for n in names]),
                    f"{tag}_state_keys": np.array(list(m.state_dict().keys()))})
        if cond is not None:
            out.update({f"{tag}_cond": cond.detach().numpy(), f"{tag}_dcond": cond.grad.numpy()})
        print(tag, "ok", float(y.abs().max()), len(names))
    np.savez_compressed(f"{OUT}/transformer.npz", **out)


def g_unet_full():
    torch.manual_seed(0)
    m = build_ref_unet(tiny=False).eval()
    sd = fill_module(m, seed=12)
    g = torch.Generator().manual_seed(101)
    x = torch.randn(1, 24, 128, 128, generator=g)
    t = torch.tensor([41])
    te = torch.randn(1, 10, 512, generator=g)
    with torch.no_grad():
        y = m(x, t, te)
    idx = sample_idx(y.numel(), 4096, 5)
    np.savez_compressed(f"{OUT}/unet_libero_full.npz", weights_abs_sum=wsum(sd), n_params=sum(p.numel() for p in m.parameters()),
                        idx=idx, y_sampled=y.flatten()[idx].numpy(), y_sum=float(y.double().sum()),
                        y_abs_sum=float(y.double().abs().sum()), n_state=len(sd))
    print("unet_full ok", float(y.double().abs().sum()))


def g_c3_row():
    """BASELINE configs[2] pinned on the REFERENCE: row 0 of the full-size 50-step DDIM sample (Unet_Libero, 8-frame 128 x 128) made by the
    imported reference's own GoalGaussianDiffusion.sample at batch 1 with the inputs tests/test_video_gpu.py feeds row 0 of its B = 16
    call (rows of a batch are independent).  ~5 minutes on 8 cores.  Stored: every second pixel of the row + sums of the whole row."""
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    torch.manual_seed(0)
    m = build_ref_unet(tiny=False).eval()
    sd = fill_module(m, seed=12)
    steps, B = 50, 16
    d = GoalGaussianDiffusion(m, image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=steps, loss_type="l2",
                              objective="pred_v", beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0)
    gen = torch.Generator().manual_seed(41)
    x_cond = torch.rand(B, 3, 128, 128, generator=gen)
    te = torch.randn(B, 10, 512, generator=gen)
    n0 = torch.randn(B, 21, 128, 128, generator=gen)
    calls = []
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def fake_randn(*shape, **kw):                 # the reference draws its start image with torch.randn(shape, device=...)
        calls.append(tuple(shape[0]) if isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape))
        return n0[:1].clone()

    def fake_randn_like(t, **kw):                 # eta = 0: sigma = 0 multiplies every later draw
        calls.append(tuple(t.shape))
        return torch.zeros_like(t)

    torch.randn, torch.randn_like = fake_randn, fake_randn_like
    try:
        with torch.no_grad():
            out = d.sample(x_cond[:1], te[:1], batch_size=1)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    assert out.shape == (1, 21, 128, 128), out.shape
    np.savez_compressed(f"{OUT}/c3_row.npz", weights_abs_sum=wsum(sd), steps=steps, batch=B, seed=41, n_noise_calls=len(calls),
                        row0_sub=out[0, :, ::2, ::2].numpy().astype(np.float32), row0_sum=float(out.double().sum()),
                        row0_abs_sum=float(out.double().abs().sum()), row0_sq_sum=float((out.double() ** 2).sum()))
    print("c3_row ok", out.shape, len(calls), float(out.double().abs().sum()))


def g_c5_row():
    """BASELINE configs[4]'s video half pinned on the REFERENCE at its full shape: 256 x 256, 1 conditioning + 15 predicted frames (45
    channels), the 201 M-parameter Unet_Libero, 2 DDIM steps at batch 1 made by the imported reference's own GoalGaussianDiffusion.sample
    (the attention layers see L = 1024 and 256 keys here).  ~4 minutes and ~12 GB on 8 cores.  Stored: every fourth pixel of every third
    frame channel + sums of the whole sample."""
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    torch.manual_seed(0)
    m = build_ref_unet(tiny=False).eval()
    sd = fill_module(m, seed=12)
    steps = 2
    d = GoalGaussianDiffusion(m, image_size=(256, 256), channels=45, timesteps=100, sampling_timesteps=steps, loss_type="l2",
                              objective="pred_v", beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0)
    gen = torch.Generator().manual_seed(43)
    x_cond = torch.rand(1, 3, 256, 256, generator=gen)
    te = torch.randn(1, 10, 512, generator=gen)
    n0 = torch.randn(1, 45, 256, 256, generator=gen)
    calls = []
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def fake_randn(*shape, **kw):
        calls.append(1)
        return n0.clone()

    def fake_randn_like(t, **kw):                 # eta = 0: sigma = 0 multiplies every later draw
        calls.append(1)
        return torch.zeros_like(t)

    torch.randn, torch.randn_like = fake_randn, fake_randn_like
    try:
        with torch.no_grad():
            out = d.sample(x_cond, te, batch_size=1)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    assert out.shape == (1, 45, 256, 256), out.shape
    np.savez_compressed(f"{OUT}/c5_row.npz", weights_abs_sum=wsum(sd), steps=steps, seed=43, n_noise_calls=len(calls),
                        row0_sub=out[0, ::3, ::4, ::4].numpy().astype(np.float32), row0_sum=float(out.double().sum()),
                        row0_abs_sum=float(out.double().abs().sum()), row0_sq_sum=float((out.double() ** 2).sum()))
    print("c5_row ok", out.shape, len(calls), float(out.double().abs().sum()))


def g_policy():
    torch.manual_seed(0)
    pol = build_ref_policy()
    sd = fill_module(pol, seed=13)
    names = [n for n, p in pol.named_parameters() if p.requires_grad and n not in ("_dummy_variable", "obs_encoder._dummy_variable")]
    out = {"weights_abs_sum": wsum(sd), "param_names": np.array(names), "state_keys": np.array(list(sd.keys())),
           "n_params": sum(p.numel() for p in pol.parameters())}
    g = torch.Generator().manual_seed(102)
    B = 2
    batch = {"obs": {"img_obs_1": torch.rand(B, 1, 3, 128, 128, generator=g),
                     "img_goal_1": torch.rand(B, 1, 3, 128, 128, generator=g)},
             "action": torch.rand(B, 16, 7, generator=g) * 2 - 1}
    out.update(img_obs=batch["obs"]["img_obs_1"].numpy(), img_goal=batch["obs"]["img_goal_1"].numpy(),
               action=batch["action"].numpy())
    # compute_loss RNG order in train mode: SpatialSoftmax draws randn_like(kp) per encoder (noise_std=0),
    # then noise = randn(B,16,7), then t = randint(0,100,(B,))  (base_nets.py:262-264; policy :246-252)
    pol.train()
    torch.manual_seed(55)
    loss = pol.compute_loss(batch)
    loss.backward()
    torch.manual_seed(55)
    torch.randn(B, 32, 2); torch.randn(B, 32, 2)
    noise = torch.randn(B, 16, 7)
    ts = torch.randint(0, 100, (B,)).long()
    out.update(loss=loss.item(), noise=noise.numpy(), timesteps=ts.numpy())
    gn, gs = [], []
    P = dict(pol.named_parameters())
    for n in names:
        gr = P[n].grad
        gn.append(float(gr.double().norm()))
        idx = sample_idx(gr.numel(), 8, 9)
        gs.append(gr.flatten()[idx].numpy())
    out.update(grad_norms=np.array(gn), grad_samples=np.stack(gs))
    # 3 optimiser steps: clip 1.0 -> AdamW -> zero -> EMA(power .75) exactly as lb_online_trainer_v7.py:604-624
    from ema_pytorch import EMA
    opt = torch.optim.AdamW(pol.parameters(), lr=1e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6)
    ema = EMA(pol, update_after_step=0, inv_gamma=1.0, power=0.75, min_value=0.0, update_every=1, include_online_model=False)
    losses, norms = [], []
    for it in range(3):
        torch.manual_seed(60 + it)
        l = pol.compute_loss(batch)
        opt.zero_grad()
        l.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(pol.parameters(), 1.0)))
        opt.step(); opt.zero_grad(); ema.update()
        losses.append(l.item())
    out.update(train_losses=np.array(losses), train_gnorms=np.array(norms),
               train_param_norms=np.array([float(P[n].double().norm()) for n in names]),
               train_ema_norms=np.array([float(dict(ema.ema_model.named_parameters())[n].double().norm()) for n in names]),
               thirdparty_note=np.array("AdamW/clip from torch (present); EMA via restated ema_pytorch 0.2.3; "
                                        "resnet18 topology + diffusers schedulers restated: parity unpinned"))
    # predict_action on the ORIGINAL weights
    pol2 = build_ref_policy()
    fill_module(pol2, seed=13)
    pol2.eval()
    torch.manual_seed(70)
    o = pol2.predict_action(batch["obs"], use_ddim=True)
    out.update(ddim_action=o["action"].numpy(), ddim_action_pred=o["action_pred"].numpy())
    torch.manual_seed(71)
    o = pol2.predict_action(batch["obs"], use_ddim=False)
    out.update(ddpm_action_pred=o["action_pred"].numpy())
    np.savez_compressed(f"{OUT}/policy.npz", **out)
    print("policy ok loss", out["loss"], "train", losses, norms)


def g_replay():
    from diffuser.datasets.env_img_replay_buffer import Global_EnvReplayBuffer_Img

    class _EnvList:
        camera_list = ["agentview"]

    out = {}
    for seed in (0, 123, 9001):
        rng = np.random.RandomState(seed + 1)
        lens = rng.randint(121, 145, size=12)
        buf = Global_EnvReplayBuffer_Img(task_list=["t"], max_num_unitBufs=1200, max_len_uB=700, min_len_uB=30,
                                         env_list=_EnvList(), render_img_size=(4, 4),
                                         env_buf_config={"sample_act_seq_len": 16})
        for e, L in enumerate(lens):
            # payload encodes (episode, frame) so the gather can be checked exactly: img[...] = e*1000 + i
            imgs = [torch.full((3, 4, 4), float(e * 1000 + i)) for i in range(L)]
            acts = [torch.full((7,), float(e * 1000 + i)) for i in range(L - 1)]
            buf.add_one_episode("t", "agentview", e, imgs, acts)
        np.random.seed(seed); random.seed(seed)
        eps, starts = [], []
        for _ in range(5):
            s, gl, a, _, info = buf.sample_random_batch_seq(64)
            ep = (s[:, 0, 0, 0] // 1000).long().numpy()
            st = (s[:, 0, 0, 0] % 1000).long().numpy()
            assert ((gl[:, 0, 0, 0] % 1000).long().numpy() == st + 16).all()
            assert (a[:, 0, 0].long().numpy() == ep * 1000 + st).all() and a.shape == (64, 16, 7)
            eps.append(ep); starts.append(st)
        out[f"lens_{seed}"] = lens
        out[f"episodes_{seed}"] = np.stack(eps)
        out[f"starts_{seed}"] = np.stack(starts)
    np.savez_compressed(f"{OUT}/replay.npz", **out)
    print("replay ok", out["episodes_123"][0][:8], out["starts_123"][0][:8])


def g_replay_mixed():
    """The two-buffer draw: LB_Online_Trainer_V7.sample_from_bufs in 'rand_prob' mode (lb_online_trainer_v7.py:826-851) + merge_batch
    (diffuser/models/train_utils.py:40-74), called on the REFERENCE method with an attribute bag for `self`; payload encodes
    (buffer, episode, frame) so that rows can be decoded from the returned tensors."""
    from types import SimpleNamespace
    from diffuser.datasets.env_img_replay_buffer import Global_EnvReplayBuffer_Img
    from diffuser.libero.lb_online_trainer_v7 import LB_Online_Trainer_V7 as Ref
    assert "/root/reference" in sys.modules[Ref.__module__].__file__

    class _EnvList:
        camera_list = ["agentview"]

    out = {}
    for seed in (0, 77, 4242):
        rng = np.random.RandomState(seed + 5)
        bufs, lens_all = [], []
        for which, n_ep in ((0, 10), (1, 6)):
            lens = rng.randint(40, 90, size=n_ep) if which == 0 else rng.randint(60, 200, size=n_ep)
            buf = Global_EnvReplayBuffer_Img(task_list=["t"], max_num_unitBufs=1200, max_len_uB=700, min_len_uB=30, env_list=_EnvList(),
                                             render_img_size=(4, 4), env_buf_config={"sample_act_seq_len": 16})
            for e, L in enumerate(lens):
                base = which * 1000000 + e * 1000
                buf.add_one_episode(f"task{which}_{e}", "agentview", e, [torch.full((3, 4, 4), float(base + i)) for i in range(L)],
                                    [torch.full((7,), float(base + i)) for i in range(L - 1)])
            bufs.append(buf); lens_all.append(lens)
        me = SimpleNamespace(envBuf_rand=bufs[0], envBuf_vid=bufs[1], buf_sample_batch_size=64, buf_sample_method='rand_prob',
                             buf_sample_randBuf_prob=0.3, input_img_size=(4, 4), init_rand_steps=10, iter_type='rand-bias')
        np.random.seed(seed); random.seed(seed)
        rows, nr = [], []
        for _ in range(4):
            s, gl, a, tasks, info = Ref.sample_from_bufs(me)
            code = s[:, 0, 0, 0].long().numpy()
            assert (gl[:, 0, 0, 0].long().numpy() == code + 16).all() and (a[:, 0, 0].long().numpy() == code).all()
            assert len(tasks) == 64 and len(info["cams_str"]) == 64
            which = code // 1000000
            assert (np.diff(which) >= 0).all()                      # rand rows first, rollout rows after
            rows.append(code); nr.append(int((which == 0).sum()))
        out[f"lens_rand_{seed}"], out[f"lens_vid_{seed}"] = lens_all
        out[f"codes_{seed}"] = np.stack(rows)
        out[f"n_rand_{seed}"] = np.array(nr)
        # the stream continues: one more np / python draw after the four batches pins the generator end states
        out[f"tail_{seed}"] = np.array([np.random.randint(0, 1 << 30), random.randint(0, 1 << 30)])
    np.savez_compressed(f"{OUT}/replay_mixed.npz", **out)
    print("replay_mixed ok", out["n_rand_77"], out["codes_77"][0][:6])


def g_policy_limits():
    """Non-identity action limits (lb_action_minmax_orn01: orientation channels in +-0.1, diffuser/datasets/__init__.py:30-37):
    compute_loss and DDIM-8 predict_action of the reference with that normaliser."""
    import tools.ref_build as RB
    from diffuser.datasets import lb_action_minmax_orn01_f
    import diffuser.datasets as D
    orig = D.lb_action_minmax_f
    D.lb_action_minmax_f = lb_action_minmax_orn01_f            # build_ref_policy imports the name at call time
    try:
        torch.manual_seed(0)
        pol = RB.build_ref_policy()
        sd = fill_module(pol, seed=13)
        assert float(pol.normalizer["action"].maxs.flatten()[4]) == np.float32(0.1)
        g = torch.Generator().manual_seed(102)
        B = 2
        batch = {"obs": {"img_obs_1": torch.rand(B, 1, 3, 128, 128, generator=g), "img_goal_1": torch.rand(B, 1, 3, 128, 128, generator=g)},
                 "action": torch.rand(B, 16, 7, generator=g) * 2 - 1}
        batch["action"][..., 3:6] *= 0.1
        pol.train()
        torch.manual_seed(55)
        loss = pol.compute_loss(batch)
        torch.manual_seed(55)
        torch.randn(B, 32, 2); torch.randn(B, 32, 2)
        noise = torch.randn(B, 16, 7)
        ts = torch.randint(0, 100, (B,)).long()
        pol.eval()
        torch.manual_seed(70)
        o = pol.predict_action(batch["obs"], use_ddim=True)
        np.savez_compressed(f"{OUT}/policy_orn01.npz", weights_abs_sum=wsum(sd), action=batch["action"].numpy(), loss=loss.item(),
                            noise=noise.numpy(), timesteps=ts.numpy(), ddim_action_pred=o["action_pred"].numpy(),
                            act_min=pol.normalizer["action"].mins.flatten().numpy(), act_max=pol.normalizer["action"].maxs.flatten().numpy())
        print("policy_orn01 ok loss", loss.item(), float(o["action_pred"].abs().max()))
    finally:
        D.lb_action_minmax_f = orig


from tests.tools_schedule import _schedule_trace, schedule_stub, SCHEDULE_CFGS  # noqa: E402  (shared with tests/test_joint_loop.py)


def g_schedule():
    from diffuser.libero.lb_online_trainer_v7 import LB_Online_Trainer_V7 as Ref
    assert "/root/reference" in sys.modules[Ref.__module__].__file__
    out = {}
    for name, cfg in SCHEDULE_CFGS.items():
        n = 30000 if name == "released" else 4000
        out[name] = np.packbits(_schedule_trace(Ref, lambda: schedule_stub(cfg), n), axis=0)
        out[name + "_n"] = np.array(n)
    np.savez_compressed(f"{OUT}/schedule.npz", **out)
    print("schedule ok", {k: v.shape for k, v in out.items()})


def g_wrappers():
    """The other AVDC wrappers (reference flowdiffusion/flowdiffusion/unet.py:7-192), forward outputs of the REFERENCE classes with
    name-derived weights: full output (small resolutions), inputs regenerated from seeds on the test side."""
    import flowdiffusion.flowdiffusion.unet as U
    assert "/root/reference" in U.__file__
    out = {}
    for cls_name, ci, res, frames in [("UnetThor", 3, 16, 2), ("UnetMWFlow", 2, 32, 2), ("UnetBridge", 3, 16, 2), ("UnetMW", 3, 32, 2)]:
        torch.manual_seed(0)
        m = getattr(U, cls_name)().eval()
        sd = fill_module(m, seed=21)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(1, frames * ci + 3, res, res, generator=g)
        t = torch.tensor([17])
        te = torch.randn(1, 4, 512, generator=g)
        with torch.no_grad():
            y = m(x, t, te)
        out[f"{cls_name}_y"] = y.numpy()
        out[f"{cls_name}_wsum"] = np.array(wsum(sd))
        print(cls_name, tuple(y.shape), float(y.abs().max()))
    np.savez_compressed(f"{OUT}/wrappers.npz", **out)


GROUPS = {"wrappers": g_wrappers, "tables": g_tables, "unet_tiny": g_unet_tiny, "unet_full": g_unet_full, "policy": g_policy, "replay": g_replay, "replay_mixed": g_replay_mixed, "policy_limits": g_policy_limits, "schedule": g_schedule, "video_train": g_video_train, "transformer": g_transformer, "c3_row": g_c3_row, "c5_row": g_c5_row}

if __name__ == "__main__":
    which = sys.argv[1:] or list(GROUPS)
    for w in which:
        GROUPS[w]()
