"""CPU suite: the oracle against the golden vectors generated from the reference itself (tests/golden/*.npz, made by
tools/make_golden.py in the build container).  These pin the oracle; the GPU suite then pins the HIP path against both."""
import numpy as np
import pytest
import torch


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_tables_vs_reference(golden_dir):
    from oracle import goal_diffusion as GD
    from oracle import schedulers as S
    g = np.load(f"{golden_dir}/tables.npz")
    T = GD.cosine_tables(100, min_snr_loss_weight=True, objective="pred_v")
    for k in GD.TABLE_NAMES:
        assert np.array_equal(T[k].numpy(), g[k]), k
    # known answers recorded in SURVEY.md 8a (V2)
    ac = T["alphas_cumprod"].numpy()
    assert abs(ac[0] - 0.99936873) < 1e-7 and abs(ac[50] - 0.47826463) < 1e-7 and abs(ac[99] - 2.4285723e-07) < 1e-12
    assert np.array_equal(np.array(GD.ddim_time_pairs(100, 50)), g["ddim_pairs_100_50"])
    assert GD.ddim_time_pairs(100, 50)[0] == (99, 97) and GD.ddim_time_pairs(100, 50)[-1] == (1, -1)
    # third-party restated (parity unpinned by reference tests): self-consistency with the generator's restatement
    assert np.array_equal(S.squaredcos_alphas_cumprod(100).numpy(), g["thirdparty_squaredcos_alphas_cumprod"])
    assert S.ddim_timesteps(100, 8) == [int(v) for v in g["thirdparty_ddim8_timesteps"]] == [84, 72, 60, 48, 36, 24, 12, 0]


def test_product_tables_match_reference(golden_dir):
    """Host logic of the product: GoalGaussianDiffusion's 13 registered buffers are bit-identical to the reference's."""
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    g = np.load(f"{golden_dir}/tables.npz")
    d = GoalGaussianDiffusion(torch.nn.Identity(), image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=100,
                              loss_type="l2", objective="pred_v", beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0)
    bufs = dict(d.named_buffers())
    assert len(bufs) == 13 and not d.is_ddim_sampling and d.num_timesteps == 100
    for k, v in bufs.items():
        assert v.dtype == torch.float32 and np.array_equal(v.numpy(), g[k]), k
    d2 = GoalGaussianDiffusion(torch.nn.Identity(), image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=50,
                               objective="pred_v", beta_schedule="cosine")
    assert d2.is_ddim_sampling and d2.guidance_weight == 2.0 and d2.var_temp == 1.0


def _tiny_sd():
    from flowdiffusion.flowdiffusion.unet import Unet_Tiny
    from oracle.param_fill import fill_module
    torch.manual_seed(0)
    m = Unet_Tiny()
    return m, fill_module(m, seed=11)


def test_oracle_unet_and_sampler_vs_reference(golden_dir):
    from oracle.video_unet import UNetCfg, unet_libero_forward
    from oracle import goal_diffusion as GD
    from tools_wsum import wsum
    g = np.load(f"{golden_dir}/unet_tiny.npz", allow_pickle=True)
    m, sd = _tiny_sd()
    assert sorted(n for n, _ in m.named_parameters()) == [str(x) for x in g["param_names"]]
    assert abs(wsum(sd) - float(g["weights_abs_sum"])) < 1e-9 * float(g["weights_abs_sum"])
    cfg = UNetCfg(in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
                  num_head_channels=16)
    x, t, te = torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_t"]), torch.from_numpy(g["fwd_te"])
    with torch.no_grad():
        y = unet_libero_forward(sd, x, t, te, cfg)
    assert rel(y, g["fwd_y"]) < 1e-6
    T = GD.cosine_tables(100)
    fn = lambda xx, tt, ee: unet_libero_forward(sd, xx, tt, ee, cfg)
    x_cond = torch.from_numpy(g["x_cond"])
    for name, steps, gw in [("ddim50", 50, 0.0), ("ddim10_cfg", 10, 1.5)]:       # (100-step DDPM is covered on the GPU)
        torch.manual_seed(1234)
        noises = [torch.randn(2, 9, 32, 32) for _ in range(steps + 1)]
        out = GD.sample(fn, T, noises, x_cond, te, guidance_weight=gw, num_timesteps=100, sampling_timesteps=steps)
        assert rel(out, g[f"sample_{name}"]) < 1e-5, name
    # round 3: pred_noise / pred_x0 (model_predictions :521-532, with and without classifier-free guidance) -- reference-made fixtures
    for name, steps, gw, obj in [("ddim50_pred_noise", 50, 0.0, "pred_noise"), ("ddim50_pred_x0", 50, 0.0, "pred_x0"),
                                 ("ddim10_cfg_pred_x0", 10, 1.5, "pred_x0"), ("ddim10_cfg_pred_noise", 10, 1.5, "pred_noise")]:
        torch.manual_seed(1234)
        noises = [torch.randn(2, 9, 32, 32) for _ in range(steps + 1)]
        out = GD.sample(fn, T, noises, x_cond, te, guidance_weight=gw, num_timesteps=100, sampling_timesteps=steps, objective=obj)
        assert rel(out, g[f"sample_{name}"]) < 1e-5, name


def test_oracle_unet_wrappers_vs_reference(golden_dir):
    """UnetThor / UnetMWFlow / UnetBridge / UnetMW (reference flowdiffusion/flowdiffusion/unet.py:7-192): the oracle's UNet under the
    wrappers' pack / unpack against forward outputs of the REFERENCE classes (tools/make_golden.py wrappers)."""
    import flowdiffusion.flowdiffusion.unet as U
    from oracle.param_fill import fill_module
    from oracle.video_unet import UNetCfg, unet_forward
    from tools_wsum import wsum
    g = np.load(f"{golden_dir}/wrappers.npz")
    for cls_name, ci, res, frames in [("UnetThor", 3, 16, 2), ("UnetMWFlow", 2, 32, 2), ("UnetBridge", 3, 16, 2), ("UnetMW", 3, 32, 2)]:
        torch.manual_seed(0)
        m = getattr(U, cls_name)()
        sd = fill_module(m, seed=21)
        assert abs(wsum(sd) - float(g[f"{cls_name}_wsum"])) < 1e-9 * float(g[f"{cls_name}_wsum"]), cls_name
        cm = m.unet
        cfg = UNetCfg(in_channels=cm.in_channels, model_channels=cm.model_channels, out_channels=cm.out_channels,
                      num_res_blocks=cm.num_res_blocks, attention_resolutions=cm.attention_resolutions, channel_mult=cm.channel_mult,
                      num_head_channels=32)
        gen = torch.Generator().manual_seed(5)
        x = torch.randn(1, frames * ci + 3, res, res, generator=gen)
        t = torch.tensor([17])
        te = torch.randn(1, 4, 512, generator=gen)
        cond = x[:, -3:, None].expand(1, 3, frames, res, res)
        xx = x[:, :-3].reshape(1, frames, ci, res, res).permute(0, 2, 1, 3, 4)
        with torch.no_grad():
            yo = unet_forward(sd, torch.cat([xx, cond], 1), t, te, cfg, pre="unet.")
        yo = yo.permute(0, 2, 1, 3, 4).reshape(1, frames * cm.out_channels, res, res)
        assert rel(yo, g[f"{cls_name}_y"]) < 1e-5, (cls_name, rel(yo, g[f"{cls_name}_y"]))


def test_oracle_video_p_losses_vs_reference(golden_dir):
    """oracle p_losses + autograd through the oracle UNet == the reference's GoalGaussianDiffusion.forward + backward."""
    from oracle.video_unet import UNetCfg, unet_libero_forward
    from oracle import goal_diffusion as GD
    g = np.load(f"{golden_dir}/video_train.npz", allow_pickle=True)
    m, sd = _tiny_sd()
    names = [str(n) for n in g["param_names"]]
    assert names == [n for n, _ in m.named_parameters()]
    cfg = UNetCfg(in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
                  num_head_channels=16)
    img, cond, te = (torch.from_numpy(g[k]) for k in ("img", "cond", "te"))
    for tag, lt, obj in (("l2_v", "l2", "pred_v"), ("l1_noise", "l1", "pred_noise")):
        P = {k: (v.clone().requires_grad_(True) if torch.is_floating_point(v) else v) for k, v in sd.items()}
        T = GD.cosine_tables(100, objective=obj)
        fn = lambda xx, tt, ee: unet_libero_forward(P, xx, tt, ee, cfg)
        loss = GD.p_losses(fn, T, img * 2 - 1, torch.from_numpy(g[f"{tag}_t"]), cond, te, torch.from_numpy(g[f"{tag}_noise"]), obj, lt)
        assert abs(loss.item() - float(g[f"{tag}_loss"])) < 2e-6 * max(1.0, abs(float(g[f"{tag}_loss"])))
        loss.backward()
        gn = np.array([float(P[n].grad.double().norm()) for n in names])
        assert np.max(np.abs(gn - g[f"{tag}_grad_norms"]) / (g[f"{tag}_grad_norms"] + 1e-6 * g[f"{tag}_grad_norms"].max())) < 2e-4, tag


TRANSFORMER_CFGS = {
    "dec_causal": dict(input_dim=7, output_dim=7, horizon=16, n_obs_steps=3, cond_dim=64, n_layer=2, n_head=4, n_emb=64, p_drop_emb=0.0,
                       p_drop_attn=0.0, causal_attn=True, time_as_cond=True, obs_as_cond=True, n_cond_layers=2),
    "dec_mlp": dict(input_dim=4, output_dim=4, horizon=10, n_obs_steps=2, cond_dim=32, n_layer=1, n_head=2, n_emb=96, p_drop_emb=0.0,
                    p_drop_attn=0.0, causal_attn=False, time_as_cond=True, obs_as_cond=True, n_cond_layers=0),
    "bert_causal": dict(input_dim=7, output_dim=7, horizon=12, n_layer=2, n_head=4, n_emb=64, p_drop_emb=0.0, p_drop_attn=0.0,
                        causal_attn=True, time_as_cond=False),
}


@pytest.mark.parametrize("tag", list(TRANSFORMER_CFGS))
def test_oracle_transformer_vs_reference(golden_dir, tag):
    """Shell (names, buffers, optimiser groups) and oracle forward / autograd of TransformerForDiffusion vs the reference class."""
    from flowdiffusion.flowdiffusion.diffusion_policy_baseline.transformer_for_diffusion import TransformerForDiffusion
    from oracle import transformer as OT
    from oracle.param_fill import fill_module
    from tools_wsum import wsum
    g = np.load(f"{golden_dir}/transformer.npz", allow_pickle=True)
    cfg = TRANSFORMER_CFGS[tag]
    torch.manual_seed(0)
    m = TransformerForDiffusion(**cfg)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g[f"{tag}_names"]]
    assert list(m.state_dict().keys()) == [str(n) for n in g[f"{tag}_state_keys"]]
    for n, p in m.named_parameters():                       # initialisation rule: unit LayerNorm, zero biases, N(0, 0.02) matrices
        if "norm" in n or n.startswith("ln_f"):
            assert torch.equal(p, torch.ones_like(p) if n.endswith("weight") else torch.zeros_like(p)), n
        elif p.dim() == 1:
            assert float(p.abs().max()) == 0.0, n
        elif p.dim() == 2:
            assert 0.015 < float(p.detach().std()) < 0.025, n
    groups = m.get_optim_groups(1e-3)
    assert sum(len(gr["params"]) for gr in groups) == len(list(m.parameters())) and all(p.dim() == 2 for p in groups[0]["params"])
    sd = fill_module(m, seed=21)
    assert abs(wsum({k: v for k, v in sd.items() if "mask" not in k}) - float(g[f"{tag}_wsum"])) < 1e-9 * float(g[f"{tag}_wsum"])
    P = {k: (v.clone().requires_grad_(True) if k in dict(m.named_parameters()) else v) for k, v in sd.items()}
    x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    cond = torch.from_numpy(g[f"{tag}_cond"]).requires_grad_(True) if f"{tag}_cond" in g else None
    y = OT.forward(P, x, torch.from_numpy(g[f"{tag}_t"]), cond, cfg["n_head"], cfg["n_layer"], cfg.get("n_cond_layers", 0), m.encoder_only)
    assert rel(y, g[f"{tag}_y"]) < 1e-5
    (y * torch.from_numpy(g[f"{tag}_R"])).sum().backward()
    assert rel(x.grad, g[f"{tag}_dx"]) < 1e-4
    if cond is not None:
        assert rel(cond.grad, g[f"{tag}_dcond"]) < 1e-4
    names = [str(n) for n in g[f"{tag}_names"]]
    gn = np.array([float(P[n].grad.double().norm()) for n in names])
    assert np.max(np.abs(gn - g[f"{tag}_grad_norms"]) / (g[f"{tag}_grad_norms"] + 1e-6 * g[f"{tag}_grad_norms"].max())) < 1e-4
    with pytest.raises(RuntimeError):                        # the product module has no CPU path
        m.eval()(x.detach(), torch.from_numpy(g[f"{tag}_t"]), None if cond is None else cond.detach())


def test_oracle_policy_vs_reference(golden_dir):
    from oracle import policy as OP
    from oracle.param_fill import fill_module
    from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
    from tools_wsum import wsum
    g = np.load(f"{golden_dir}/policy.npz", allow_pickle=True)
    torch.manual_seed(0)
    pol = build_policy(DEFAULT_CONF)
    sd = fill_module(pol, seed=13)
    assert list(pol.state_dict().keys()) == [str(k) for k in g["state_keys"]]           # checkpoint key layout incl. aliases
    assert abs(wsum(sd) - float(g["weights_abs_sum"])) < 1e-9 * float(g["weights_abs_sum"])
    batch = {"obs": {"img_obs_1": torch.from_numpy(g["img_obs"]), "img_goal_1": torch.from_numpy(g["img_goal"])},
             "action": torch.from_numpy(g["action"])}
    names = [str(n) for n in g["param_names"]]
    loss, grads = OP.loss_and_grads(sd, batch, torch.from_numpy(g["noise"]), torch.from_numpy(g["timesteps"]), names=names)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    gn = np.array([float(grads[n].double().norm()) for n in names])
    assert np.max(np.abs(gn - g["grad_norms"]) / (g["grad_norms"] + 1e-6)) < 1e-4
    torch.manual_seed(70)
    o = OP.predict_action(sd, batch["obs"], torch.randn(2, 16, 7), [], use_ddim=True)
    assert rel(o["action_pred"], g["ddim_action_pred"]) < 1e-6 and rel(o["action"], g["ddim_action"]) < 1e-6


def test_oracle_decision_routing_is_neutral_and_the_tie_check_bites(golden_dir):
    """The flip-aware machinery of the GPU gradient tests (oracle.policy `dec`, tests/flip_aware.py), checked without a GPU: recording the
    encoders' ReLU / max-pool decisions changes nothing; routing forward and backward through the RECORDED decisions reproduces the free
    run's loss and gradients (so a comparison through another forward's decisions only differs where that forward decided differently);
    a decision flipped on a genuine tie passes check_routing, one flipped on a clear pre-activation does not."""
    from oracle import policy as OP
    from oracle.param_fill import fill_module
    from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
    from flip_aware import check_routing
    g = np.load(f"{golden_dir}/policy.npz", allow_pickle=True)
    torch.manual_seed(0)
    pol = build_policy(DEFAULT_CONF)
    sd = fill_module(pol, seed=13)
    batch = {"obs": {"img_obs_1": torch.from_numpy(g["img_obs"]), "img_goal_1": torch.from_numpy(g["img_goal"])},
             "action": torch.from_numpy(g["action"])}
    noise, ts = torch.from_numpy(g["noise"]), torch.from_numpy(g["timesteps"])
    names = [str(n) for n in g["param_names"]]
    loss0, g0 = OP.loss_and_grads(sd, batch, noise, ts, names=names)
    rec = {"record": {}}
    loss1, g1 = OP.loss_and_grads(sd, batch, noise, ts, names=names, dec=rec)
    assert loss1.item() == loss0.item() and all(torch.equal(g0[n], g1[n]) for n in names)
    own = {k: v for k, v in rec["record"].items() if not k.endswith(":z")}
    assert len(own) == 2 * (2 + 2 * 8) and sum(v.numel() for v in own.values()) > 2_000_000
    loss2, g2 = OP.loss_and_grads(sd, batch, noise, ts, names=names, dec={"use": own})
    assert abs(loss2.item() - loss0.item()) <= 1e-7 * abs(loss0.item())
    gsc = max(float(g0[n].norm()) for n in names)
    assert max(float((g2[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-3 * gsc) for n in names) <= 1e-6
    assert check_routing(rec["record"], own)[0] == 0
    # one flipped ReLU decision: on the pre-activation closest to zero (a tie: accepted), then on the largest one (rejected)
    key = next(k for k in own if k.endswith(".4.0.relu1"))
    z = rec["record"][key + ":z"]
    for pick, ok in ((z.abs().argmin(), z.abs().min() <= 3e-5 * z.abs().max()), (z.abs().argmax(), False)):
        flipped = dict(own)
        m = own[key].clone()
        m.view(-1)[pick] = ~m.view(-1)[pick]
        flipped[key] = m
        if ok:
            assert check_routing(rec["record"], flipped)[0] == 1
        else:
            with pytest.raises(AssertionError):
                check_routing(rec["record"], flipped)
    # one max-pool winner (an interior window: all nine taps lie inside the map) moved to the smallest element of its window: rejected
    key = next(k for k in own if k.endswith(".pool"))
    zz = rec["record"][key + ":z"]
    win = zz[0, 0, 2 * 5 - 1:2 * 5 + 2, 2 * 5 - 1:2 * 5 + 2].reshape(-1)
    assert float(win.max() - win.min()) > 1e-3 * float(zz.abs().max())
    t = own[key].clone()
    t[0, 0, 5, 5] = int(win.argmin())
    with pytest.raises(AssertionError):
        check_routing(rec["record"], dict(own, **{key: t}))


def test_oracle_policy_action_limits_vs_reference(golden_dir):
    """Non-identity action limits (lb_action_minmax_orn01, diffuser/datasets/__init__.py:30-37): the oracle's normalise /
    unnormalise against the reference run with that normaliser; the product shell accepts the limits and rejects what the HIP
    loaders do not implement (wrong limits must fail, not mis-scale)."""
    import copy
    import dataclasses
    from oracle import policy as OP
    from oracle.param_fill import fill_module
    from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF, _RESOLVERS
    g = np.load(f"{golden_dir}/policy_orn01.npz")
    p0 = np.load(f"{golden_dir}/policy.npz", allow_pickle=True)
    conf = copy.deepcopy(DEFAULT_CONF)
    img = {"shape": [3, 128, 128], "minmax_shape": _RESOLVERS["image_minmax_01"](), "type": "rgb"}
    conf["shape_meta"] = {"obs": {"img_obs_1": dict(img), "img_goal_1": dict(img)},
                          "action": {"shape": [7], "minmax_shape": _RESOLVERS["lb_action_minmax_orn01"]()}}
    torch.manual_seed(0)
    pol = build_policy(conf)
    sd = fill_module(pol, seed=13)
    assert np.allclose(pol._cfg.act_limits[1], g["act_max"]) and np.allclose(pol._cfg.act_limits[0], g["act_min"])
    cfg = dataclasses.replace(OP.LIBERO_POLICY, act_min=tuple(g["act_min"].tolist()), act_max=tuple(g["act_max"].tolist()))
    batch = {"obs": {"img_obs_1": torch.from_numpy(p0["img_obs"]), "img_goal_1": torch.from_numpy(p0["img_goal"])},
             "action": torch.from_numpy(g["action"])}
    loss = OP.compute_loss(sd, batch, torch.from_numpy(g["noise"]), torch.from_numpy(g["timesteps"]), cfg)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    torch.manual_seed(70)
    o = OP.predict_action(sd, batch["obs"], torch.randn(2, 16, 7), [], cfg, use_ddim=True)
    assert rel(o["action_pred"], g["ddim_action_pred"]) < 1e-6
    # what the fused loaders do not implement is refused at construction
    bad = copy.deepcopy(conf)
    bad["shape_meta"]["obs"]["img_obs_1"]["minmax_shape"] = (np.zeros(3, np.float32), np.full(3, 255, np.float32), [1, 3, 1, 1])
    with pytest.raises(NotImplementedError):
        build_policy(bad)
    from diffuser.diffusion_policy import schedulers as S
    sch = S.DDPMScheduler(num_train_timesteps=100, beta_schedule="squaredcos_cap_v2", clip_sample=True, prediction_type="epsilon")
    sch.alphas_cumprod = torch.linspace(0.999, 0.01, 100)                       # e.g. a linear-beta scheduler object
    with pytest.raises(NotImplementedError):
        type(pol)._check_fused_constants(conf["shape_meta"], sch, None)


def test_oracle_optimiser_vs_torch():
    """oracle/optim.py against torch.optim.AdamW + clip_grad_norm_ (both present here) and the ema decay formula."""
    from oracle import optim as O
    torch.manual_seed(0)
    ps = [torch.randn(5, 7), torch.randn(11), torch.randn(3, 3, 3)]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.AdamW(ref, lr=1e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6)
    ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    em = [p.clone() for p in ps]
    st = O.EmaState(power=0.75)
    for step in range(1, 6):
        gs = [torch.randn_like(p) * 3 for p in ps]
        for r, g in zip(ref, gs):
            r.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt.step(); opt.zero_grad()
        tn2 = O.train_tail(ps, [g.clone() for g in gs], ms, vs, em, step, st)
        assert abs(float(tn - tn2)) < 1e-6
        assert max((a - b).abs().max().item() for a, b in zip(ps, ref)) == 0.0
    # decay(step) = clamp(1 - (1 + step)^-0.75): in-tree statement diffuser/diffusion_policy/model/ema_model.py:44-54
    assert O.ema_decay(1) == 0.0 and abs(O.ema_decay(2) - (1 - 2 ** -0.75)) < 1e-12 and O.ema_decay(10 ** 9) == 0.9999


def test_policy_schedulers_against_published_closed_forms():
    """R7 (diffusers DDPMScheduler / DDIMScheduler, absent third party): the restatement is checked against answers that follow from
    the PUBLISHED definitions by a different route than the restated loops (no network here: the sources are cited, the numbers
    are derived in fp64 below, independently of oracle/schedulers.py's code path):
      * squaredcos_cap_v2 = Nichol & Dhariwal 2021 (arXiv:2102.09672) eq. 17: alpha_bar(t) = f(t)/f(0), f(t) = cos^2((t/T + s)/(1 + s) pi/2),
        s = 0.008, beta_t = min(1 - alpha_bar(t)/alpha_bar(t-1), 0.999).  While the cap is inactive the cumulative product telescopes:
        alphas_cumprod[i] = f((i+1)/T)/f(0) in closed form; f(1) = 0, so only the LAST beta is capped: alphas_cumprod[T-1] =
        alphas_cumprod[T-2] * 0.001.
      * "leading" timestep spacing (Lin et al. 2023, arXiv:2305.08891 table 2; diffusers' default): t_i = i * floor(T / n), descending:
        T = 100, n = 8 -> 84, 72, ..., 0.
      * one ancestral step, Ho et al. 2020 (arXiv:2006.11239): the posterior mean of eq. 7 with x0 from eq. 15 equals eq. 11,
        mu = (x_t - beta_t / sqrt(1 - alpha_bar_t) eps) / sqrt(alpha_t), whenever the x0 clip is inactive; "fixed_small" variance =
        beta_t (1 - alpha_bar_{t-1}) / (1 - alpha_bar_t) (eq. 7's beta-tilde).
      * one DDIM step, Song et al. 2021 (arXiv:2010.02502) eq. 12 with sigma = 0."""
    import math
    from oracle import schedulers as S
    T = 100
    f = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    ac = S.squaredcos_alphas_cumprod(T).double().numpy()
    closed = np.array([f((i + 1) / T) / f(0.0) for i in range(T)])
    assert np.abs(ac[:T - 1] / closed[:T - 1] - 1).max() < 2e-5          # fp32 cumulative product of 99 factors vs the closed form
    assert abs(ac[T - 1] / (ac[T - 2] * 0.001) - 1) < 2e-5               # the 0.999 cap on the last beta (1 - fp32(0.999) = 1.0000129e-3)
    betas = S.squaredcos_betas(T).double().numpy()
    assert abs(betas[T - 1] - 0.999) < 1e-7 and (betas[:T - 1] < 0.999).all() and (np.diff(betas[:T - 1]) > 0).all()
    # three anchor values from the closed form (fp64), to 5 significant digits
    for i, v in ((0, f(0.01) / f(0.0)), (49, f(0.5) / f(0.0)), (98, f(0.99) / f(0.0))):
        assert abs(ac[i] / v - 1) < 2e-5, (i, ac[i], v)
    assert S.ddim_timesteps(100, 8) == [84, 72, 60, 48, 36, 24, 12, 0]
    assert S.ddim_timesteps(100, 10) == list(range(90, -1, -10))
    g = torch.Generator().manual_seed(0)
    act = S.squaredcos_alphas_cumprod(T)
    x_t = torch.randn(4, 16, 7, generator=g) * 0.12
    eps = torch.randn(4, 16, 7, generator=g) * 0.12
    noise = torch.randn(4, 16, 7, generator=g)
    for t in (7, 50):
        a_bar, a_prev = float(act[t]), float(act[t - 1])
        alpha_t = a_bar / a_prev
        beta_t = 1 - alpha_t
        x0 = (x_t.double() - math.sqrt(1 - a_bar) * eps.double()) / math.sqrt(a_bar)
        assert float(x0.abs().max()) < 1.0                                 # clip inactive: eq. 11 applies
        mu = (x_t.double() - beta_t / math.sqrt(1 - a_bar) * eps.double()) / math.sqrt(alpha_t)
        sigma = math.sqrt(beta_t * (1 - a_prev) / (1 - a_bar))
        got = S.ddpm_step(act, eps, t, x_t, noise, T)
        assert rel(got, mu + sigma * noise.double()) < 2e-5
    # DDIM (eta = 0): x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps, previous step 12 below; the final step lands on alpha_bar = 1
    for t in (84, 12, 0):
        a_bar = float(act[t])
        a_prev = float(act[t - 12]) if t >= 12 else 1.0
        x0 = ((x_t.double() - math.sqrt(1 - a_bar) * eps.double()) / math.sqrt(a_bar)).clamp(-1, 1)
        want = math.sqrt(a_prev) * x0 + math.sqrt(1 - a_prev) * eps.double()
        assert rel(S.ddim_step(act, eps, t, x_t, T, 8), want) < 2e-5
