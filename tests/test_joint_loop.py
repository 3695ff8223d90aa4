"""The joint loop around the hot path (SURVEY.md 8b row `diffuser.libero.*`, 8f ranks 1-2): schedule state machines against a trace
recorded from the reference class, the random-action episode readers, the synthetic environment protocol, and -- on the GPU --
a short end-to-end run: fill buffer -> train -> video-guided rollout -> mixed-buffer train -> save / load -> evaluation."""
import os
import random
import sys
import numpy as np
import pytest
import torch

from tools_schedule import _schedule_trace, schedule_stub, SCHEDULE_CFGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_state_machines_match_reference_trace(golden_dir):
    from diffuser.libero.lb_online_trainer_v7 import LB_Online_Trainer_V7 as Ours
    assert "/root/reference" not in sys.modules[Ours.__module__].__file__
    z = np.load(os.path.join(golden_dir, "schedule.npz"))
    for name, cfg in SCHEDULE_CFGS.items():
        n = int(z[name + "_n"])
        got = _schedule_trace(Ours, lambda: schedule_stub(cfg), n)
        want = np.unpackbits(z[name], axis=0)[:n]
        assert np.array_equal(got, want), name
        assert got[:, 0].any() or name == "released" and n <= 10000      # the vid-bias phase is actually reached


def test_host_utils_match_reference_semantics():
    from diffuser.libero import _host_utils as U
    assert U.number_by_ratio(64, [0.75, 0.25]) == [48, 16]           # config/libero/lb_tk8_65to72.py:100-101
    assert U.number_by_ratio(64, [0.25, 0.75]) == [16, 48]
    assert U.number_by_ratio(10, [0.2, 0.8]) == [2, 8]
    imgs = np.random.RandomState(0).randint(0, 256, (2, 8, 8, 3)).astype(np.uint8)
    t = U.imgs_preproc_simple_noCrop_v1(imgs)
    assert t.shape == (2, 3, 8, 8) and t.dtype == torch.float32
    assert torch.equal(t, torch.from_numpy(imgs).permute(0, 3, 1, 2).float() / 255.0)


def test_synthetic_env_protocol_and_determinism():
    from diffuser.libero.lb_synthetic_env import SyntheticLiberoEnvList
    e = SyntheticLiberoEnvList()
    assert e.num_tasks == 8 and e.camera_list == ['agent'] and e.num_seed_per_task == 1
    tk = e.task_list[3]
    e.check_no_envs_exist()
    env = e.init_1_given_env(tk, 0, e_seed=7)
    a = e.render_an_env(tk, 'agent', 0)
    assert a.shape == (128, 128, 3) and a.dtype == np.uint8
    for _ in range(10):
        _, _, done, _ = e.step_an_env(tk, 0, np.array([0.3, -0.2, -0.5, 0, 0, 0, 1.0]))
    b = e.render_a_given_env(env, 'agent')
    assert (a != b).any()
    rgb, depth = e.render_an_env_with_depth(tk, 'gripper', 0)
    assert depth.shape == (128, 128, 1) and (depth >= 0).all()
    assert e.get_an_env_obs(tk, 0)['robot0_eef_pos'].shape == (3,)
    e.close_1_given_env(tk, 0)
    e.check_no_envs_exist()
    env2 = e.init_1_given_env(tk, 0, e_seed=7)
    assert np.array_equal(e.render_a_given_env(env2, 'agent'), a)
    e.close_1_given_env(tk, 0)


def test_randsam_readers_roundtrip(tmp_path):
    from diffuser.libero.lb_synthetic_env import SyntheticLiberoEnvList
    from diffuser.libero.lb_randsam_io import RandSamSynthetic, RandSamNpzDir, open_randsam
    e = SyntheticLiberoEnvList(image_hw=(32, 32))
    src = RandSamSynthetic(e, episodes_per_task=3, ep_len=(20, 25))
    tk = e.task_list[0]
    imgs, acts = src.episode(tk, 1)
    assert imgs.dtype == np.uint8 and imgs.shape[1:] == (32, 32, 3) and len(imgs) == len(acts) + 1 and acts.shape[1] == 7
    assert np.abs(acts[:, 3:6]).max() <= 0.1 and np.abs(acts).max() <= 1.0
    imgs2, acts2 = src.episode(tk, 1)
    assert np.array_equal(imgs, imgs2) and np.array_equal(acts, acts2)
    for i in range(3):
        RandSamNpzDir.write_episode(str(tmp_path), tk, i, *src.episode(tk, i))
    rd = open_randsam(str(tmp_path))
    assert rd.num_episodes(tk) == 3 and rd.has(tk, 2) and not rd.has(tk, 3)
    i3, a3 = rd.episode(tk, 1)
    assert np.array_equal(i3, imgs) and np.array_equal(a3, acts)
    with pytest.raises((ImportError, FileNotFoundError)):
        open_randsam(str(tmp_path / "missing.hdf5"))


def test_text_tower_fallback_shapes():
    os.environ["V2A_TEXT_ENCODER"] = "hash"
    try:
        from diffuser.libero.lb_train_utils import build_text_tower
        tok, enc = build_text_tower()
        ids = tok(["put the red mug on the plate", "open drawer"], return_tensors='pt', padding=True, truncation=True, max_length=128)
        h = enc(**ids).last_hidden_state
        assert h.shape == (2, ids["input_ids"].shape[1], 512)
        h2 = enc(**tok(["put the red mug on the plate", "open drawer"])).last_hidden_state
        assert torch.equal(h, h2)
    finally:
        os.environ.pop("V2A_TEXT_ENCODER")


# ----------------------------------------------------------------------------------------------------------------- GPU
def _small_trainer_dict(**over):
    from diffuser.libero.lb_constants import LB_GRASP_actdown_value_range_1
    td = dict(num_init_rand_Ep_per_tk=2, envBuf_max_num_uB_rand=64, envBuf_max_num_uB_vid=32, max_len_uB=700, min_len_uB=30,
              allow_small_buffers=True, pool_frames_rand=64 * 150, pool_frames_vid=32 * 400, is_stop_at_suc=False,
              model_act_horizon=16, init_rand_steps=2, rand_cycle_steps=1, vid_cycle_steps=4, use_env_rand_reset=True,
              video_explo_freq=3, rand_explo_freq=4, rand_explo_num_Ep_per_tk=1, n_acts_per_pred=8, n_preds_betw_vframes=(1, 2),
              batch_size=4, batch_size_v=1, buf_sample_batch_size=8, buf_sample_ratio_rand=[0.75, 0.25],
              buf_sample_ratio_vid=[0.25, 0.75], buf_sample_method='rand_prob', buf_sample_randBuf_prob=0.3, enable_noExp=False,
              n_acts_down_range=(4, 4), n_acts_close_grp=2, act_down_val=None,
              act_down_val_range_per_tk=LB_GRASP_actdown_value_range_1, close_grp_force=0.98, close_grp_act_down_val=0,
              rand_explo_type='from_h5', randsam_filename='synthetic:6', grasp_z_diff_limit=0.36, grasp_abs_z_limit=0.56, seed=0)
    td.update(over)
    return td


def _build_trainer(tmp_path, tasks=2, **over):
    os.environ["V2A_TEXT_ENCODER"] = "hash"
    from types import SimpleNamespace
    from diffuser.diffusion_policy import Init_Diffusion_Policy
    from diffuser.libero._host_utils import LB_ACTION_MIN, LB_ACTION_MAX
    from diffuser.libero.lb_online_dataset import LB_Online_Dataset
    from diffuser.libero.lb_synthetic_env import SyntheticLiberoEnvList, LB_TASKS_65to72
    from diffuser.libero.lb_train_utils import LB_Init_Trainer
    from diffuser.libero.lb_video_model_utils import lb_get_video_model_gcp_v2
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    td = _small_trainer_dict(**over)
    env_list = SyntheticLiberoEnvList(task_list=list(LB_TASKS_65to72)[:tasks])
    env_list.task_to_task_idx = {tk: LB_TASKS_65to72[tk] for tk in env_list.task_list}
    ds = LB_Online_Dataset(env_list, target_size=(128, 128), dataset_config=dict(act_min_max=(LB_ACTION_MIN, LB_ACTION_MAX), combo_type='all'))
    args = SimpleNamespace(trainer_type='v7', trainer_dict=td, input_img_size=(128, 128), model_yl_path=None)
    init_tr = LB_Init_Trainer(args)
    video = lb_get_video_model_gcp_v2(ckpts_dir=str(tmp_path / "no_ckpt"), milestone=1, timestep=2, g_w=0, sample_per_seq=8,
                                      target_size=(128, 128), allow_random_init=True,
                                      text_tower=(init_tr.tokenizer, init_tr.text_encoder))
    dp = Init_Diffusion_Policy(args)
    tr = init_tr.trainer_cls(init_diff_policy=dp, video_model=video, tokenizer=init_tr.tokenizer, text_encoder=init_tr.text_encoder,
                             train_set=ds, valid_set=ds, channels=3, train_batch_size=td['batch_size'], video_batch_size=1,
                             train_num_steps=8, opt_params=dict(lr=1e-4, betas=[0.95, 0.999], eps=1e-8, weight_decay=1e-6),
                             ema_params=dict(update_after_step=0, inv_gamma=1.0, power=0.75, min_value=0.0, update_every=1,
                                             include_online_model=False),
                             render_img_size=(128, 128), input_img_size=(128, 128), save_freq=4, label_freq=4, log_freq=2,
                             sample_freq=0, results_folder=str(tmp_path / "run"), trainer_dict=td)
    return tr, ds, dp, video


@pytest.mark.gpu
def test_batched_exploration_round_equals_the_one_by_one_rows(tmp_path):
    """LB_Online_Trainer_V7.video_guided_explore samples the goal frames of ALL of a round's combinations in ONE
    batched sampler call (one Philox seed per row).  The same round with the rows sampled through bs-1 calls (same token rows, same
    seeds: trainer_dict['_explore_rows_one_by_one']) must give the same initial noise bit for bit, the same frames up to the kernels'
    batch-size-dependent summation order, and the same number of episodes; the reference's one-at-a-time order (explore_batched=False)
    still runs."""
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    vids, eps = {}, {}
    for mode in ("batched", "rows", "reference"):
        over = {"batched": {}, "rows": {"_explore_rows_one_by_one": True}, "reference": {"explore_batched": False}}[mode]
        tr, ds, dp, video = _build_trainer(tmp_path / mode, tasks=3, **over)
        torch.manual_seed(11); torch.cuda.manual_seed(11); np.random.seed(11); random.seed(11)
        tr.video_guided_explore()
        assert len(tr.envBuf_vid) == 3 and tr.cnt_vid_rollouts == 3
        vids[mode] = getattr(tr, "_last_explore_videos", None)
        eps[mode] = tr.envBuf_vid.episode_lengths().copy()
        del tr, ds, dp, video
        torch.cuda.empty_cache()
    a, b = vids["batched"], vids["rows"]
    assert a.shape == b.shape == (3, 7, 3, 128, 128) and vids["reference"] is None
    d = float((a - b).abs().max())
    print(f"[batched exploration] frames of the B = 3 call vs three bs-1 calls: max |diff| {d:.2e}")
    assert d <= 1e-4, d
    assert float((a[0] - a[1]).abs().max()) > 1e-2                     # (the rows do differ from each other)
    # one Philox seed per row: a batched draw reproduces the one-row draws bit for bit
    from v2a_hip import ops
    seeds = torch.tensor([5, 6, 7], dtype=torch.int64, device="cuda:0")
    z3 = ops.philox_normal_rows(torch.empty(3, 21, 128, 128, device="cuda:0"), seeds)
    for j in range(3):
        z1 = ops.philox_normal_rows(torch.empty(1, 21, 128, 128, device="cuda:0"), seeds[j:j + 1])
        assert torch.equal(z1[0], z3[j]) and torch.equal(ops.philox_normal(torch.empty(21, 128, 128, device="cuda:0"), int(seeds[j])), z3[j])


@pytest.mark.gpu
def test_joint_loop_train_rollout_checkpoint_eval(tmp_path):
    tr, ds, dp, video = _build_trainer(tmp_path)
    # the entry script's smoke test (scripts/train_libero_dp.py:126-135): autograd loss through the plugin + opt.zero_grad
    img1, img2, _, act = ds.sample_random_tensor(1, 16, tr.device)
    loss = dp.diffusion_policy.compute_loss(tr.to_batch_dict(img1, img2, act))
    loss.backward()
    tr.opt.zero_grad()
    assert all(p.grad is None for p in dp.diffusion_policy.parameters())

    tr.train()
    assert tr.step == 8
    assert len(tr.envBuf_rand) >= 4 and len(tr.envBuf_vid) >= 2          # rollouts at steps 3 and 6 (2 tasks each)
    assert tr.cnt_vid_rollouts == len(tr.envBuf_vid) and tr.num_steps_in_env > 0
    lens = tr.envBuf_vid.episode_lengths()
    assert ((lens - 1) % 8 == 0).sum() + ((lens - 1 - 6) % 8 == 0).sum() >= len(lens)   # n*8 actions (+4 down +2 grasp)
    assert torch.isfinite(tr.ptrainer.loss).all()
    metrics = [l for l in open(tmp_path / "run" / "metrics.jsonl")]
    assert len(metrics) >= 4

    # mixed minibatch: rand rows first, then rollout rows, straight from the shared HBM pool
    o0, o1, oa, tasks, info = tr.sample_from_bufs()
    assert o0.shape == (8, 3, 128, 128) and oa.shape == (8, 16, 7) and len(tasks) == 8 and o0.is_cuda
    assert float(o0.min()) >= 0 and float(o0.max()) <= 1

    # checkpoint keys (lb_online_trainer_v7.py:371-382) and an exact save -> load round trip of optimiser / EMA state
    ck = torch.load(tmp_path / "run" / "model-8.pt", map_location="cpu", weights_only=False)
    assert set(ck) >= {'step', 'num_steps_in_env', 'gcp_model', 'opt', 'ema', 'scaler', 'version', 'cnt_vid_rollouts', 'cnt_vid_rout_per_tk'}
    assert ck['step'] == 8 and len(ck['opt']['state']) == len(tr.ptrainer.names) == 276
    assert len(ck['opt']['param_groups'][0]['params']) == len(list(dp.diffusion_policy.parameters()))
    k0 = min(ck['opt']['state'])
    assert float(ck['opt']['state'][k0]['step']) == 8.0 and int(ck['ema']['step']) == 8
    ref_opt = torch.optim.AdamW(dp.diffusion_policy.parameters(), lr=1e-4)
    ref_opt.load_state_dict(ck['opt'])                                      # torch's own AdamW accepts the layout
    m_before, v_before = tr.ptrainer.opt.m.clone(), tr.ptrainer.opt.v.clone()
    ema_before = {k: v.clone() for k, v in tr.ptrainer.ema_policy.state_dict().items()}
    tr.ptrainer.opt.m.zero_(); tr.ptrainer.opt.v.zero_()
    tr.load(8)
    assert torch.equal(tr.ptrainer.opt.m, m_before) and torch.equal(tr.ptrainer.opt.v, v_before)
    assert tr.ptrainer.opt.counters() == (8, 8, True)
    for k, v in tr.ptrainer.ema_policy.state_dict().items():
        assert torch.equal(v, ema_before[k]), k
    tr.train_num_steps = 10
    tr.train()                                                              # resumes: two more steps on the restored state
    assert tr.step == 10 and tr.ptrainer.opt.counters()[0] == 10

    # evaluation harness on top (lb_eval_helper.py): 1 task x 1 seed, one video prediction, 1 policy call per frame
    from diffuser.libero.lb_eval_helper import LB_DP_Eval
    ev = LB_DP_Eval(gcp_model=None, ema=tr.ema, video_model=video, trainer=tr, env_list=ds.env_list, task_list=ds.task_list[:1],
                    cam_list=ds.cam_list, valid_seeds=[100], max_episode_steps=500, render_img_size=tr.render_img_size,
                    rendered_imgs_preproc_fn=tr.rendered_imgs_preproc_fn, is_video_ddim=True, is_dp_ddim=True,
                    eval_n_preds_betw_vframes=1, save_path=str(tmp_path), num_vid_pred_per_ep=1, use_vid_first_n_frames=2,
                    device=tr.device)
    res = ev.run_evals()
    assert res['num_evals'] == 1 and 0.0 <= res['suc_rate'] <= 1.0 and set(res['suc_rate_per_tk']) == set(ds.task_list[:1])


@pytest.mark.gpu
def test_mixed_pool_gather_matches_per_store_gather(tmp_path):
    """PolicyTrainer's one-launch gather over the shared pool == the reference's two per-buffer samples concatenated."""
    from v2a_hip.replay import ReplayStore, sample_mixed
    from v2a_hip._lib import lib, check
    from v2a_hip import ops
    dev = "cuda:0"
    a, b = ReplayStore.pair(16, 8, 700, 30, capacity_a=16 * 60, capacity_b=8 * 90, device=dev)
    rng = np.random.RandomState(3)
    for i in range(10):
        n = int(rng.randint(40, 60))
        a.add_one_episode("t", "agent", 0, torch.from_numpy(rng.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)),
                          torch.from_numpy(rng.uniform(-1, 1, (n - 1, 7)).astype(np.float32)))
    for i in range(5):
        n = int(rng.randint(60, 90))
        b.add_one_episode("t", "agent", 0, torch.from_numpy(rng.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)),
                          torch.from_numpy(rng.uniform(-1, 1, (n - 1, 7)).astype(np.float32)))
    np.random.seed(11); random.seed(11)
    o0, o1, oa, _, info = sample_mixed(a, b, 16, 0.3)
    assert len(info["episodes"]) == 16
    # same draw again through pool-absolute rows and ONE gather on the root tensors
    np.random.seed(11); random.seed(11)
    from v2a_hip.replay import count_uniform_below, sample_indices
    nr = count_uniform_below(16, 0.3)
    e0, s0 = sample_indices(a.episode_lengths(), nr, 16)
    e1, s1 = sample_indices(b.episode_lengths(), 16 - nr, 16)
    rows = torch.from_numpy(np.concatenate([a.pool_rows(e0, s0), b.pool_rows(e1, s1)])).to(dev)
    p0, p1, pa = torch.empty_like(o0), torch.empty_like(o1), torch.empty_like(oa)
    check(lib.v2a_replay_gather(a.root_frames.data_ptr(), 1, a.root_acts.data_ptr(), rows.data_ptr(), p0.data_ptr(), p1.data_ptr(),
                                pa.data_ptr(), 16, 128, 128, 16, 7, 0, 1, ops._stream()), "replay_gather")
    assert torch.equal(p0, o0) and torch.equal(p1, o1) and torch.equal(pa, oa)
