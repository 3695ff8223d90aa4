"""Hyper-parameters of the released Libero 8-task run, grouped by what they steer (values as published in the reference's
config/libero/lb_tk8_65to72.py:33-165).  `scripts/train_libero_dp.py --config <this file>` reads `base['diffusion']`."""
from diffuser.libero.lb_constants import LB_GRASP_actdown_value_range_1
from diffuser.datasets import LB_ACTION_MIN, LB_ACTION_MAX

ACTION_HORIZON = 16          # actions predicted per policy call (8 of them are executed)

REPLAY = dict(               # the two episode buffers and how minibatches mix them
    envBuf_max_num_uB_rand=1200, envBuf_max_num_uB_vid=600, max_len_uB=700, min_len_uB=30,
    buf_sample_batch_size=64, buf_sample_method='rand_prob', buf_sample_randBuf_prob=0.3,
    buf_sample_ratio_rand=[0.75, 0.25], buf_sample_ratio_vid=[0.25, 0.75],
    batch_size=4, batch_size_v=1,
)
RANDOM_DATA = dict(          # random-action episodes read from disk
    rand_explo_type='from_h5', randsam_filename='lb_randsam_8tk_perTk500.hdf5',
    num_init_rand_Ep_per_tk=50, rand_explo_freq=500, rand_explo_num_Ep_per_tk=2,
)
SCHEDULE = dict(             # phases of the joint loop
    init_rand_steps=10000, rand_cycle_steps=100, vid_cycle_steps=400, video_explo_freq=200, use_env_rand_reset=True,
    enable_noExp=True, noExp_start_buf_len_rand=500, noExp_start_buf_len_vid=500,
    Exp_noExp_rand=(1000, 1000), Exp_noExp_vid=(1000, 1000), is_stop_at_suc=False,
)
ROLLOUT = dict(              # video-guided execution and the grasp heuristic
    model_act_horizon=ACTION_HORIZON, n_acts_per_pred=8, n_preds_betw_vframes=(4, 6),
    n_acts_down_range=(16, 16), n_acts_close_grp=8, act_down_val=None,
    act_down_val_range_per_tk=LB_GRASP_actdown_value_range_1, close_grp_force=0.98, close_grp_act_down_val=0,
    grasp_z_diff_limit=0.36, grasp_abs_z_limit=0.56,
)

trainer_dict = {}
for _group in (REPLAY, RANDOM_DATA, SCHEDULE, ROLLOUT):
    trainer_dict.update(_group)

base = {
    'dataset': "libero-8tk-65to72-v3",
    'diffusion': dict(
        # models
        model_yl_path='config/diff_policy/lb_train_diffusion_unet_image_orn10.yaml',
        vid_diffusion=dict(ckpts_dir='./ckpts/libero/libero_ep20_bs12_aug', milestone=180000, timestep=100, g_w=0, cls_free_prob=0.0,
                           sample_per_seq=8),
        input_img_size=(128, 128), render_img_size=(128, 128),
        # data
        loader='diffuser.libero.lb_online_dataset.LB_Online_Dataset',
        dataset_config=dict(act_min_max=(LB_ACTION_MIN, LB_ACTION_MAX), combo_type='all'),
        # optimisation
        loss_type='l2', n_train_steps=2e5, gradient_accumulate_every=1,
        opt_params=dict(lr=1.0e-4, betas=[0.95, 0.999], eps=1.0e-8, weight_decay=1.0e-6),
        ema_params=dict(update_after_step=0, inv_gamma=1.0, power=0.75, min_value=0.0, update_every=1, include_online_model=False),
        # trainer
        trainer_type='v7', do_train_resume=False, trainer_dict=trainer_dict,
        logbase='logs', prefix='diffusion/', save_freq=1000, sample_freq=5000, log_freq=100, n_saves=5, n_samples=1,
    ),
}
