"""LB_DP_Eval (diffuser/libero/lb_eval_helper.py:14-376): closed-loop evaluation of the EMA policy under video guidance --
per (task, seed): render, predict a goal video (re-planned `num_vid_pred_per_ep` times, each time after `use_vid_first_n_frames`
frames were consumed), follow each frame with `eval_n_preds_betw_vframes` policy calls of 8 actions, stop at success.
Same constructor, `run_evals()` result keys and `eval_1_env()` return tuple; the matplotlib / mp4 side outputs are not produced
(`img12` is None)."""
from typing import Dict, List
import numpy as np
import torch
from . import _host_utils as utils
from ._host_utils import imgs_preproc_simple_noCrop_v1

LB_1_VIDEO_PRED = []


class LB_DP_Eval(object):
    def __init__(self, gcp_model, ema, video_model, trainer, env_list, task_list: List[str], cam_list: List[str], valid_seeds,
                 max_episode_steps, render_img_size, rendered_imgs_preproc_fn, is_video_ddim: bool, is_dp_ddim: bool,
                 eval_n_preds_betw_vframes: int, save_path, vid_use_autocast=True, num_vid_pred_per_ep=5, use_vid_first_n_frames=2,
                 device='cuda'):
        self.gcp_model, self.ema, self.video_model, self.trainer = gcp_model, ema, video_model, trainer
        self.env_list, self.task_list, self.cam_list = env_list, task_list, cam_list
        self.valid_seeds = valid_seeds
        self.max_episode_steps = max_episode_steps
        self.render_img_size = render_img_size
        self.rendered_imgs_preproc_fn = rendered_imgs_preproc_fn
        self.n_acts_per_pred = self.trainer.n_acts_per_pred
        self.input_img_size = self.trainer.input_img_size
        self.accelerator = self.trainer.accelerator
        self.video_model.ema.ema_model.is_ddim_sampling = is_video_ddim
        self.is_dp_ddim = is_dp_ddim
        self.eval_n_preds_betw_vframes = eval_n_preds_betw_vframes
        self.num_vid_pred_per_ep = num_vid_pred_per_ep
        self.use_vid_first_n_frames = use_vid_first_n_frames
        self.save_path = save_path
        self.vid_use_autocast = vid_use_autocast
        self.device = device
        assert max_episode_steps == 500
        assert self.rendered_imgs_preproc_fn == imgs_preproc_simple_noCrop_v1
        self.pre_vid_gen_fn = lambda **kargs: None
        self.pvG_fn_args = {}
        self.after_vid_gen_fn = lambda **kargs: None
        self.avG_fn_args = {}
        self.is_stop_at_suc = True

    def run_evals(self, vis_gif=False, cur_num_iters=''):
        is_sucs_all, run_times_all = [], []
        is_sucs_per_tk: Dict[str, list] = {}
        run_times_per_tk = {}
        for tk in self.task_list:
            is_sucs_per_tk[tk], run_times_per_tk[tk] = [], []
            for cam_name in self.cam_list:
                if cam_name in ['agentview_image', 'agentview_rgb']:
                    cam_name = 'agent'
                for env_seed in self.valid_seeds:
                    e_idx = self.env_list.seed_sets[tk][0]
                    env = self.env_list.init_1_given_env(tk, env_idx=e_idx, e_seed=env_seed)
                    is_suc, _, run_time, _, _ = self.eval_1_env(env, tk, cam_name)
                    self.env_list.close_1_given_env(tk, e_idx)
                    is_sucs_all.append(is_suc)
                    is_sucs_per_tk[tk].append(is_suc)
                    run_times_all.append(run_time)
                    run_times_per_tk[tk].append(run_time)
        suc_rate_per_tk = {tk: np.mean(is_sucs_per_tk[tk]).item() for tk in self.task_list}
        return dict(suc_rate=np.mean(is_sucs_all).item(), num_evals=len(is_sucs_all), n_seeds=len(self.valid_seeds),
                    suc_rate_per_tk=suc_rate_per_tk, is_sucs_per_tk=is_sucs_per_tk, is_sucs_all=is_sucs_all,
                    run_times_all=run_times_all, run_times_per_tk=run_times_per_tk, seeds=self.valid_seeds)

    def eval_1_env(self, env, tk, cam_name):
        self.ema.ema_model.eval()
        self.video_model.ema.ema_model.eval()
        timer = utils.Timer()
        img_r = self.env_list.render_a_given_env(env, cam_name)
        assert type(img_r) == np.ndarray
        img_st = self.rendered_imgs_preproc_fn(img_r[None])
        assert img_st.ndim == 4 and img_st.shape[1] == 3 and img_st.shape[2:4] == self.input_img_size
        tasks_str = [tk]
        v_hzn = self.video_model.video_future_horizon
        is_suc = False
        imgs_out_dense = [img_st]
        all_full_pred_v = []
        cnt_vid_pred = 0
        num_vid_ppp = 1 if tk in LB_1_VIDEO_PRED else self.num_vid_pred_per_ep
        num_total_frames = (num_vid_ppp - 1) * self.use_vid_first_n_frames + v_hzn
        g_idx, pred_v = 0, None
        for fr_idx in range(num_total_frames):
            if cnt_vid_pred < num_vid_ppp and (fr_idx == 0 or g_idx == self.use_vid_first_n_frames - 1):
                self.pre_vid_gen_fn(**self.pvG_fn_args)
                with torch.no_grad():
                    preds_video = self.video_model.forward(img_st.to(self.device), tasks_str)
                self.after_vid_gen_fn(**self.avG_fn_args)
                assert len(preds_video) == 1
                pred_v = preds_video.detach()[0]
                all_full_pred_v.append(torch.cat([img_st.cpu(), pred_v.cpu()], dim=0))
                cnt_vid_pred += 1
                g_idx = 0
            else:
                g_idx += 1
            img_goal = pred_v[None, g_idx]
            n_preds = self.eval_n_preds_betw_vframes
            assert type(n_preds) == int
            for i_p in range(n_preds):
                img_st = img_st.to(self.device)
                with torch.no_grad():
                    batch = self.trainer.to_batch_dict(img_st, img_goal, None)
                    act = self.ema.ema_model.predict_action(batch['obs'], use_ddim=self.is_dp_ddim)['action'].cpu()
                act = act[0]
                assert len(act) == self.n_acts_per_pred
                act = act.clamp(min=self.trainer.act_min, max=self.trainer.act_max)
                assert act.shape[-1] == 7
                for i_a in range(self.n_acts_per_pred):
                    _, _, e_done, info = env.step(act[i_a].numpy())
                    img_cur = self.rendered_imgs_preproc_fn(self.env_list.render_a_given_env(env, cam_name=cam_name)[None])
                    imgs_out_dense.append(img_cur)
                    is_suc = bool(e_done) or is_suc
                img_st = torch.clone(imgs_out_dense[-1])
                assert img_st.ndim == 4 and img_st.shape[0] == 1
            if is_suc and self.is_stop_at_suc:
                break
        run_time = timer()
        imgs_np = [img[0].permute(1, 2, 0).cpu().numpy() for img in imgs_out_dense]
        return is_suc, imgs_np, run_time, all_full_pred_v, None
