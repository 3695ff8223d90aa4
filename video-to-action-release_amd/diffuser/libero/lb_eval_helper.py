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
        # plain keyword -> attribute copies (the reference's attribute names are the surface other modules read)
        for name, val in dict(valid_seeds=valid_seeds, max_episode_steps=max_episode_steps, render_img_size=render_img_size,
                              rendered_imgs_preproc_fn=rendered_imgs_preproc_fn, is_dp_ddim=is_dp_ddim,
                              eval_n_preds_betw_vframes=eval_n_preds_betw_vframes, num_vid_pred_per_ep=num_vid_pred_per_ep,
                              use_vid_first_n_frames=use_vid_first_n_frames, save_path=save_path, vid_use_autocast=vid_use_autocast,
                              device=device).items():
            setattr(self, name, val)
        for name in ("n_acts_per_pred", "input_img_size", "accelerator"):          # mirrored from the trainer
            setattr(self, name, getattr(trainer, name))
        video_model.ema.ema_model.is_ddim_sampling = is_video_ddim
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

    # ------------------------------------------------------------------------------------------------ one episode
    def _observe(self, env, cam_name):
        """Current camera frame as the model input: [1,3,H,W] float in [0,1] on the host."""
        frame = self.env_list.render_a_given_env(env, cam_name=cam_name)
        if type(frame) is not np.ndarray:
            raise TypeError("the environment must render numpy frames")
        obs = self.rendered_imgs_preproc_fn(frame[None])
        assert obs.ndim == 4 and obs.shape[:2] == (1, 3) and tuple(obs.shape[2:4]) == tuple(self.input_img_size)
        return obs

    def _plan_video(self, obs, task):
        """One call of the video model from the current observation -> predicted frames [T,3,H,W] (device tensor)."""
        self.pre_vid_gen_fn(**self.pvG_fn_args)
        with torch.no_grad():
            video = self.video_model.forward(obs.to(self.device), [task])
        self.after_vid_gen_fn(**self.avG_fn_args)
        assert len(video) == 1
        return video.detach()[0]

    def _act_towards(self, env, cam_name, obs, goal, frames_out):
        """One policy call towards `goal` and the execution of its n_acts_per_pred actions.  Returns (new observation, success)."""
        with torch.no_grad():
            query = self.trainer.to_batch_dict(obs.to(self.device), goal, None)
            acts = self.ema.ema_model.predict_action(query['obs'], use_ddim=self.is_dp_ddim)['action'].cpu()[0]
        assert len(acts) == self.n_acts_per_pred and acts.shape[-1] == 7
        acts = acts.clamp(min=self.trainer.act_min, max=self.trainer.act_max)
        success = False
        for a in acts:
            _, _, done, _ = env.step(a.numpy())
            frames_out.append(self._observe(env, cam_name))
            success = success or bool(done)
        return torch.clone(frames_out[-1]), success

    def eval_1_env(self, env, tk, cam_name):
        """Closed-loop episode (reference :168-373).  The goal video is re-planned `num_vid_pred_per_ep` times: after the first
        `use_vid_first_n_frames` frames of a plan were followed, a new plan starts from the current observation; every frame is
        pursued with `eval_n_preds_betw_vframes` policy calls.  Returns (is_suc, observed frames as HWC float arrays, seconds,
        [start frame + predicted video per plan], None)."""
        self.ema.ema_model.eval()
        self.video_model.ema.ema_model.eval()
        clock = utils.Timer()
        obs = self._observe(env, cam_name)
        horizon = self.video_model.video_future_horizon
        plans_allowed = 1 if tk in LB_1_VIDEO_PRED else self.num_vid_pred_per_ep
        total_goals = (plans_allowed - 1) * self.use_vid_first_n_frames + horizon
        n_calls = self.eval_n_preds_betw_vframes
        assert type(n_calls) == int
        observed, plans = [obs], []
        video, goal_idx, solved = None, 0, False
        for step in range(total_goals):
            replan = len(plans) < plans_allowed and (step == 0 or goal_idx == self.use_vid_first_n_frames - 1)
            if replan:
                video = self._plan_video(obs, tk)
                plans.append(torch.cat([obs.cpu(), video.cpu()], dim=0))
                goal_idx = 0
            else:
                goal_idx += 1
            goal = video[None, goal_idx]
            for _ in range(n_calls):
                obs, ok = self._act_towards(env, cam_name, obs, goal, observed)
                solved = solved or ok
            if solved and self.is_stop_at_suc:
                break
        seconds = clock()
        return solved, [o[0].permute(1, 2, 0).cpu().numpy() for o in observed], seconds, plans, None
