"""LB_Online_Trainer_V7 -- the joint loop of the reference (diffuser/libero/lb_online_trainer_v7.py:29-1347) on the MI355X path:

    every `video_explo_freq` steps:   for each (task, cam, env) of this rank:
                                          video model samples 7 goal frames from the current render        (HIP UNet sampler)
                                          EMA policy follows them, `n_preds` x 8 actions per frame         (HIP predict_action)
                                          grasp heuristic on the wrist depth; episode -> envBuf_vid        (uint8, HBM resident)
    every `rand_explo_freq` steps:    more random-action episodes from the HDF5 file -> envBuf_rand
    every step:                       sample_from_bufs -> compute_loss -> backward -> clip -> AdamW -> zero -> EMA
                                      = one hipGraph replay of v2a_hip.trainer.PolicyTrainer (+ RCCL all-reduce when world > 1)

Same constructor keywords, attributes (`.opt .ema .accelerator .results_folder .step .gcp_model .video_model .envBuf_rand
.envBuf_vid ...`), schedule state machines (`update_iter_type`, `update_explo_type`), RNG call order and checkpoint keys as the
reference, so `scripts/train_libero_dp.py` and `LB_DP_Eval` drive it unchanged.  Differences, all on purpose:
  * the replay buffers are `v2a_hip.replay.ReplayStore`s over one HBM pool (uint8 frames) instead of deques of fp32 CPU tensors;
  * clip / AdamW / zero_grad / EMA run fused on the device: `.opt` and `.ema` are handles over that state which keep the
    `zero_grad / state_dict / load_state_dict / ema_model / update` surface (torch.optim.AdamW / ema_pytorch key layouts);
  * `accelerate` is not required: `.accelerator` is a small object with the attributes the scripts read; one process per GPU, rank r
    explores the (task, cam, env) combinations r, r+N, ... and trains on its own buffers;
  * figure / gif / wandb side effects of the reference's debug mode are not produced (metrics go to `metrics.jsonl`).
"""
import contextlib
import json
import os
import os.path as osp
from copy import deepcopy
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from v2a_hip.replay import ReplayStore, sample_mixed
from v2a_hip.trainer import PolicyTrainer
from . import _host_utils as utils
from ._host_utils import imgs_preproc_simple_noCrop_v1
from .lb_randsam_io import open_randsam, check_and_clip_actions

__version__ = "v2a-mi355x-0.1"


def exists(x):
    return x is not None


def cycle(dl):
    while True:
        for data in dl:
            yield data


_PHASES = ('explo', 'no-explo')


def _tick_phase(obj, buf, counted, quota, always_check):
    """One call of the explore / pause machine of buffer `buf` ('rand' | 'vid'); state lives on `obj` under the reference's
    attribute names (explo_type_<buf>, cnt_exp_<buf>, cnt_no_exp_<buf>: they are checkpoint / logging surface)."""
    kind_attr = f'explo_type_{buf}'
    counter = {'explo': f'cnt_exp_{buf}', 'no-explo': f'cnt_no_exp_{buf}'}
    if counted:
        kind = getattr(obj, kind_attr)
        if kind not in counter:
            raise AssertionError(f"unknown exploration phase {kind!r}")
        setattr(obj, counter[kind], getattr(obj, counter[kind]) + 1)
    elif not always_check:
        return
    for i, kind in enumerate(_PHASES):
        if getattr(obj, counter[kind]) == quota[i]:
            setattr(obj, counter[kind], 0)
            setattr(obj, kind_attr, _PHASES[1 - i])


class _Accelerator:
    """The attributes of accelerate.Accelerator that train_libero_dp.py / plan_lb.py / LB_DP_Eval touch."""

    def __init__(self):
        self.num_processes = int(os.environ.get("WORLD_SIZE", "1"))
        self.process_index = int(os.environ.get("RANK", "0"))
        self.local_process_index = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise RuntimeError("LB_Online_Trainer_V7 needs a GPU: the policy / video hot path has no CPU fallback")
        self.device = torch.device("cuda", self.local_process_index)
        torch.cuda.set_device(self.device)
        self.process_group = None
        if self.num_processes > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("nccl", device_id=self.device)        # RCCL over xGMI
            self.process_group = dist.group.WORLD
        self.scaler = None
        self.native_amp = False
        self.is_main_process = self.process_index == 0
        self.is_local_main_process = self.local_process_index == 0

    def prepare(self, *objs):
        return objs[0] if len(objs) == 1 else objs

    def unwrap_model(self, m):
        return m

    def get_state_dict(self, m):
        return m.state_dict()

    def autocast(self):
        return contextlib.nullcontext()           # the HIP path picks its own precision (v2a_hip.set_precision)

    def wait_for_everyone(self):
        if self.num_processes > 1:
            import torch.distributed as dist
            dist.barrier(group=self.process_group)

    def print(self, *a, **k):
        if self.is_main_process:
            print(*a, **k)


class _OptHandle:
    """`trainer.opt`: torch.optim.AdamW's surface over the fused device optimiser."""

    def __init__(self, ptrainer: PolicyTrainer):
        self._t = ptrainer
        named = list(ptrainer.policy.named_parameters())
        fused_index = {n: i for i, n in enumerate(ptrainer.names)}
        self._ckpt_names = [n for n, _ in named]                                # .parameters() order, as AdamW(params) saw it
        self._order = [fused_index.get(n, -1) for n in self._ckpt_names]        # -1: never receives a gradient (no state)

    @property
    def param_groups(self):
        return [dict(self._t.opt.hyper, params=[dict(self._t.policy.named_parameters())[n] for n in self._ckpt_names])]

    def zero_grad(self, set_to_none=True):
        self._t.arena.zero_()
        for p in self._t.policy.parameters():
            p.grad = None

    def state_dict(self):
        return self._t.opt.state_dict(order=self._order)

    def load_state_dict(self, sd):
        self._t.opt.load_state_dict(sd, order=self._order)

    def step(self):
        raise RuntimeError("the optimiser step is fused into PolicyTrainer.step(); call trainer.train() / trainer.train_step()")


class _EMAHandle:
    """`trainer.ema`: ema_pytorch.EMA's surface (`.ema_model`, `.update()`, state_dict with online_model./ema_model./initted/step)."""

    def __init__(self, ptrainer: PolicyTrainer):
        self._t = ptrainer
        self._fresh = -1

    @property
    def online_model(self):
        return self._t.policy

    @property
    def ema_model(self):
        if self._fresh != self._t.step_count:           # the fused kernel updates the EMA weights behind torch's back
            self._t.ema_for_inference()
            self._fresh = self._t.step_count
        return self._t.ema_policy

    def update(self):
        pass                                            # done by the fused optimiser kernel of the same step

    def to(self, device):
        return self

    def eval(self):
        self._t.ema_policy.eval()
        return self

    def state_dict(self):
        _, ema_step, initted = self._t.opt.counters()
        sd = {}
        if self._t.ema_include_online_model:            # ema_pytorch 0.2.3: `include_online_model=False` keeps online_model.* out
            sd.update({f"online_model.{k}": v for k, v in self._t.policy.state_dict().items()})
        sd.update({f"ema_model.{k}": v for k, v in self._t.ema_policy.state_dict().items()})
        sd["initted"] = torch.tensor([initted])
        sd["step"] = torch.tensor([ema_step])
        return sd

    def load_state_dict(self, sd):
        self._t.ema_policy.load_state_dict({k[len("ema_model."):]: v for k, v in sd.items() if k.startswith("ema_model.")})
        step, _, _ = self._t.opt.counters()
        self._t.opt.set_counters(step, int(sd["step"].item()), bool(sd["initted"].item()))
        self._fresh = -1


class LB_Online_Trainer_V7(object):
    def __init__(self, init_diff_policy, video_model, tokenizer, text_encoder, train_set, valid_set, channels=3, *,
                 train_batch_size=1, video_batch_size=1, valid_batch_size=1, gradient_accumulate_every=1,
                 augment_horizontal_flip=True, train_num_steps=100000, opt_params, ema_params, render_img_size=(320, 240),
                 input_img_size=(128, 128), sample_freq=1000, save_freq=1000, label_freq=50000, log_freq=100, num_samples=3,
                 results_folder='./results', amp=True, fp16=True, split_batches=False, trainer_dict, **kwargs):
        super().__init__()
        self.tokenizer = tokenizer
        self.text_encoder = text_encoder
        self.accelerator = _Accelerator()
        acc = self.accelerator

        self.gcp_model = init_diff_policy.diffusion_policy
        self.video_model = video_model.to(self.device)
        assert tuple(video_model.ema.ema_model.image_size) == tuple(input_img_size)
        self.video_model.requires_grad_(False)
        self.video_model.eval()

        self.channels = channels
        self.num_samples = num_samples
        self.sample_freq, self.save_freq, self.label_freq, self.log_freq = sample_freq, save_freq, label_freq, log_freq
        self.render_img_size = render_img_size
        self.input_img_size = tuple(input_img_size)
        self.batch_size_rand = train_batch_size
        self.valid_batch_size = valid_batch_size
        if gradient_accumulate_every != 1:
            raise NotImplementedError("the fused train step takes one minibatch per optimiser step (config: gradient_accumulate_every=1)")
        self.gradient_accumulate_every = gradient_accumulate_every
        self.train_num_steps = int(train_num_steps)

        self.env_list = train_set.env_list
        self.task_list = self.env_list.task_list
        self.train_set = train_set
        self.train_set_len = len(train_set)

        dl = DataLoader(self.train_set, batch_size=train_batch_size, shuffle=True, num_workers=0)
        self.dl = cycle(dl)
        assert video_batch_size == 1                 # (the DataLoader's batch, as in the reference; the exploration round batches the SAMPLER itself)
        mine = list(range(acc.process_index, len(self.train_set), acc.num_processes))     # this rank's rollout combinations
        self.dl_vid = DataLoader(Subset(self.train_set, mine), batch_size=video_batch_size, shuffle=True, num_workers=0)

        self.gcp_model.to(self.device)
        if hasattr(self.text_encoder, "to"):
            self.text_encoder.to(self.device)
        assert not getattr(self.text_encoder, "training", False)

        self.results_folder = Path(results_folder)
        self.results_folder.mkdir(exist_ok=True, parents=True)
        self.step = 0
        self.num_steps_in_env = 0

        self.act_min_np, self.act_max_np = self.train_set.act_min_max
        assert (np.abs(self.act_min_np) == self.act_max_np).all(), 'necessary for our sampling'
        self.act_min, self.act_max = torch.tensor(self.act_min_np), torch.tensor(self.act_max_np)
        self.act_dim = len(self.act_max)

        self.trainer_dict = trainer_dict
        self.lr_warmupDecay = trainer_dict.get('lr_warmupDecay', False)
        if self.lr_warmupDecay:
            raise NotImplementedError("lr_warmupDecay (unused by the released configs)")
        self.cur_mode = trainer_dict.get('cur_mode', 'train')
        self._opt_params, self._ema_params = dict(opt_params), dict(ema_params)
        self.init_helpers()
        self.control_mode = 'delta'
        self.debug = False

    # ------------------------------------------------------------------------------------------------------------ set-up
    def init_helpers(self):
        td = self.trainer_dict
        self.num_envs_per_tk = self.env_list.num_seed_per_task
        assert self.num_envs_per_tk == 1
        self.num_envs_all_tks = self.num_envs_per_tk * self.env_list.num_tasks
        assert self.train_set_len == self.num_envs_all_tks

        self.envBuf_max_num_uB_rand = td['envBuf_max_num_uB_rand']
        self.envBuf_max_num_uB_vid = td['envBuf_max_num_uB_vid']
        self.max_len_uB, self.min_len_uB = td['max_len_uB'], td['min_len_uB']
        if not td.get('allow_small_buffers', False):                       # the reference's sanity floor (:198-199)
            assert self.envBuf_max_num_uB_rand >= 1000
            assert self.envBuf_max_num_uB_vid >= 500
        self.model_act_horizon = td['model_act_horizon']

        # both buffers in one HBM pool: a mixed minibatch is one gather launch (v2a_hip.replay.ReplayStore.pair)
        self.envBuf_rand, self.envBuf_vid = ReplayStore.pair(
            self.envBuf_max_num_uB_rand, self.envBuf_max_num_uB_vid, self.max_len_uB, self.min_len_uB,
            capacity_a=td.get('pool_frames_rand'), capacity_b=td.get('pool_frames_vid'), image_hw=self.input_img_size,
            act_dim=self.act_dim, act_len=self.model_act_horizon, device=self.device)

        self.num_init_rand_episodes_per_tk = td['num_init_rand_Ep_per_tk']
        self.buf_sample_method = td.get('buf_sample_method', 'iter_bias_fix')
        assert self.buf_sample_method in ['iter_bias_fix', 'rand_prob', 'iter_bias_rand']
        if self.buf_sample_method != 'rand_prob':
            raise NotImplementedError("only buf_sample_method='rand_prob' (the released configs) is fused into the train step")
        self.buf_sample_randBuf_prob = td['buf_sample_randBuf_prob']

        rs_fname = td['randsam_filename']
        if rs_fname.startswith('synthetic') or osp.isabs(rs_fname):
            self.randsam_file_path = rs_fname
        else:
            self.randsam_file_path = osp.join('./data_dir/scratch/libero/env_rand_samples/', rs_fname)
        assert self.cur_mode in ['train', 'eval']
        self.randsam = open_randsam(self.randsam_file_path, env_list=self.env_list)
        self.h5_total_num_ep_per_task = self.randsam.num_episodes(self.task_list[0])
        if self.cur_mode == 'train':
            assert self.h5_total_num_ep_per_task >= self.num_init_rand_episodes_per_tk

        self.task_env_states = deepcopy(self.env_list.env_init_states)
        self.rendered_imgs_preproc_fn = imgs_preproc_simple_noCrop_v1

        self.iter_type = 'rand-bias'
        self.init_rand_steps = td['init_rand_steps']
        self.rand_cycle_steps, self.vid_cycle_steps = td['rand_cycle_steps'], td['vid_cycle_steps']
        self.use_env_rand_reset = td['use_env_rand_reset']
        self.video_explo_freq, self.rand_explo_freq = td['video_explo_freq'], td['rand_explo_freq']
        self.rand_explo_type = td['rand_explo_type']
        assert self.rand_explo_type == 'from_h5'
        self.rand_explo_num_episodes_per_tk = td['rand_explo_num_Ep_per_tk']

        self.explo_type_rand = 'explo'
        self.explo_type_vid = 'explo'
        self.enable_noExp = td.get('enable_noExp', False)
        self.noExp_start_buf_len_rand = td.get('noExp_start_buf_len_rand')
        self.noExp_start_buf_len_vid = td.get('noExp_start_buf_len_vid')
        self.Exp_noExp_rand = td.get('Exp_noExp_rand')
        self.Exp_noExp_vid = td.get('Exp_noExp_vid')
        self.cnt_no_exp_rand = self.cnt_exp_rand = self.cnt_no_exp_vid = self.cnt_exp_vid = 0
        if self.init_rand_steps != -1:
            assert self.init_rand_steps % self.rand_cycle_steps == 0
        self.rand_iter_cnt = self.vid_iter_cnt = 0

        self.cnt_explore_suc = 0
        self.cnt_vid_rollouts = 0
        self.cnt_explo_suc_per_tk = {tk: 0 for tk in self.task_list}
        self.cnt_vid_rout_per_tk = {tk: 0 for tk in self.task_list}

        self.n_acts_per_pred = td['n_acts_per_pred']
        self.n_preds_betw_vframes = td['n_preds_betw_vframes']
        self.max_acts_betw_vframes = self.n_acts_per_pred * self.n_preds_betw_vframes[1]
        total_rollout_acts = self.max_acts_betw_vframes * self.video_model.video_future_horizon
        assert self.max_len_uB > total_rollout_acts, 'must store all rollout'

        self.buf_sample_batch_size = td['buf_sample_batch_size']
        bsr_rand, bsr_vid = td['buf_sample_ratio_rand'], td['buf_sample_ratio_vid']
        self.nums_buf_sample_rand = utils.number_by_ratio(self.buf_sample_batch_size, bsr_rand)
        self.nums_buf_sample_vid_bias = utils.number_by_ratio(self.buf_sample_batch_size, bsr_vid)
        self.r_prob_rand, self.r_prob_vid = bsr_rand[0], bsr_vid[0]

        self.n_acts_down_range = td['n_acts_down_range']
        self.n_acts_close_grp = td['n_acts_close_grp']
        self.act_down_val = td['act_down_val']
        if 'act_down_val_range_per_tk' in td:
            self.act_down_val_range_per_tk = td['act_down_val_range_per_tk']
            assert self.act_down_val is None
        else:
            assert self.act_down_val <= -0.5
        self.close_grp_force = td['close_grp_force']
        self.close_grp_act_down_val = td['close_grp_act_down_val']
        assert self.close_grp_act_down_val <= 0
        self.grasp_z_diff_limit = td['grasp_z_diff_limit']
        self.grasp_abs_z_limit = td['grasp_abs_z_limit']
        self.grp_cam_name = 'gripper'
        self.is_all_randsam_visited = False
        self.is_stop_at_suc = td['is_stop_at_suc']

        # ---- the fused device train step + the handles the scripts expect
        acc = self.accelerator
        self.ptrainer = PolicyTrainer(self.gcp_model, self.envBuf_rand, batch_size=self.buf_sample_batch_size,
                                      opt_params=self._opt_params, ema_params=self._ema_params,
                                      seed=td.get('seed', 0), use_graph=td.get('use_graph', True), process_group=acc.process_group,
                                      world_size=acc.num_processes, rank=acc.process_index, store_vid=self.envBuf_vid,
                                      rand_prob=self.buf_sample_randBuf_prob)
        self.opt = _OptHandle(self.ptrainer)
        self.ema = _EMAHandle(self.ptrainer)
        self.ema.ema_model.normalizer.to_device(self.device)
        self._graphed_predict = None

    @property
    def device(self):
        return self.accelerator.device

    # ------------------------------------------------------------------------------------------------------ checkpoints
    # keys of the reference's checkpoint dict (lb_online_trainer_v7.py:371-382): the compatibility contract of save / load
    _CKPT_SCALARS = ('step', 'num_steps_in_env', 'cnt_vid_rollouts', 'cnt_vid_rout_per_tk')

    def save(self, milestone):
        self.ptrainer.verify_exchange()          # (every rank) a step whose gradient exchange gave up must not reach a checkpoint
        if not self.accelerator.is_local_main_process:
            return
        data = {k: getattr(self, k) for k in self._CKPT_SCALARS}
        data.update(gcp_model=self.accelerator.get_state_dict(self.gcp_model), opt=self.opt.state_dict(), ema=self.ema.state_dict(),
                    scaler=self.ptrainer.opt.scaler_state_dict(), version=__version__)      # GradScaler layout (reference :377), None outside fp16
        savepath = str(self.results_folder / f'model-{milestone}.pt')
        torch.save(data, savepath)
        utils.print_color(f'[ utils/training ] Saved model to {savepath}', c='y')

    def load(self, milestone):
        data = torch.load(str(self.results_folder / f'model-{milestone}.pt'), map_location=self.device, weights_only=False)
        self.gcp_model.load_state_dict(data['gcp_model'])
        self.step = data['step']
        self.num_steps_in_env = data['num_steps_in_env']
        self.opt.load_state_dict(data['opt'])
        self.ema.load_state_dict(data['ema'])
        if data.get('scaler') and self.ptrainer.loss_scaling:      # fp16 mode: resume the loss scale and its growth tracker (reference :406-407)
            self.gcp_model.engine.loss_scale_ptr = self.ptrainer.opt.load_scaler_state_dict(data['scaler'])
        self.gcp_model.engine.refresh_packs()            # parameters changed behind the packed copies
        if 'version' in data:
            print(f"loading from version {data['version']}")

    def encode_batch_text(self, batch_text):
        ids = self.tokenizer(batch_text, return_tensors='pt', padding=True, truncation=True, max_length=128).to(self.device)
        return self.text_encoder(**ids).last_hidden_state

    # ---------------------------------------------------------------------------------------------------- env utilities
    def env_get_preproc_imgs(self, tasks_str, cams_str, env_idxs):
        assert len(tasks_str) == 1
        imgs = []
        for i_sam, tk in enumerate(tasks_str):
            img = self.env_list.render_an_env(tk, cams_str[i_sam], env_idxs[i_sam])
            assert img.shape[:2] == self.input_img_size
            imgs.append(img)
        return self.rendered_imgs_preproc_fn(np.array(imgs))

    # --------------------------------------------------------------------------------------------------- schedule logic
    def update_explo_type(self):
        """Per buffer, alternate an exploring phase and a pause once the buffer holds `noExp_start_buf_len_*` episodes: a phase ends
        after `Exp_noExp_*[phase]` counted calls.  Behaviour pinned by a trace of the reference's method (:432-468) over three
        configurations (tests/golden/schedule.npz); the rand buffer evaluates its phase ends on every call, the rollout buffer only
        on counted calls -- visible when a quota is 0."""
        if not self.enable_noExp:
            return
        _tick_phase(self, 'rand', len(self.envBuf_rand) >= self.noExp_start_buf_len_rand, self.Exp_noExp_rand, always_check=True)
        _tick_phase(self, 'vid', len(self.envBuf_vid) >= self.noExp_start_buf_len_vid, self.Exp_noExp_vid, always_check=False)

    def update_iter_type(self):
        """Which buffer the next batches favour (reference :942-970, pinned by the schedule trace fixture): the rand buffer for the
        first `init_rand_steps`; afterwards the two counters hand over to each other when one reaches its cycle length; a cycle
        length of 0 disables that side altogether."""
        RAND, VID = 'rand-bias', 'vid-bias'
        step, warm = self.step, self.init_rand_steps
        if step < warm:
            self.iter_type = RAND
        elif step == warm:
            self.rand_iter_cnt = 0                     # the cycles start counting here
        elif self.rand_iter_cnt == self.rand_cycle_steps:
            self.rand_iter_cnt, self.iter_type = 0, VID
        elif self.vid_iter_cnt == self.vid_cycle_steps:
            self.vid_iter_cnt, self.iter_type = 0, RAND
        forced = RAND if self.vid_cycle_steps == 0 else (VID if self.rand_cycle_steps == 0 else None)
        if forced is not None:
            self.iter_type = forced
        if self.iter_type not in (RAND, VID):
            raise AssertionError(f"iter_type {self.iter_type!r}")

    # -------------------------------------------------------------------------------------------------------- train loop
    def train_step(self):
        """Steps 6-7 + the optimiser segment of the reference loop (:558-624) as one device-side step."""
        loss = self.ptrainer.step()
        self.step += 1
        return loss

    def _explore_due(self, every, phase):
        """Exploration of one kind runs on every `every`-th step after the initial random phase, while its phase is 'explo'."""
        return self.step > self.init_rand_steps and self.step % every == 0 and phase == 'explo'

    def _top_up_rand_buffer(self):
        """The next `rand_explo_num_Ep_per_tk` episodes per task from the random-action file; the cursor wraps around the file, a
        window never straddles its end (reference :517-526)."""
        per_task = self.h5_total_num_ep_per_task
        first = self.h5_randsam_start_idx % per_task
        count = min(per_task - first, self.rand_explo_num_episodes_per_tk)
        self.h5_add_rand_act_episodes_to_Buf(first, first + count)
        self.h5_randsam_start_idx += count
        self.is_all_randsam_visited = self.is_all_randsam_visited or self.h5_randsam_start_idx >= per_task

    def _count_iteration(self):
        counter = {'rand-bias': 'rand_iter_cnt', 'vid-bias': 'vid_iter_cnt'}.get(self.iter_type)
        if counter is None:
            raise NotImplementedError(self.iter_type)
        if self.iter_type == 'vid-bias' and len(self.envBuf_rand) == 0 and self.init_rand_steps != -1:
            raise AssertionError("vid-bias iterations need a non-empty random buffer")
        setattr(self, counter, getattr(self, counter) + 1)

    def train(self):
        acc = self.accelerator
        timer = utils.Timer()
        self.gcp_model.train()
        if len(self.envBuf_rand) == 0:                 # (a second train() call in the same process keeps its buffers)
            print('Start training, fill init rand buf')
            self.h5_add_rand_act_episodes_to_Buf(0, self.num_init_rand_episodes_per_tk)
            assert len(self.envBuf_rand) == self.env_list.num_tasks * self.num_init_rand_episodes_per_tk
            self.h5_randsam_start_idx = self.num_init_rand_episodes_per_tk
        metrics_path = self.results_folder / 'metrics.jsonl'

        while self.step < self.train_num_steps:
            self.update_iter_type()
            self.update_explo_type()
            if self._explore_due(self.video_explo_freq, self.explo_type_vid):
                self.video_guided_explore()
            if self._explore_due(self.rand_explo_freq, self.explo_type_rand):
                self._top_up_rand_buffer()
            self._count_iteration()

            loss = self.train_step()

            if acc.is_main_process:
                if self.step % self.save_freq == 0 or self.step == 1:
                    self.save(self.step // self.label_freq * self.label_freq)
                if self.step % self.log_freq == 0 or self.step == 1:
                    lv = float(loss.item())
                    print(f'{self.step}: {lv:8.4f} | t: {timer():8.4f}', flush=True)
                    m = {'train/it': self.step, 'train/loss': lv, 'train/lr': self.ptrainer.opt.hyper['lr'],
                         'train/num_steps_in_env': self.num_steps_in_env, 'train/cnt_explore_suc': self.cnt_explore_suc,
                         'buf/len_envBuf_rand': len(self.envBuf_rand), 'buf/len_envBuf_vid': len(self.envBuf_vid),
                         'explo/cnt_vid_rollouts': self.cnt_vid_rollouts}
                    m.update(self.make_wandb_dict_per_tk())
                    with open(metrics_path, 'a') as f:
                        f.write(json.dumps(m) + '\n')
                if self.sample_freq and self.step % self.sample_freq == 0:
                    self.ema.ema_model.eval()
        acc.print('training complete')

    # ------------------------------------------------------------------------------------------------------ buffer input
    def h5_add_rand_act_episodes_to_Buf(self, start_ep_idx, end_ep_idx):
        """Random-action episodes [start, end) of every task -> envBuf_rand (reference :718-780): range check (+-0.012 slack),
        clip to the action limits, uint8 frames go to HBM as they are."""
        buf, src = self.envBuf_rand, self.randsam
        had = len(buf)
        cam = self.env_list.camera_list[0]
        last = min(end_ep_idx, self.h5_total_num_ep_per_task)
        for tk in self.task_list:
            if end_ep_idx > last:                      # asked past the end of the file: it must really end there
                assert not src.has(tk, max(start_ep_idx, last))
            env_idx = self.env_list.seed_sets[tk][0]
            for ep in range(start_ep_idx, last):
                frames, actions = src.episode(tk, ep)
                actions = check_and_clip_actions(actions, self.act_min_np, self.act_max_np)
                if len(frames) != len(actions) + 1:
                    raise AssertionError(f"{tk}/{ep}: {len(frames)} frames for {len(actions)} actions")
                if not self.is_all_randsam_visited:    # environment steps are only counted on the first pass over the file
                    self.num_steps_in_env += len(actions)
                buf.add_one_episode(tk, cam, env_idx, torch.from_numpy(np.ascontiguousarray(frames)), torch.from_numpy(actions))
        utils.print_color(f'[Rand Buf Size Before Load] ep {had}', c='y')
        utils.print_color(f'[Rand Buf Size After Load] ep {len(buf)}', c='y')

    def sample_from_bufs(self):
        """(imgs_start, imgs_goal, acts, tasks_str, info) with the reference's mixing rule (:787-851); tensors are on the GPU."""
        return sample_mixed(self.envBuf_rand, self.envBuf_vid, self.buf_sample_batch_size, self.buf_sample_randBuf_prob)

    # ------------------------------------------------------------------------------------------- video-guided exploration
    def video_guided_explore(self):
        """One exploration round (reference :859-938): for every (task, camera, env) combination of this rank sample the 7 goal frames from
        the current render (HIP UNet sampler), follow them with the EMA policy (lb_rollout.py) and push the episode into envBuf_vid.

        trainer_dict['explore_batched'] (default True): the round's sampler calls run as ONE batched call -- every combination's environment
        is created and rendered first, `video_model.forward` runs once at B = number of combinations (rows are independent: per-sample
        GroupNorm, per-frame attention; one Philox seed per row, so row j draws exactly the noise of the bs-1 call with that seed and
        its frames equal that call's up to the kernels' batch-size-dependent summation order, <= 1e-4 in fp32: tests/test_joint_loop.py),
        then the rollouts run in the same order.  2.3 s instead of 5.2 s of sampling per 8-task round
        (bench.py `video_round8`).  Every task string is still tokenised and encoded on its own (no padding: `Video_PredModel.encode_rows`),
        so each row's conditioning is the one-at-a-time loop's; what differs from it: the host generators see the round's environment draws
        before its rollout draws.  False: the reference's
        order, one combination at a time."""
        self.env_list.check_no_envs_exist()
        n_before = len(self.envBuf_vid)
        utils.print_color(f'[Vid Exp] self.step {self.step}', c='y')
        if self.trainer_dict.get('explore_batched', True):
            combos = []
            for tasks, cams, idxs in self.dl_vid:
                tasks, cams, idxs = list(tasks), list(cams), idxs.cpu().numpy()
                combos += [(tasks[i], cams[i], idxs[i]) for i in range(len(tasks))]
            starts = []
            for tk, cam, idx in combos:
                self.env_list.init_1_given_env(tk_name=tk, env_idx=idx, is_rand=True)
                starts.append(self.env_get_preproc_imgs([tk], [cam], [idx]))
            start_all = torch.cat(starts, dim=0)
            all_tasks = [c[0] for c in combos]
            from flowdiffusion.flowdiffusion.goal_diffusion import _draw_philox_seed
            seeds = [_draw_philox_seed(self.device) for _ in combos]          # one sampler seed per combination, in round order
            from ..models.video_model import _spaced
            with torch.no_grad():
                # every task string is tokenised and encoded ON ITS OWN (unpadded, as the reference's bs = 1 loop does); only the sampler
                # call is batched -- a ragged list of token features (GoalGaussianDiffusion.sample embeds each row separately)
                emb = self.video_model.encode_rows(_spaced(all_tasks))
                if not self.trainer_dict.get('_explore_rows_one_by_one'):
                    videos = self.video_model.forward(start_all.to(self.device), emb, row_seeds=seeds)
                else:     # test hook: the same rows through bs-1 calls of the plain [1, L, 512] form (the reference's call), same seeds
                    videos = torch.cat([self.video_model.forward(start_all[j:j + 1].to(self.device), emb[j], row_seeds=seeds[j:j + 1])
                                        for j in range(len(combos))], dim=0)
            self._last_explore_videos = videos
            for j, (tk, cam, idx) in enumerate(combos):
                frames, acts = self.envs_video_guided_execute([tk], [cam], [idx], start_all[j:j + 1], videos[j:j + 1])
                self.env_list.close_1_given_env(tk_name=tk, env_idx=idx)
                self.envBuf_vid.add_one_episode(tk, cam, idx, frames[0], acts[0])
        else:
            for tasks, cams, idxs in self.dl_vid:
                tasks, cams, idxs = list(tasks), list(cams), idxs.cpu().numpy()
                if len(tasks) != 1:
                    raise AssertionError("rollouts run one combination at a time (video_batch_size == 1)")
                self.env_list.init_1_given_env(tk_name=tasks[0], env_idx=idxs[0], is_rand=True)
                start = self.env_get_preproc_imgs(tasks, cams, idxs)
                with torch.no_grad():
                    video = self.video_model.forward(start.to(self.device), tasks)
                frames, acts = self.envs_video_guided_execute(tasks, cams, idxs, start, video)
                self.env_list.close_1_given_env(tk_name=tasks[0], env_idx=idxs[0])
                for i, tk in enumerate(tasks):
                    self.envBuf_vid.add_one_episode(tk, cams[i], idxs[i], frames[i], acts[i])
        utils.print_color(f'Finish Vid Explore, vid buf before: {n_before}, after: {len(self.envBuf_vid)}')
        self.env_list.check_no_envs_exist()

    def _predict(self, img_st, img_goal):
        """EMA policy, DDIM-8: `[1,3,H,W]` start / goal -> clamped actions [n_acts_per_pred, 7] on the host."""
        batch = self.to_batch_dict(img_st, img_goal, None)
        # default: one hipGraph replay per call with the eight scheduler steps as one persistent launch (v2a_hip.inference, 4.8 ms against
        # ~3x that for the eager layer-by-layer call); the trajectory noise then comes from the device Philox stream.  graphed_rollout=False
        # keeps the eager predict_action, whose noise is torch.randn's (the reference's call: diffusion_unet_image_policy.py:97-101)
        if self.trainer_dict.get('graphed_rollout', True):
            from v2a_hip.inference import GraphedPredictAction
            ema = self.ema.ema_model                                   # refreshes the packed EMA weights if stale
            if self._graphed_predict is None:
                # per-rank noise stream (as PolicyTrainer's: seed + 7919 x rank), or every data-parallel rank would draw the same
                # trajectory noise for its k-th call
                self._graphed_predict = GraphedPredictAction(ema, batch_size=1, use_ddim=True,
                                                             seed=self.trainer_dict.get('seed', 0) + 7919 * int(self.accelerator.process_index))
            act = self._graphed_predict(batch['obs'])['action'].cpu()
        else:
            act = self.ema.ema_model.predict_action(batch['obs'], use_ddim=True)['action'].cpu()
        act = act[0]
        assert len(act) == self.n_acts_per_pred
        return act.clamp(min=self.act_min, max=self.act_max)

    def _rollout_runner(self):
        td = self.trainer_dict
        from .lb_rollout import RolloutRunner
        return RolloutRunner(self.env_list, lambda o, g: self._predict(o.to(self.device), g), self.rendered_imgs_preproc_fn,
                             n_acts=self.n_acts_per_pred, n_preds=self.n_preds_betw_vframes, grip_force=self.close_grp_force,
                             descend_steps=self.n_acts_down_range, close_steps=self.n_acts_close_grp, descend_speed=self.act_down_val,
                             descend_speed_per_task=getattr(self, 'act_down_val_range_per_tk', None),
                             close_descend_speed=self.close_grp_act_down_val, z_gap=self.grasp_z_diff_limit,
                             z_ceiling=self.grasp_abs_z_limit, wrist_cam=self.grp_cam_name, stop_at_success=self.is_stop_at_suc)

    def envs_video_guided_execute(self, tasks_str, cams_str, env_idxs, imgs_start, preds_video, vis_rollout=False):
        """Per sample: follow the 7 predicted frames with the EMA policy (diffuser/libero/lb_rollout.py) -> uint8 [T+1,H,W,3] frames
        and float [T,7] actions, the episode format of the HBM replay store."""
        preds_video = preds_video.detach().to(self.device)
        if tuple(imgs_start.shape[2:4]) != self.input_img_size or preds_video.shape[1] != self.video_model.video_future_horizon:
            raise AssertionError("start images / predicted video do not have the configured size")
        assert self.control_mode == 'delta'
        runner = self._rollout_runner()
        frames_out, acts_out = [], []
        for i, tk in enumerate(tasks_str):
            frames, acts, ok = runner.run(tk, cams_str[i], env_idxs[i], imgs_start[i:i + 1], preds_video[i])
            frames_out.append(frames)
            acts_out.append(acts)
            self.cnt_explore_suc += int(ok)
            self.cnt_explo_suc_per_tk[tk] += int(ok)
            self.cnt_vid_rollouts += 1
            self.cnt_vid_rout_per_tk[tk] += 1
        self.num_steps_in_env += runner.env_steps
        return frames_out, acts_out

    # ------------------------------------------------------------------------------------------------------------ misc
    def to_batch_dict(self, imgs_start, imgs_goal, acts_gt, goal_embed=None):
        assert imgs_start.ndim == 4
        batch = dict(obs={'img_obs_1': imgs_start[:, None, ...], 'img_goal_1': imgs_goal[:, None, ...]})
        if acts_gt is not None:
            assert acts_gt.ndim == 3 and acts_gt.shape[-1] == 7
            batch['action'] = acts_gt
        return batch

    def make_wandb_dict_per_tk(self):
        m = {}
        for tk in self.cnt_vid_rout_per_tk:
            m[f'explo/{tk}-cnt_vid_rollouts'] = self.cnt_vid_rout_per_tk[tk]
            m[f'explo/{tk}-cnt_explore_suc_vsR'] = self.cnt_explo_suc_per_tk[tk]
        return m
