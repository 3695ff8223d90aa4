"""`Global_EnvReplayBuffer_Img` under the reference's import path, as a view over the HBM-resident `ReplayStore`.

Reference: diffuser/datasets/env_img_replay_buffer.py:10-116 (constructor keywords :12-16, `add_one_episode` :45-64,
`sample_random_batch_seq` :68-116), built by the trainer at lb_online_trainer_v7.py:208-213 with
`env_buf_config={'sample_act_seq_len': model_act_horizon}`.  The reference keeps a deque of per-episode deques of fp32 CPU tensors;
here every frame lives once in HBM (uint8), the (episode, window) draws consume numpy's / CPython's live generator states exactly as
the reference's `np.random.randint` + `random.randint` calls do (bit-exact stream: tests/test_replay.py), and the payload is one HIP
gather.  Returned tensors stay on the GPU (the reference returns CPU tensors and the trainer copies them over, :586).
"""
import torch
from v2a_hip.replay import ReplayStore


class Global_EnvReplayBuffer_Img(ReplayStore):
    def __init__(self, task_list, max_num_unitBufs, max_len_uB, min_len_uB, env_list, render_img_size, env_buf_config={}, *,
                 device="cuda:0", capacity_frames=None, dtype=torch.uint8, pool=None, pool_offset=0):
        assert max_num_unitBufs <= 1e4                                   # the reference's own bound (:41)
        act_len = int(env_buf_config['sample_act_seq_len'])              # KeyError when absent, like the reference (:31)
        hw = tuple(render_img_size) if render_img_size is not None else (128, 128)
        super().__init__(max_num_unitBufs, max_len_uB, min_len_uB, image_hw=hw, act_len=act_len, device=device,
                         capacity_frames=capacity_frames, dtype=dtype, pool=pool, pool_offset=pool_offset)
        self.task_list = task_list
        self.env_list = env_list
        self.camera_list = getattr(env_list, "camera_list", None)
        self.num_cams = len(self.camera_list) if self.camera_list is not None else 0
        self.render_img_size = render_img_size
        self.sample_act_seq_len = act_len
        self.per_sample_gap = 1
        self.max_num_unitBuf = max_num_unitBufs
        self.max_len_uB, self.min_len_uB = max_len_uB, min_len_uB

    # the reference exposes the per-episode metadata as parallel deques
    @property
    def buffers(self):
        return self.episodes

    @property
    def bufs_task(self):
        return [e[2] for e in self.episodes]

    @property
    def bufs_cam(self):
        return [e[3] for e in self.episodes]
