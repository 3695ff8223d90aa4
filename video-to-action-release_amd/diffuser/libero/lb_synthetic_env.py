"""A deterministic stand-in for `LiberoEnvList_V3` (environment/libero/lb_env_v3.py) -- the MuJoCo/robosuite simulator is outside
the hot path ("simulator pieces stubbed by a synthetic env for benchmarks").  It implements exactly the
methods the trainer and the eval harness call on `env_list` (lb_online_trainer_v7.py:418-431, 673-716, 859-1122;
lb_eval_helper.py:84-373), over a toy kinematic world:

  * state = end-effector position (3) + gripper opening (1) + one movable object (3) per task; a 7-DoF *delta* action moves the
    end effector by `0.02 * act[:3]`, act[6] > 0 closes the gripper; a closed gripper within reach carries the object;
  * the task is done when the object lies within 3 cm of the task's target;
  * `render_*` rasterises the scene with numpy into an [H,W,3] uint8 image (table gradient, target ring, object disc, gripper
    cross) -- enough structure for the policy to have something to learn, cheap enough (tens of microseconds) to stay off the
    profile; the gripper depth camera returns the height above whatever lies under the gripper.
"""
import copy
import numpy as np

LB_TASKS_65to72 = {       # task names and LIBERO-90 indices of the released 8-task split (lb_constants.py:2-11)
    'put the red mug on the left plate': 65,
    'put the red mug on the right plate': 66,
    'put the white mug on the left plate': 67,
    'put the yellow and white mug on the right plate': 68,
    'put the chocolate pudding to the left of the plate': 69,
    'put the chocolate pudding to the right of the plate': 70,
    'put the red mug on the plate': 71,
    'put the white mug on the plate': 72,
}

_TABLE_Z = 0.42
_EE_REST_Z = 0.74           # z_diff = ee_z - depth == 0.32 with nothing under the gripper (lb_online_trainer_v7.py:1174)


class _ToyEnv:
    """One task instance.  `step(action) -> (obs, reward, done, info)` like the robosuite wrapper the reference steps."""

    def __init__(self, task_idx, seed, image_hw):
        self.task_idx = task_idx
        self.H, self.W = image_hw
        self.seed = seed
        self._did_see_sim_exception = False
        self.reset(seed)

    def reset(self, seed=None):
        if seed is not None:
            self.seed = seed
        rng = np.random.RandomState((self.task_idx * 7919 + self.seed * 104729) % (2 ** 31))
        self.ee = np.array([0.0, 0.0, _EE_REST_Z]) + rng.uniform(-0.02, 0.02, 3)
        self.grip = 1.0                                     # 1 open .. 0 closed
        self.obj = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.2, 0.2), _TABLE_Z + 0.04])
        self.target = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.2, 0.2), _TABLE_Z + 0.04])
        hue = rng.uniform(0.2, 1.0, 3)
        self.obj_rgb = (255 * hue / hue.max()).astype(np.uint8)
        self.holding = False
        self.t = 0
        return self.get_obs()

    def get_state(self):
        return np.concatenate([self.ee, [self.grip], self.obj, [float(self.holding)]])

    def get_obs(self):
        return {"robot0_eef_pos": self.ee.copy(), "robot0_gripper_qpos": np.array([self.grip, -self.grip]) * 0.04}

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        assert action.shape == (7,)
        self.ee = self.ee + 0.02 * np.clip(action[:3], -1, 1)
        self.ee[0:2] = np.clip(self.ee[0:2], -0.4, 0.4)
        self.ee[2] = np.clip(self.ee[2], _TABLE_Z + 0.02, 1.0)
        self.grip = float(np.clip(self.grip - 0.25 * np.sign(action[6]), 0.0, 1.0))
        near = np.linalg.norm(self.ee - self.obj) < 0.06
        if self.grip <= 0.25 and near:
            self.holding = True
        if self.grip > 0.5 and self.holding:
            self.holding = False
            self.obj[2] = _TABLE_Z + 0.04
        if self.holding:
            self.obj = self.ee.copy()
        self.t += 1
        done = bool(np.linalg.norm(self.obj[:2] - self.target[:2]) < 0.03 and not self.holding)
        return self.get_obs(), float(done), done, {}

    # ------------------------------------------------------------------ rendering
    def _to_px(self, xy):
        return (self.W * (0.5 + xy[0] / 0.9), self.H * (0.5 - xy[1] / 0.9))

    def render(self, cam_name="agent"):
        H, W = self.H, self.W
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        img = np.empty((H, W, 3), np.float32)
        img[..., 0] = 90 + 60 * xx / W
        img[..., 1] = 70 + 50 * yy / H
        img[..., 2] = 60 + 0.2 * (self.task_idx % 8) * 100
        tx, ty = self._to_px(self.target)
        ring = np.abs(np.hypot(xx - tx, yy - ty) - 0.07 * W) < 0.012 * W
        img[ring] = (235, 235, 235)
        ox, oy = self._to_px(self.obj)
        r_obj = (0.045 + 0.05 * (self.obj[2] - _TABLE_Z)) * W
        img[np.hypot(xx - ox, yy - oy) < r_obj] = self.obj_rgb
        ex, ey = self._to_px(self.ee)
        arm = (0.02 + 0.03 * self.grip) * W
        cross = ((np.abs(xx - ex) < 0.008 * W) & (np.abs(yy - ey) < arm)) | ((np.abs(yy - ey) < 0.008 * H) & (np.abs(xx - ex) < arm))
        shade = 40 + 160 * (self.ee[2] - _TABLE_Z) / (1.0 - _TABLE_Z)
        img[cross] = (shade, shade, 255 - shade)
        return np.clip(img, 0, 255).astype(np.uint8)

    def render_depth(self, cam_name="gripper"):
        """(rgb, depth[H,W,1]) from the wrist camera: metric height of the surface under the gripper (table or the object)."""
        H, W = self.H, self.W
        depth = np.full((H, W, 1), _TABLE_Z, np.float32)
        if np.linalg.norm(self.ee[:2] - self.obj[:2]) < 0.05 and not self.holding:
            h0, h1 = round(H * 0.70), round(H * 0.88)
            w0, w1 = round(W * 0.30), round(W * 0.70)
            depth[h0:h1, w0:w1] = self.obj[2] - 0.10        # an object under the fingers reads closer than the table
        return self.render("agent"), depth


class SyntheticLiberoEnvList:
    """`env_list` protocol of the trainer.  One environment per (task, env_idx); env_idx values are the task's seed set."""

    def __init__(self, task_list=None, image_hw=(128, 128), num_seed_per_task=1, name="synthetic-8tk-65to72"):
        self.name = name
        self.task_to_task_idx = dict(LB_TASKS_65to72) if task_list is None else {tk: 65 + i for i, tk in enumerate(task_list)}
        self.task_list = list(self.task_to_task_idx.keys())
        self.num_tasks = len(self.task_list)
        self.camera_list = ['agent']
        self.num_seed_per_task = num_seed_per_task
        self.image_hw = tuple(image_hw)
        self.seed_sets = {tk: list(range(num_seed_per_task)) for tk in self.task_list}
        self.env_init_states = {tk: {s: None for s in self.seed_sets[tk]} for tk in self.task_list}
        self._envs = {}
        for tk in self.task_list:                       # initial states, as the reference stores them per (task, seed)
            for s in self.seed_sets[tk]:
                self.env_init_states[tk][s] = _ToyEnv(self.task_to_task_idx[tk], s, self.image_hw).get_state()

    # -- lifecycle ---------------------------------------------------------------------------------------------------
    def check_no_envs_exist(self):
        assert len(self._envs) == 0, "an environment was left open"

    def init_1_given_env(self, tk_name, env_idx, is_rand=False, e_seed=None):
        seed = int(e_seed) if e_seed is not None else (int(np.random.randint(0, 2 ** 31 - 1)) if is_rand else int(env_idx))
        env = _ToyEnv(self.task_to_task_idx[tk_name], seed, self.image_hw)
        self._envs[(tk_name, int(env_idx))] = env
        return env

    def close_1_given_env(self, tk_name, env_idx):
        self._envs.pop((tk_name, int(env_idx)))

    def recreate_given_envs(self, tasks_str, env_idxs, is_rand=True):
        for tk, e in zip(tasks_str, env_idxs):
            self.init_1_given_env(tk, e, is_rand=is_rand)

    def get_an_env_ref(self, tk, env_idx):
        return self._envs[(tk, int(env_idx))]

    # -- stepping / observation --------------------------------------------------------------------------------------
    def step_an_env(self, tk, env_idx, action):
        return self._envs[(tk, int(env_idx))].step(action)

    def get_an_env_obs(self, tk, env_idx):
        return self._envs[(tk, int(env_idx))].get_obs()

    def get_an_env_state(self, tk, env_idx):
        return copy.deepcopy(self._envs[(tk, int(env_idx))].get_state())

    def render_an_env(self, tk, cam_name, env_idx):
        return self._envs[(tk, int(env_idx))].render(cam_name)

    def render_a_given_env(self, env, cam_name):
        return env.render(cam_name)

    def render_an_env_with_preproc(self, tk, cam_name, env_idx, imgs_preproc_fn):
        """-> tensor [3,H,W] in [0,1] (the reference applies the batch preprocessing to a 1-image batch and strips the axis)."""
        return imgs_preproc_fn(self.render_an_env(tk, cam_name, env_idx)[None])[0]

    def render_an_env_with_depth(self, tk, cam_name, env_idx):
        return self._envs[(tk, int(env_idx))].render_depth(cam_name)
