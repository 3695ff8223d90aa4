"""Video-guided rollout: the caller that sits between the video sampler, `predict_action` and the rollout replay buffer in the joint
loop (behaviour of lb_online_trainer_v7.py:995-1291, which SURVEY 2a marks as simulator-bound host control flow).

Only what the hot path's callers need is kept, as two small objects written for this repo:

  * `GraspTrigger` -- the wrist-depth test that decides when the hand stops following the video and grasps;
  * `RolloutRunner` -- follows the predicted frames: per frame `randint(*n_preds)` policy calls of `n_acts` actions each (gripper channel
    overridden: open until the trigger fired, closed after), and once, when the trigger fires, a scripted descend + close sequence.

Draw order on the host generators is the reference's (python `random.randint` per frame and for the descend length, `np.random.uniform`
for a per-task descend speed), because the replay sampler shares those generators and its index stream is a parity contract.
Frames leave as uint8 HWC -- the layout the HBM replay store keeps.
"""
import random
import numpy as np
import torch


class GraspTrigger:
    """Fires when the ground under the fingers (mean wrist-camera depth in a fixed window, rows 75-82 %, columns 35-65 %) is further
    than `z_gap` from the end-effector height while the end effector is below `z_ceiling`.  Fires at most once per rollout."""
    ROWS, COLS = (0.75, 0.82), (0.35, 0.65)

    def __init__(self, z_gap, z_ceiling):
        self.z_gap, self.z_ceiling = float(z_gap), float(z_ceiling)
        self.fired = False

    def __call__(self, depth, ee_pos):
        if self.fired:
            return False
        d = np.asarray(depth)
        if d.shape[:2] != (128, 128) or (d < 0).any():
            raise AssertionError("wrist depth map must be 128x128 and non-negative")
        h, w = d.shape[:2]
        r0, r1 = (round(h * f) for f in self.ROWS)
        c0, c1 = (round(w * f) for f in self.COLS)
        ee = np.asarray(ee_pos)
        if ee.shape != (3,):
            raise AssertionError("end-effector position must have 3 components")
        gap = abs(float(ee[2]) - float(d[r0:r1, c0:c1].mean()))
        self.fired = gap > self.z_gap and float(ee[2]) < self.z_ceiling
        return self.fired


class RolloutRunner:
    def __init__(self, env_list, predict, preproc, *, n_acts, n_preds, grip_force, descend_steps, close_steps, descend_speed,
                 descend_speed_per_task, close_descend_speed, z_gap, z_ceiling, wrist_cam='gripper', stop_at_success=False):
        """predict(img_start [1,3,H,W], img_goal [1,3,H,W]) -> clamped actions [n_acts, 7] (host tensor);
        preproc(uint8 [N,H,W,3]) -> float [N,3,H,W] in [0,1]."""
        self.env, self.predict, self.preproc = env_list, predict, preproc
        self.n_acts, self.n_preds, self.grip = n_acts, tuple(n_preds), float(grip_force)
        self.descend_steps, self.close_steps = tuple(descend_steps), int(close_steps)
        self.descend_speed, self.descend_speed_per_task = descend_speed, descend_speed_per_task
        self.close_descend_speed = float(close_descend_speed)
        self.z_gap, self.z_ceiling, self.wrist_cam, self.stop_at_success = z_gap, z_ceiling, wrist_cam, stop_at_success
        self.env_steps = 0

    def _scripted_grasp(self, tk):
        """[descend x k | close x close_steps] action rows."""
        k = random.randint(*self.descend_steps)
        if self.descend_speed is None:
            lo, hi = self.descend_speed_per_task[self.env.task_to_task_idx[tk]]
            vz = float(np.random.uniform(low=lo, high=hi, size=1).item())
        else:
            vz = float(self.descend_speed)
        if vz > 0:
            raise AssertionError("the descend speed must point down")
        down = torch.zeros(k, 7)
        down[:, 2] = vz
        close = torch.zeros(self.close_steps, 7)
        close[:, 2] = self.close_descend_speed
        close[:, 6] = self.grip
        return down, close

    def run(self, tk, cam, env_idx, img_start, goal_frames):
        """-> (frames uint8 [T+1,H,W,3], actions float [T,7], success)."""
        env = self.env
        frames = [env.render_an_env(tk, cam, env_idx)]
        chunks = []
        trigger = GraspTrigger(self.z_gap, self.z_ceiling)
        grasping = False
        success = False
        obs = img_start

        def apply(rows):
            done = False
            for a in rows:
                done = bool(env.step_an_env(tk, env_idx, a.numpy())[2])
                frames.append(env.render_an_env(tk, cam, env_idx))
            chunks.append(rows)
            return done

        for goal in goal_frames:
            for _ in range(random.randint(*self.n_preds)):
                with torch.no_grad():
                    act = self.predict(obs, goal[None])
                act[:, 6] = self.grip if grasping else -self.grip
                success = apply(act) or success           # the reference keeps the flag of the chunk's LAST env.step
                self.env_steps += len(act)
                obs = self.preproc(frames[-1][None])
                depth = env.render_an_env_with_depth(tk, self.wrist_cam, env_idx)[1]
                if trigger(depth, env.get_an_env_obs(tk, env_idx)['robot0_eef_pos']):
                    grasping = True
                    for rows in self._scripted_grasp(tk):
                        apply(rows)
                    obs = self.preproc(frames[-1][None])
            if success and self.stop_at_success:
                break
        acts = torch.cat(chunks).float()
        if len(frames) != len(acts) + 1:
            raise AssertionError("one frame per action plus the initial frame")
        return torch.from_numpy(np.stack(frames)), acts, success
