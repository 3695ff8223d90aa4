"""Latency path of the policy: `predict_action` (reference diffusion_unet_image_policy.py:88-201) captured
once into a hipGraph and replayed per control step -- two image encoders + N scheduler steps of ConditionalUnet1D + the fused
DDIM/DDPM update + unnormalise, with the observation, the initial trajectory noise and the per-step noises in static buffers.
The rollout loop of the reference calls this ~35x per sub-goal at B=1 (lb_online_trainer_v7.py:1060-1122)."""
import torch
from . import ops
from .policy_sched import ddim_coeffs, ddpm_coeffs, ddim_timesteps


class GraphedPredictAction:
    def __init__(self, policy, batch_size=1, use_ddim=True, seed=0, persistent=None):
        """persistent: run all scheduler steps of the ConditionalUnet1D as ONE persistent launch (v2a_hip/policy_persist.py,
        csrc/policy_persist.hip: plain fp32 FMA arithmetic, ~30 grid barriers per step instead of ~100 launches).  None = at batch 1 (the
        rollout loop's batch, where it is the faster path: 3.9 against 5.0 ms for the eight steps; at batch 2 it also runs but is no
        faster); False = the layer-by-layer kernels of the training path."""
        self.policy = policy
        self.eng = policy.engine
        dev = self.eng.device
        self.B, self.use_ddim = batch_size, use_ddim
        H, W = policy._cfg.image_hw
        self.obs = {k: torch.zeros((batch_size, 3, H, W), dtype=torch.float32, device=dev) for k in policy._cfg.rgb_keys}
        T, Da = policy.horizon, policy.action_dim
        self.Ttr = policy.noise_scheduler.config.num_train_timesteps
        self.steps = ddim_timesteps(self.Ttr, policy.num_inference_steps_ddim) if use_ddim else list(range(self.Ttr - 1, -1, -1))
        self.init = torch.zeros((batch_size, T, Da), dtype=torch.float32, device=dev)
        self.step_noise = None if use_ddim else torch.zeros((len(self.steps), batch_size, T, Da), dtype=torch.float32, device=dev)
        self.tt = [torch.full((batch_size,), t, dtype=torch.long, device=dev) for t in self.steps]
        self.seed, self.counter = seed, torch.zeros(1, dtype=torch.int64, device=dev)
        self.out = None
        self.graph = None
        from .policy_persist import PersistentDenoiser
        auto = persistent is None
        if auto:
            persistent = batch_size == 1
        self.pp = None
        self.pp_fallback = None                           # why the automatic choice fell back to the layer-by-layer kernels, if it did
        if persistent:
            try:
                # statistics area of the kernel: 2 floats per (sample, group), 64 floats in all
                if batch_size * self.eng.cfg.n_groups * 2 > 64:
                    raise ValueError(f"policy_persist: batch {batch_size} x {self.eng.cfg.n_groups} groups exceeds the kernel's statistics area")
                self.pp = PersistentDenoiser(self.eng, batch_size, self.steps, use_ddim, policy.num_inference_steps_ddim, self.init,
                                             self.step_noise)
            except (ValueError, RuntimeError) as e:
                # persistent=True is a demand: re-raise.  The automatic choice (None) covers the shapes the kernel was built for (horizon 16,
                # kernel sizes 5 / 3, layers that fit the LDS); any other config takes the working layer-by-layer path, and says so
                if not auto:
                    raise
                self.pp_fallback = f"{type(e).__name__}: {e}"
                print(f"[GraphedPredictAction] persistent denoiser not used ({self.pp_fallback}); layer-by-layer kernels instead", flush=True)

    def _run(self):
        eng, pol = self.eng, self.policy
        gc = eng.global_cond(self.obs)
        if self.pp is not None:
            self.out = self.pp.launch(gc)             # every scheduler step, the scheduler updates and the un-normalisation: one launch
            return
        traj = self.init
        for i, t in enumerate(self.steps):
            eps = eng.unet_fwd(traj, self.tt[i], gc)
            if self.use_ddim:
                traj = ops.policy_sched_step(eps, traj, None, ddim_coeffs(eng.ac_host, t, self.Ttr, pol.num_inference_steps_ddim), mode=1)
            else:
                traj = ops.policy_sched_step(eps, traj, self.step_noise[i] if t > 0 else None, ddpm_coeffs(eng.ac_host, t, self.Ttr), mode=0)
        self.out = ops.unnormalize_action(traj, eng.act_limits)

    def _draw(self):
        ops.philox_normal(self.init, self.seed, offset_dev=self.counter)
        n = (self.init.numel() + 3) // 4
        if self.step_noise is not None:
            ops.philox_normal(self.step_noise, self.seed ^ 0x9E3779B9, offset_dev=self.counter)
            n += (self.step_noise.numel() + 3) // 4
        from ._lib import lib, check
        check(lib.v2a_advance_counter(self.counter.data_ptr(), n, ops._stream()), "advance_counter")

    @torch.no_grad()
    def __call__(self, obs_dict, init_noise=None, step_noises=None):
        """obs_dict[key]: [B,To,3,H,W] in [0,1].  Noise is drawn on the device (Philox) unless given (parity runs)."""
        To = self.policy.n_obs_steps
        for k, buf in self.obs.items():
            buf.copy_(obs_dict[k][:, :To].reshape(buf.shape), non_blocking=True)
        if init_noise is None:
            self._draw()
        else:
            self.init.copy_(init_noise)
            if step_noises is not None and self.step_noise is not None:
                self.step_noise[:len(step_noises)].copy_(torch.stack(list(step_noises)))
        if self.graph is None:
            self._run()                                   # warm-up (packs, workspace)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, **ops.graph_capture_mode()):
                self._run()
        self.graph.replay()
        if self.pp is not None:
            # Under replay no host code of launch() runs, so the abandoned-barrier word is read HERE.  It is written by the kernel before it
            # terminates: synchronise first -- the caller is about to read the action on the host anyway (the rollout's .cpu()), and a garbage
            # action must not reach the simulator or the replay buffer
            torch.cuda.current_stream().synchronize()
            self.pp.check()
        start = To - 1
        return {"action": self.out[:, start:start + self.policy.n_action_steps], "action_pred": self.out}
