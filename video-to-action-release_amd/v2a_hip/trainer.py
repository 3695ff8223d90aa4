"""One policy train step on the MI355X, end to end, as a replayable hipGraph:

    replay indices (host, bit-exact stream)  ->  HIP gather from the HBM-resident store (R1-R3)
    -> Philox noise / timesteps (R6)  ->  HIP forward + backward into one flat gradient arena (R4-R8)
    -> [RCCL all-reduce of the arena over xGMI when world_size > 1]
    -> fused clip + AdamW + zero + EMA (R9)  ->  weight re-pack for the next step

Mirrors the order of lb_online_trainer_v7.py:558-624 (sample_from_bufs -> compute_loss -> backward -> clip_grad_norm_ ->
opt.step -> zero_grad -> ema.update).  Data parallel = one process per GPU, every rank holds a full replica (parameters, Adam
moments, EMA) and its own replay shard / RNG stream (seed + rank); the only collective is one sum all-reduce of the 87.2 M fp32
gradients per step (issued as two asynchronous slice all-reduces: the ConditionalUnet1D slice travels under the image-encoder
backward), averaged by folding 1/world into the optimiser's gradient scale.
"""
import copy
import numpy as np
import torch
from . import ops
from .optim import FusedAdamWEMA
from .replay import ReplayStore, sample_indices, count_uniform_below
from .dp import GradReducer, alloc_arena
from ._lib import lib, check


class PolicyTrainer:
    def __init__(self, policy, store: ReplayStore, batch_size=64, opt_params=None, ema_params=None, seed=0, use_graph=True,
                 process_group=None, world_size=1, rank=0, store_vid: ReplayStore = None, rand_prob=0.3, force_dp=False, dp_wire="fp32",
                 loss_scale_init=65536.0, fuse_packs=True, presum=True, dp_algo="rccl"):
        """`store_vid` (optional, same HBM pool as `store`: ReplayStore.pair) is the video-guided-rollout buffer; minibatches
        then follow sample_from_bufs' 'rand_prob' rule (lb_online_trainer_v7.py:787-851): all rows from `store` while `store_vid` is
        empty, otherwise n_rand = #(U[0,1) < rand_prob) rows from `store` first and the rest from `store_vid`.
        force_dp: the data-parallel step structure for a single rank too (exercises the RCCL path on a one-GPU box); dp_wire: "fp32" |
        "bf16" wire format of the gradient all-reduce; dp_algo: "rccl" | "direct" (peer-pointer exchange, v2a_hip/dp.py); loss_scale_init: initial dynamic loss scale of the fp16 mode (<= 0: no scaling);
        fuse_packs / presum: the optimiser's update kernel writes the forward conv operands / the ConditionalUnet1D slice's gradient-norm
        partial sums run ahead of the serial tail (both bit-equal to their separate-launch forms: tests/test_policy_gpu.py)."""
        self.policy = policy
        self.store = store
        self.store_vid = store_vid
        self.rand_prob = rand_prob
        if store_vid is not None:
            assert store_vid.root_frames.data_ptr() == store.root_frames.data_ptr(), "build the two stores with ReplayStore.pair"
        self.B = batch_size
        self.eng = policy.engine
        self.device = self.eng.device
        self.world, self.rank, self.pg = world_size, rank, process_group
        # the data-parallel step structure (three + one graphs, two asynchronous slice all-reduces); force_dp selects it for a single
        # rank too
        self.dp = world_size > 1 or (process_group is not None and force_dp)
        # data parallel: the deferred ConditionalUnet1D weight-gradient branch runs as a graph of its own on a side stream, the `model.*`
        # slice's all-reduce leaves from that stream when it is done, and the encoder backward runs on the main stream meanwhile
        self.dp_defer = self.dp
        if self.dp_defer:
            self.eng.split_deferred = True
        self._side = None
        self._g_wg = None
        opt_params = dict(lr=1e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6) if opt_params is None else dict(opt_params)
        ema_params = dict(update_after_step=0, inv_gamma=1.0, power=0.75, min_value=0.0, update_every=1) if ema_params is None else dict(ema_params)
        # ema_pytorch keeps the online model out of the EMA module tree (and out of its state_dict) when this is False -- the released
        # config's setting; the checkpoint writer (_EMAHandle.state_dict) honours it
        self.ema_include_online_model = bool(ema_params.pop("include_online_model", True))
        # EMA replica (ema_pytorch deep-copies the online model: lb_online_trainer_v7.py:135)
        self._ema_policy = copy.deepcopy(policy)
        self._ema_policy.requires_grad_(False)
        self.names = policy.trainable_names()
        P = dict(policy.named_parameters())
        EP = dict(self.ema_policy.named_parameters())
        total = sum(P[n].numel() for n in self.names)
        # data parallel: the arena is an allocation of its own (the unit the peer-pointer exchange maps into the other ranks)
        self.arena = alloc_arena(total, self.device) if self.dp else torch.zeros(total, dtype=torch.float32, device=self.device)
        gviews = self.eng.grad_views(self.arena, self.names)
        self.opt = FusedAdamWEMA([P[n].data for n in self.names], [gviews[n] for n in self.names], [EP[n].data for n in self.names],
                                 lr=opt_params["lr"], betas=tuple(opt_params["betas"]), eps=opt_params["eps"],
                                 weight_decay=opt_params["weight_decay"], max_norm=1.0, ema_inv_gamma=ema_params["inv_gamma"],
                                 ema_power=ema_params["power"], ema_min_value=ema_params["min_value"],
                                 ema_beta=ema_params.get("beta", 0.9999), ema_update_after_step=ema_params["update_after_step"],
                                 ema_update_every=ema_params["update_every"])
        # fp16 MFMA mode (v2a_hip.set_precision("fp16")): dynamic loss scaling inside the fused tail, GradScaler's contract -- the scaled loss
        # gradient starts the backward, unscale rides on the clip factor, a non-finite gradient norm skips the update and halves the scale
        import v2a_hip as _v
        self._loss_scale_init = float(loss_scale_init)
        self.loss_scaling = _v.get_precision() == "fp16" and self._loss_scale_init > 0
        if self.loss_scaling:
            self.eng.loss_scale_ptr = self.opt.enable_loss_scaling(self._loss_scale_init)
        self.seed = int(seed) + 7919 * rank
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)       # Philox offset, advanced on device
        T, Da = policy.horizon, policy.action_dim
        self.frame_start = torch.zeros(batch_size, dtype=torch.int64, device=self.device)
        # ring of pinned staging buffers: the H2D copy of step k may still be queued when the host prepares step k+1
        self._fs_ring = [torch.zeros(batch_size, dtype=torch.int64).pin_memory() for _ in range(8)]
        self._fs_evt = [None] * 8
        self.noise = torch.empty((batch_size, T, Da), dtype=torch.float32, device=self.device)
        self.timesteps = torch.empty(batch_size, dtype=torch.int64, device=self.device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.use_graph = use_graph
        self._g_fb = None
        self._g_enc = None
        self._g_opt = None
        self._slices = self.eng.arena_slices(self.names)
        # the one collective of the path (v2a_hip/dp.py): slice 0 = ConditionalUnet1D gradients (final after phase 1), slice 1 = encoders
        # dp_wire="bf16": opt-in bf16 wire format (half the bytes per xGMI link; the sum is rounded to bf16 -- not the parity path)
        self._dp_wire = dp_wire
        self.reducer = GradReducer(self.arena, self._slices, process_group, world_size, wire=dp_wire, algo=dp_algo) if self.dp else None
        self.feed = None               # parity / test hook (eager mode): dict(rows=int64[B] pool offsets, noise=[B,T,Da], timesteps=int64[B])
        self.on_grads_ready = None     # diagnostics hook (eager mode): called with the arena right before the optimiser consumes it
        self.comm_events = None        # bench: [(before_wait, after_wait)] HIP event pairs bracketing the stream's wait on the communicator
        self.phase_events = None       # bench (data-parallel step structure): [(e0 .. e4)] HIP events at the boundaries of the three graphs
        self._st = None
        self._warm = 0
        self.step_count = 0
        # the ConditionalUnet1D's transposed data-gradient packs (HBM-bound launches) leave the serial tail: they run at the start of the
        # NEXT step on a side stream, under the encoder forward
        self.fuse_packs = bool(fuse_packs)             # forward conv operands written by the optimiser's update kernel
        # gradient norm: the ConditionalUnet1D slice (one end of the arena, 75 % of the parameters) is summed on the deferred weight-gradient
        # stream while the encoder backward runs (single GPU only: under data parallelism the slices are all-reduced first; not with a
        # gradient hook, which may still edit the arena).  The range is handed to the optimiser step EXPLICITLY, and only by a step whose
        # own backward ran the pre-sum (_presummed): nothing is remembered across steps or optimisers.
        self._presum_range = (0, 0)
        self._presummed = False
        if presum and not self.dp:
            mi = [i for i, n in enumerate(self.names) if n.startswith("model.")]
            if mi and mi[-1] - mi[0] + 1 == len(mi):         # (the model.* group is contiguous in the arena, first or last)
                self._presum_range = self.opt.chunk_range(mi[0], mi[-1] + 1)
                self.eng.on_unet_wgrads_done = self._presum
        self._pack_serial = -1
        self._packs_fused = False
        self._pack_side = None
        self._wg_keep = None

    def _presum(self):
        """Hook on the deferred weight-gradient stream (PolicyEngine.on_unet_wgrads_done).  Only inside this trainer's own step (any other
        backward through the engine, e.g. policy.compute_loss(...).backward(), fires the hook too and must not arm anything)."""
        if self._in_step and self._presum_range[1] and self.on_grads_ready is None:
            self.opt.presum(*self._presum_range)
            self._presummed = True

    _in_step = False

    @property
    def ema_policy(self):
        """The EMA replica (ema_pytorch's ema_model)."""
        return self._ema_policy

    # ------------------------------------------------------------------ pieces
    def _draw_indices(self):
        sv = self.store_vid
        if self.feed is not None:                      # injected rows (parity tests): no RNG state is consumed
            if self.use_graph and self._warm >= 2:
                raise RuntimeError("PolicyTrainer.feed is an eager-mode hook (use_graph=False)")
            offs = np.asarray(self.feed["rows"], dtype=np.int64)
            assert offs.shape == (self.B,)
            ep = st = None
        elif sv is None or len(sv) == 0:
            ep, st = sample_indices(self.store.episode_lengths(), self.B, self.store.act_len)
            offs = self.store.pool_rows(ep, st)
        elif len(self.store) == 0:
            ep, st = sample_indices(sv.episode_lengths(), self.B, sv.act_len)
            offs = sv.pool_rows(ep, st)
        else:
            n_rand = count_uniform_below(self.B, self.rand_prob)
            if n_rand == 0 or n_rand == self.B:
                raise RuntimeError("stack expects a non-empty TensorList")     # the reference's torch.stack([]) on an empty draw
            e0, s0 = sample_indices(self.store.episode_lengths(), n_rand, self.store.act_len)
            e1, s1 = sample_indices(sv.episode_lengths(), self.B - n_rand, sv.act_len)
            offs = np.concatenate([self.store.pool_rows(e0, s0), sv.pool_rows(e1, s1)])
            ep, st = (e0, e1), (s0, s1)
        slot = self.step_count % len(self._fs_ring)
        if self._fs_evt[slot] is not None:
            self._fs_evt[slot].synchronize()
        self._fs_ring[slot].copy_(torch.from_numpy(np.asarray(offs, dtype=np.int64)))
        self.frame_start.copy_(self._fs_ring[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._fs_evt[slot] = ev
        return ep, st

    def _fwd_bwd(self):
        st = self.store
        B = self.B
        ops.tstamp_reset()
        ops.tstamp("step begin")
        self._presummed = False
        # last step's ConditionalUnet1D weights -> packed operands under the encoder forward (joined before the ConditionalUnet1D forward)
        if self._pack_side is None:
            self._pack_side = torch.cuda.Stream(device=self.device)
        self._pack_side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._pack_side):
            self.eng.refresh_packs("unet", skip_fwd=self._packs_fused)
        self.eng._pack_join = self._pack_side
        oo = torch.empty((2 * B, 3, st.H, st.W), dtype=torch.float32, device=self.device)      # start | goal frames, one allocation
        o0, o1 = oo[:B], oo[B:]
        oa = torch.empty((B, st.act_len, st.act_dim), dtype=torch.float32, device=self.device)
        check(lib.v2a_replay_gather(st.root_frames.data_ptr(), 1 if st.dtype == torch.uint8 else 0, st.root_acts.data_ptr(),
                                    self.frame_start.data_ptr(), o0.data_ptr(), o1.data_ptr(), oa.data_ptr(), B, st.H, st.W,
                                    st.act_len, st.act_dim, 0, 1, ops._stream()), "replay_gather")
        n_noise = self.noise.numel()
        if self.feed is not None:
            self.noise.copy_(self.feed["noise"].to(self.device, torch.float32))
            self.timesteps.copy_(self.feed["timesteps"].to(self.device, torch.int64))
        else:
            ops.philox_normal(self.noise, self.seed, offset_dev=self.counter)
            ops.philox_randint(self.timesteps, self.policy.noise_scheduler.config.num_train_timesteps, self.seed ^ 0x5DEECE66D,
                               offset_dev=self.counter)
            check(lib.v2a_advance_counter(self.counter.data_ptr(), (n_noise + 3) // 4 + B, ops._stream()), "advance_counter")
        imgs = {"img_obs_1": o0, "img_goal_1": o1}
        self._st = self.eng.backward_phase1(imgs, oa, self.noise, self.timesteps, names=self.names, arena=self.arena)
        ops.copy2d(self._st["loss"], self.loss, 1, 1, 1, 1)

    def set_dp_algo(self, algo):
        """Swap the gradient exchange ("rccl" | "direct") between two steps; collective (every rank calls it with the same value).  The
        exchange launches sit between the captured graphs, so nothing is re-captured."""
        if not self.dp:
            raise RuntimeError("not a data-parallel trainer")
        if algo == self.reducer.algo:
            return
        new = GradReducer(self.arena, self._slices, self.pg, self.world, wire=self._dp_wire, algo=algo)      # raises on every rank or on none
        self.reducer.close()
        self.reducer = new

    def _bwd_encoders(self):
        self.eng.backward_phase2(self._st)

    def _slice0(self, graph=False):
        """Start the all-reduce of slice 0 (`model.*`).  With the deferred weight-gradient branch: run that branch on the side stream
        first (eagerly or as its captured graph) and launch the collective FROM the side stream; the main stream goes on with the
        encoder backward and meets the side stream again in finish()."""
        if not self.dp_defer:
            self._reduce_async(0)
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            if graph:
                self._g_wg.replay()
            else:
                self._wg_keep = self.eng.run_deferred_wgrads()      # operands stay referenced until the join in _reduce_wait()
            self._reduce_async(0)

    def _reduce_async(self, which):
        """Sum all-reduce of one arena slice (RCCL over xGMI), asynchronous: the ConditionalUnet1D slice (74 % of the bytes) is
        final before the image-encoder backward starts and travels underneath it.  No fallback: a backend that cannot do this raises."""
        self.reducer.launch(which)

    def _reduce_wait(self):
        ev = None
        if self.comm_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.dp_defer and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)      # the weight-gradient branch (and the launch of slice 0) joins here
            if not (self.use_graph and self._g_fb is not None):
                self._wg_keep = None
        self.reducer.finish(self.opt.scale_grads)           # averaging folded into the optimiser's gradient scale
        if ev is not None:
            ev[1].record()
            self.comm_events.append(ev)

    def _opt(self):
        ops.tstamp("optimiser begin")
        # the update kernel also writes the forward conv operands (and their 16-bit twins in the 16-bit MFMA modes): no pack launch re-reads
        # the parameters; the transposed data-gradient packs keep their launch
        fused = self.fuse_packs
        if fused and self._pack_serial != self.eng._mp_serial:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("optimiser pack table missing during graph capture; run one eager step first")
            rows, self._pack_serial = self.eng.opt_pack_rows(self.opt.params)
            self.opt.set_pack_rows(rows)
        self._packs_fused = fused
        self.opt.step(zero_grad=True, packs=fused, presum=self._presum_range if self._presummed else (0, 0))
        self._presummed = False
        ops.tstamp("optimiser done / packs begin")
        self.eng.refresh_packs("enc", skip_fwd=fused)
        self.eng._packs_pending = "unet"               # (re-armed on the host after every replay, see step())
        ops.tstamp("step end")

    # ------------------------------------------------------------------ step
    def step(self):
        """One optimisation step.  Returns the device tensor holding the loss (read it with .item() only when needed)."""
        import v2a_hip as _v
        if (_v.get_precision() == "fp16") != self.loss_scaling and self._loss_scale_init > 0:
            raise RuntimeError("the precision mode changed after this PolicyTrainer was built (fp16 needs the loss scaler the constructor "
                               "sets up, the other modes must not carry one): build a new trainer after v2a_hip.set_precision()")
        self._draw_indices()
        self._in_step = True
        try:
            self._step_body()
        finally:
            self._in_step = False
        self.eng._packs_pending = "unet"               # any reader of a UNet operand outside the next step refreshes it first
        self.step_count += 1
        return self.loss

    def _step_body(self):
        if not self.use_graph or self._warm < 2:
            self._fwd_bwd()
            if self.dp:
                self._slice0()
            self._bwd_encoders()
            if self.dp:
                self._reduce_async(1)
                self._reduce_wait()
            if self.on_grads_ready is not None:
                self.on_grads_ready(self.arena)
            self._opt()
            self._warm += 1
        else:
            if self._g_fb is None:
                torch.cuda.synchronize()
                self._g_fb = torch.cuda.CUDAGraph()
                if not self.dp:                       # one graph: gather -> fwd -> bwd -> optimiser -> re-pack
                    with torch.cuda.graph(self._g_fb):
                        self._fwd_bwd()
                        self._bwd_encoders()
                        self._opt()
                else:                                     # three graphs with the two slice all-reduces launched between them
                    # (thread-local capture mode: the communicator's watchdog thread polls the events of earlier collectives while these
                    # graphs are captured -- in the default global mode its hipEventQuery is an illegal call and takes the process down)
                    with torch.cuda.graph(self._g_fb, capture_error_mode="thread_local"):
                        self._fwd_bwd()
                    if self.dp_defer:
                        # this graph runs NEXT TO the encoder-backward graph: a memory pool of its own, and its operands (activations
                        # of the first graph) stay referenced for the graphs' lifetime -- a block freed here would be handed to the
                        # graphs captured after it, which then overwrite it while this one still reads
                        self._g_wg = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self._g_wg, capture_error_mode="thread_local"):
                            self._wg_keep = self.eng.run_deferred_wgrads()
                    self._g_enc = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._g_enc, pool=self._g_fb.pool(), capture_error_mode="thread_local"):
                        self._bwd_encoders()
                    self._g_opt = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool(), capture_error_mode="thread_local"):
                        self._opt()
                # capture does not execute: run the step for real
            pe = None
            if self.dp and self.phase_events is not None:
                pe = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                pe[0].record()
            self._g_fb.replay()
            if self.dp:
                if pe:
                    pe[1].record()
                self._slice0(graph=True)
                self._g_enc.replay()
                if pe:
                    pe[2].record()
                self._reduce_async(1)
                self._reduce_wait()
                if pe:
                    pe[3].record()
                self._g_opt.replay()
                if pe:
                    pe[4].record()
                    self.phase_events.append(pe)

    def verify_exchange(self):
        """Data parallel over the direct exchange: wait for the last step's exchange launches and raise if one of them gave up on a peer
        (GradReducer.check(sync=True)) -- call it before writing a checkpoint or logging a loss, so that a step whose gradients were garbage
        is reported BEFORE its parameters are kept.  A no-op for a single rank and for the process-group all-reduce (which raises by itself)."""
        red = getattr(self, "reducer", None)
        if red is not None and red.algo == "direct":
            red.check(sync=True)

    def close(self):
        """Release the data-parallel resources (peer mappings, signal block, parked arenas).  Collective when data parallel."""
        red = getattr(self, "reducer", None)
        if red is not None:
            red.close()
        from .dp import drain_arenas
        drain_arenas()

    def ema_for_inference(self):
        """EMA weights are updated by the fused kernel behind torch's back: refresh its packed copies before use."""
        self.ema_policy.engine.refresh_packs()
        return self.ema_policy
