"""Host-side executor of the goal-conditioned diffusion policy on the HIP kernels (forward, backward, sampling).

Mirrors the reference's DiffusionUnetImagePolicy.compute_loss / predict_action
(diffuser/diffusion_policy/diffusion_unet_image_policy.py:88-277) as an explicit launch sequence:
  images NCHW -> NHWC + 2x-1 (normalizer.py:139-146)  -> 2 x ResNet18-GN + SpatialSoftmax + Linear
  (model/multi_image_obs_encoder.py:144-196, common/vision_nets.py, common/base_nets.py:234-285)
  -> add_noise -> ConditionalUnet1D (model/conditional_unet1d.py:178-246) -> MSE,
then the hand-written backward of every op in reverse.  Parameters stay ordinary torch tensors with the
reference's names; the engine only borrows their pointers and keeps packed copies of the conv weights.
Everything is launched on torch's current stream and allocates only through torch's caching allocator, so a whole
train step can be captured into one hipGraph (torch.cuda.graphs) and replayed.
"""
import math
import numpy as np
import torch
from . import ops


def _dgrad(x, cv, bias, cout, kh, kw, stride=(1, 1), pad=(0, 0), **kw_):
    """conv2d launch that uses cv's data-gradient operand (forward pack + N-major loader, or the flipped pack)."""
    w, bmode = cv.dg()
    x3 = cv.eng._planes_of(x)
    if x3 is not None:
        kw_ = dict(kw_, x_p3=x3.view((3,) + tuple(x.shape)))
    return ops.conv2d(x, w, bias, cout, kh, kw, stride, pad, bmode=bmode, **kw_)


def squaredcos_alphas_cumprod(n=100, max_beta=0.999):
    """diffusers' squaredcos_cap_v2 betas -> alphas_cumprod (fp32 cumprod), restated from the published algorithm."""
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = torch.tensor([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)], dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)



# scratch lanes (ops.ws_lane: launches that may run concurrently must not share split-K slabs): 0 = the main chain, 1 + i = camera
# encoder i on its side stream (any number of rgb keys), and two fixed lanes far outside that range
_LANE_ENC0 = 1
_LANE_RC = 1001        # slabs of a ConditionalResidualBlock1D's 1 x 1 residual conv, consumed by the block's second GroupNorm launch
_LANE_WG = 1002        # deferred ConditionalUnet1D weight gradients (their own stream)


class _Conv:
    """One conv / linear / transposed-conv parameter pair and its packed operands."""

    def __init__(self, eng, wname, bname, transposed=False):
        self.eng = eng
        self.wname, self.bname = wname, bname
        w = eng.P[wname]
        self.shape = tuple(w.shape)
        if w.dim() == 2:
            self.kh, self.kw = 1, 1
        elif w.dim() == 3:
            self.kh, self.kw = 1, w.shape[2]
        else:
            self.kh, self.kw = w.shape[2], w.shape[3]
        self.co, self.ci = w.shape[0], w.shape[1]     # torch dims 0 / 1 (for transposed: [Cin][Cout])
        # data gradients read the forward pack directly (N-major loader) when the reduction channels are a multiple of 16
        self.nmaj = (self.co % 16 == 0) and (self.ci % 4 == 0)
        self.group = "unet" if wname.startswith("model.") else "enc"      # refresh_packs(which)
        self._pf = None
        self._pd = None
        self._pf_h = None          # bf16 twins of the two operands (bf16 precision mode, see PolicyEngine.refresh_packs)
        self._pd_h = None
        # three-plane bf16 copies (hi / mid / lo) of the two operands: the ConditionalUnet1D convs whose inputs arrive pre-split run on the
        # pure LDS-DMA kernel (ops.conv2d_p3); written by the optimiser's update kernel / the pack launches (PolicyEngine.refresh_packs)
        self.want_p3 = False
        self._pf_p3 = None
        self._pd_p3 = None
        # RGB stem (7x7, 3 input channels): also the channel-window pack [Cout][7][8][4] of ops.conv2d_window (zero column / channel)
        self.window = w.dim() == 4 and self.kh == 7 and self.kw == 7 and self.ci == 3
        self._pw = None
        self._ver = (None, None)

    @property
    def w(self):
        return self.eng.P[self.wname]

    @property
    def b(self):
        return self.eng.P[self.bname] if self.bname is not None else None

    def _fresh(self):
        if self.eng._packs_pending is not None and self.group == self.eng._packs_pending:
            self.eng.refresh_packs(self.eng._packs_pending)      # the trainer left this group's packs to the start of its next step
        w = self.w
        key = (w.data_ptr(), w._version)
        if key != self._ver:
            self.repack()

    def repack(self):
        w = self.w.detach()
        if self.kh * self.kw == 1:
            self._pf = w                         # [Cout][Cin] is already the K-contiguous forward operand
        else:
            if self._pf is None or self._pf is w:
                self._pf = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
            ops.pack_weight(w, 0, self._pf)
        if self.window:
            if self._pw is None:
                self._pw = torch.zeros(self.co * self.kh * (self.kw + 1) * (self.ci + 1), dtype=torch.float32, device=w.device)
            ops.pack_weight(w, 2, self._pw)
        if not self.nmaj or self._pd is not None:     # flipped pack: odd channel counts always, every layer once the bf16 mode used it
            if self._pd is None:
                self._pd = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
            ops.pack_weight(w, 1, self._pd)
        if self._pf_h is not None:                    # keep the bf16 twins in step with the fp32 operands
            from ._lib import lib, check
            for src, dst in ((self._pf, self._pf_h), (self._pd, self._pd_h)):
                check(lib.v2a_cast_f32_h(src.data_ptr(), dst.data_ptr(), src.numel() - src.numel() % 4, 1 if dst.dtype == torch.float16 else 0,
                                         ops._stream()), "cast_f32_h")
                if src.numel() % 4:
                    dst[-(src.numel() % 4):] = src.reshape(-1)[-(src.numel() % 4):].to(dst.dtype)
            ops.register_h_twin(self._pf, self._pf_h)
            ops.register_h_twin(self._pd, self._pd_h)
        if self._pf_p3 is not None:                   # ... and the three-plane copies
            ops.split3(self._pf.reshape(-1), out=self._pf_p3)
            ops.register_p3(self._pf, self._pf_p3)
            if self._pd is not None and self._pd_p3 is not None:
                ops.split3(self._pd, out=self._pd_p3)
                ops.register_p3(self._pd, self._pd_p3)
        self._ver = (self.w.data_ptr(), self.w._version)

    def pf(self):
        self._fresh()
        return self._pf

    def pw(self):
        self._fresh()
        return self._pw

    def dg(self):
        """(operand, bmode) for the data-gradient / transposed-conv launch of this weight."""
        self._fresh()
        from ._lib import lib
        if self.nmaj and lib.v2a_get_precision() == 0 and not self.eng.flip_dgrad:
            return self._pf, 1
        if self._pd is None:                      # bf16 mode: the K-contiguous flipped pack keeps data gradients on the bf16 kernel
            self._pd = torch.empty(self.w.numel(), dtype=torch.float32, device=self.w.device)
            ops.pack_weight(self.w.detach(), 1, self._pd)
        return self._pd, 0


class PolicyEngine:
    def __init__(self, cfg, params: dict):
        """cfg: object with the PolicyCfg fields (see diffuser/diffusion_policy shell); params: name -> tensor (cuda fp32)."""
        self.cfg = cfg
        self.P = params
        self.device = next(iter(params.values())).device
        self.ac_host = squaredcos_alphas_cumprod(cfg.num_train_timesteps)
        self.ac = self.ac_host.to(self.device)
        # action limits of the policy's normaliser (None = the Libero -1 / +1, where normalise is the identity up to rounding)
        lim = getattr(cfg, "act_limits", None)
        self.act_limits = None if lim is None else tuple(torch.as_tensor(v, dtype=torch.float32).reshape(-1).contiguous().to(self.device)
                                                        for v in lim)
        self._convs = {}
        # the two camera encoders (separate weights, no shared state) run as parallel branches of the step graph: +17 % steps/s
        # at B=64 (their small-grid kernels fill each other's idle CUs); their weight gradients stay on their chain's stream
        self.enc_streams = True    # (bench.py's instrumented pass sets False: every kernel alone on one stream)
        self._enc_side = []
        self._in_enc = False
        self._cur_batch = 0
        self._stem_pad = {}
        self._stem_buf = {}        # (camera, N, H, W) -> zero-bordered [N, H + 6, W + 6, 4] stem input (_stem_fwd)
        self._stem_ver = {}        # (camera, N, H, W) -> forwards written into that buffer so far
        # ConditionalUnet1D weight gradients feed nothing until the optimiser: with defer_unet_wgrad they are collected during the
        # data-gradient chain and launched as ONE extra branch next to the two encoder backward chains (one fork / one join).
        # The data-parallel trainer turns this off: there the `model.*` arena slice must be final after phase 1 so that its
        # all-reduce can travel under the encoder backward.
        self.defer_unet_wgrad = True
        # data gradients are ordinary convs over a flipped, K-contiguous pack (written by the transposing multi-pack launch): the
        # K-contiguous form is what the LDS-DMA kernels take
        self.flip_dgrad = True
        # Pre-split operands for the ConditionalUnet1D convs: built, bit-equal to the register-splitting path
        # (tests/test_policy_gpu.py::test_presplit_operand_convs_equal_the_register_split_path_bitwise runs both) and measured SLOWER in the
        # step -- 8.00 / 8.13 ms against 7.77 / 7.83 ms, alternating runs on one box -- although every conv launch alone is 1.1 ... 1.3 x
        # faster (tools/probes/r5/conv_p3_probe.py, weights warm in the 256-MB Infinity Cache): in the step the 65 M weights come from HBM
        # on every pass, and three bf16 planes are 6 bytes per element where the fp32 pack is 4 -- the optimiser and the transposing pack
        # launch write 0.78 GB more, forward and data-gradient convs read 0.26 GB more, per step.  fp32 weights split in the kernel ARE
        # the byte-optimal operand format for these weight-streaming GEMMs.  Default off; DESIGN.md section 3 "Round 5".
        self.use_p3 = False
        self._p3 = {}                # data_ptr of an fp32 activation / gradient of the ConditionalUnet1D -> its three bf16 planes (this pass)
        self._p3_active = False      # inside unet_fwd / unet_bwd: the GroupNorm launches also write the planes of their outputs
        self._deferred = []
        self.split_deferred = False  # data parallel: backward_phase2 leaves the deferred weight gradients to run_deferred_wgrads()
        # debug export for the parity tests (None: off): when a dict, encode_fwd records the forward's DISCRETE decisions of each camera
        # encoder -- the ReLU masks (NHWC bool) and the max-pool winners (int8 window tap 0..8) -- under "<module prefix>.relu0 / .pool /
        # <block prefix>.relu1 / .relu2", so that a reference backward can be routed through the same decisions (tests/test_policy_gpu.py)
        self.debug_decisions = None
        self.loss_scale_ptr = 0      # device address of the dynamic loss scale (fp16 mode: PolicyTrainer points it at its optimiser state)
        # GroupNorm parameter gradients: every layer's backward leaves its per-sample column sums in a persistent [N,2,C] buffer; ONE
        # multi-tensor launch per chain (ConditionalUnet1D, each camera encoder) reduces them over n in a fixed order -- no atomics,
        # bitwise reproducible, and 60 reduction launches per step fewer
        self._gn_cs = {}
        self._gn_chain = None
        self._gn_tables = {}
        self._gn_pinned = set()
        # weight gradients: split-K reduces postponed to one multi-tensor launch per chain (ops.WgradCollector)
        self._wgc = None
        self._wgc_on = True
        self._collect_wg = False
        self._wg_stream = None
        # grouped weight-gradient launches: one group per camera encoder at the end of its chain (and per <= 16 ConditionalUnet1D layers)
        self._build()

    # ------------------------------------------------------------------ structure
    def conv(self, wname, bname=None, transposed=False):
        c = self._convs.get(wname)
        if c is None:
            c = _Conv(self, wname, bname, transposed)
            self._convs[wname] = c
        return c

    def _build(self):
        cfg = self.cfg
        self.enc = {}
        for key in cfg.rgb_keys:
            pre = f"obs_encoder.key_model_map.{key}"
            bb = pre + ".backbone.nets"
            blocks = []
            cin = cfg.widths[0]
            for li, c in enumerate(cfg.widths):
                for bi in range(2):
                    stride = 2 if (li > 0 and bi == 0) else 1
                    bp = f"{bb}.{4 + li}.{bi}"
                    blk = dict(pre=bp, stride=stride, cin=cin, cout=c, conv1=self.conv(bp + ".conv1.weight"),
                               conv2=self.conv(bp + ".conv2.weight"), down=None)
                    if (bp + ".downsample.0.weight") in self.P:
                        blk["down"] = self.conv(bp + ".downsample.0.weight")
                    blocks.append(blk)
                    cin = c
            self.enc[key] = dict(pre=pre, bb=bb, conv1=self.conv(bb + ".0.weight"), blocks=blocks,
                                 pool=self.conv(pre + ".pool.nets.weight", pre + ".pool.nets.bias"),
                                 fc=self.conv(pre + ".nets.3.weight", pre + ".nets.3.bias"))
        m = "model."
        self.step1 = self.conv(m + "diffusion_step_encoder.1.weight", m + "diffusion_step_encoder.1.bias")
        self.step3 = self.conv(m + "diffusion_step_encoder.3.weight", m + "diffusion_step_encoder.3.bias")
        dims = [cfg.action_dim] + list(cfg.down_dims)
        n = len(cfg.down_dims)

        def rb(pre, cin, cout):
            d = dict(pre=pre, cin=cin, cout=cout, c0=self.conv(pre + ".blocks.0.block.0.weight", pre + ".blocks.0.block.0.bias"),
                     c1=self.conv(pre + ".blocks.1.block.0.weight", pre + ".blocks.1.block.0.bias"),
                     ce=self.conv(pre + ".cond_encoder.1.weight", pre + ".cond_encoder.1.bias"), rc=None)
            if cin != cout:
                d["rc"] = self.conv(pre + ".residual_conv.weight", pre + ".residual_conv.bias")
            return d

        self.down = []
        for i in range(n):
            lvl = dict(r0=rb(f"{m}down_modules.{i}.0", dims[i], dims[i + 1]), r1=rb(f"{m}down_modules.{i}.1", dims[i + 1], dims[i + 1]),
                       ds=None)
            if i < n - 1:
                lvl["ds"] = self.conv(f"{m}down_modules.{i}.2.conv.weight", f"{m}down_modules.{i}.2.conv.bias")
            self.down.append(lvl)
        mid = dims[-1]
        self.mid = [rb(f"{m}mid_modules.{i}", mid, mid) for i in range(2)]
        self.up = []
        in_out = list(zip(dims[:-1], dims[1:]))
        for i, (din, dout) in enumerate(reversed(in_out[1:])):
            self.up.append(dict(r0=rb(f"{m}up_modules.{i}.0", dout * 2, din), r1=rb(f"{m}up_modules.{i}.1", din, din),
                                us=self.conv(f"{m}up_modules.{i}.2.conv.weight", f"{m}up_modules.{i}.2.conv.bias", transposed=True)))
        self.fin0 = self.conv(m + "final_conv.0.block.0.weight", m + "final_conv.0.block.0.bias")
        self.fin1 = self.conv(m + "final_conv.1.weight", m + "final_conv.1.bias")
        # convs of the ConditionalUnet1D whose operands can arrive pre-split (channel counts in 32N, 64N output columns both ways)
        for blk in [b for lvl in self.down for b in (lvl["r0"], lvl["r1"])] + list(self.mid) + [b for lvl in self.up for b in (lvl["r0"], lvl["r1"])]:
            for cv in (blk["c0"], blk["c1"], blk["rc"]):
                if cv is not None and cv.ci % 64 == 0 and cv.co % 64 == 0:
                    cv.want_p3 = True
        for lvl in self.down:
            if lvl["ds"] is not None:
                lvl["ds"].want_p3 = True
        self.fin0.want_p3 = True
        # the 16 FiLM projections (cond_encoder = Mish -> Linear(G, 2*C), one per residual block) all read the same Mish(cond):
        # they run as ONE GEMM over concatenated weights (forward, weight gradient, data gradient), see unet_fwd / unet_bwd
        self.film = []
        for lvl in self.down:
            self.film += [lvl["r0"], lvl["r1"]]
        self.film += list(self.mid)
        for lvl in self.up:
            self.film += [lvl["r0"], lvl["r1"]]
        off = 0
        for r in self.film:
            r["film_off"] = off
            off += 2 * r["cout"]
        self.film_nf = off
        self.film_gd = self.film[0]["ce"].ci
        self._film_w = self._film_b = self._film_wd = None
        self._film_ver = None
        self._dfilm_all = None
        self.batch_film = True

    def _twin_dy(self, dy, x_h, cout):
        """bf16 twin of an output gradient, made once when the twin-fed weight-gradient kernel will take the layer (bf16-MFMA mode,
        the forward conv left a twin of its input, wide enough output); None otherwise (the convs then round on their own)."""
        if x_h is None or cout < 64 or ops.lib.v2a_get_precision() != 1:
            return None
        tw = self._take_tw(dy)                       # left by the GroupNorm backward that produced dy (same storage: no cast launch)
        if tw is not None:
            return tw.view(dy.shape)
        return ops.cast_h(dy, ops.POLICY_HALF[0])

    _tw = None
    _tw_tag = 0

    def _set_tw(self, twin, of):
        """Remember `twin` as the bf16 copy of the fp32 tensor `of` (tagged with its address: a stale twin is never handed out)."""
        self._tw, self._tw_tag = twin, (of.data_ptr() if twin is not None else 0)

    def _take_tw(self, of):
        tw = self._tw
        if tw is None or self._tw_tag != of.data_ptr() or tw.numel() != of.numel():
            return None
        self._tw, self._tw_tag = None, 0
        return tw

    def _p3_mode(self):
        """Pre-split operands are used in the fp32 three-plane conv mode only (the 16-bit MFMA modes have their twins, the exact mode no planes)."""
        return self.use_p3 and ops.lib.v2a_get_precision() == 0 and ops.lib.v2a_get_f32_conv_mode() == 1

    def _planes_of(self, t):
        """The three bf16 planes a GroupNorm launch of this pass wrote next to the fp32 tensor `t`, or None."""
        if t is None or not self._p3:
            return None
        ent = self._p3.get(t.data_ptr())
        if ent is None or ent[1] != t.numel():
            return None
        return ent[0]

    # ------------------------------------------------------------------ weight gradients off the critical path
    def _wg(self, *a, **k):
        """Weight gradients feed nothing until the optimiser: launch them on a side stream so they fill the CUs the latency-bound
        data-gradient chain leaves idle (captured as a parallel branch of the hipGraph).  Operands are kept alive until the join.
        immediate=True: the caller reads the gradient right away (no postponed split-K reduce)."""
        k = dict(k)
        immediate = k.pop("immediate", False)
        if self.defer_unet_wgrad and self._collect_wg:
            self._deferred.append((a, k))
            return None
        if self._wgb is not None and not immediate:
            kb = dict(k, slab_key=k.get("slab_key") or self._slab_key(k.get("dw")))
            if self._wgb.add(*a, **kb):
                self._wgb_keep.append((a, k))
                return None
        if self._wgc_active and not immediate:
            k = dict(k, collector=self._collector(), slab_key=k.get("slab_key") or self._slab_key(k.get("dw")))
        return ops.conv2d_wgrad(*a, **k)

    def _collector(self):
        if self._wgc is None:
            self._wgc = ops.WgradCollector(self.device)
        return self._wgc

    _wgc_active = False
    _dw_names = None
    _tmp_seq = 0
    _wgb = None            # the ops.WgradBatch weight gradients are being collected into (grouped launches), or None
    _wgb_keep = ()

    def _wgb_begin(self):
        """Collect the weight gradients that follow into grouped launches."""
        if not self._wgc_active:
            return
        self._wgb = ops.WgradBatch(self._collector())
        self._wgb_keep = []

    def _wgb_end(self):
        """Launch what was collected (operands were kept alive until here)."""
        if self._wgb is None:
            return
        if len(self._wgb):
            self._wgb.launch()
        self._wgb, self._wgb_keep = None, ()

    def _slab_key(self, dw):
        """Stable name of the layer a gradient view belongs to (its slab buffer is kept per layer, not per address: the autograd path
        hands in a fresh arena on every call)."""
        if dw is None or self._dw_names is None:
            return None
        name = self._dw_names.get(dw.data_ptr())
        if name is None:             # a temporary: two of equal shape in one un-flushed chain must not share a slab
            self._tmp_seq += 1
            return ("tmp", tuple(dw.shape), self._tmp_seq)
        return name

    def _wg_begin(self):
        """From here on weight gradients only run their main kernels; _wg_flush sums all their split slabs in one launch."""
        self._wgc_active = self._wgc_on
        self._tmp_seq = 0

    def _wg_flush(self):
        if self._wgc_active:
            self._collector().flush()
        self._wgc_active = False

    def _film_key(self):
        return tuple((r["ce"].w.data_ptr(), r["ce"].w._version, r["ce"].b._version) for r in self.film)

    def _film_rows(self):
        """Multi-pack table rows that (re)build the concatenated FiLM operands: W_cat [NF][G] and b_cat [NF] by plain copies
        (mode 0, one tap), W_cat^T [G][NF] (the data-gradient operand) by the transposing launch."""
        dev = self.device
        if self._film_w is None:
            self._film_w = torch.empty((self.film_nf, self.film_gd), dtype=torch.float32, device=dev)
            self._film_b = torch.empty(self.film_nf, dtype=torch.float32, device=dev)
            self._film_wd = torch.empty(self.film_nf * self.film_gd, dtype=torch.float32, device=dev)
        rows0, rows1 = [], []
        for r in self.film:
            w, b = r["ce"].w.detach(), r["ce"].b.detach()
            o, n2 = r["film_off"], 2 * r["cout"]
            rows0.append([w.data_ptr(), self._film_w.data_ptr() + 4 * o * self.film_gd, n2, self.film_gd, 1, 0, 0])
            rows0.append([b.data_ptr(), self._film_b.data_ptr() + 4 * o, 1, n2, 1, 0, 0])
        rows1.append([self._film_w.data_ptr(), self._film_wd.data_ptr(), self.film_nf, self.film_gd, 1, 1, 0])
        return rows0, rows1

    def _refresh_film(self):
        """Standalone rebuild (parameters changed outside refresh_packs, e.g. the autograd / compute_loss path)."""
        from ._lib import lib, check
        rows0, rows1 = self._film_rows()
        ce = lib.v2a_pack_chunk_elems()
        ch0 = [[i, s0] for i, r in enumerate(rows0) for s0 in range(0, r[2] * r[3], ce)]
        ntile = -(-self.film_nf // 64) * -(-self.film_gd // 64)
        ch1 = [[0, t] for t in range(ntile)]
        t0 = torch.tensor(rows0, dtype=torch.int64).to(self.device)
        t1 = torch.tensor(rows1, dtype=torch.int64).to(self.device)
        c0 = torch.tensor(ch0, dtype=torch.int32).to(self.device)
        c1 = torch.tensor(ch1, dtype=torch.int32).to(self.device)
        check(lib.v2a_pack_weights_multi(t0.data_ptr(), c0.data_ptr(), len(ch0), 0, ops._stream()), "pack_weights_multi")
        check(lib.v2a_pack_weights_multi(t1.data_ptr(), c1.data_ptr(), len(ch1), 1, ops._stream()), "pack_weights_multi_t")
        self._film_keep = (t0, t1, c0, c1)              # keep the tables alive until the launches ran
        self._film_ver = self._film_key()

    on_unet_wgrads_done = None     # optional callback, runs on the deferred weight-gradient stream right after those launches
    _packs_pending = None      # "unet": the trainer postponed that group's re-pack to the start of its next step (see refresh_packs)
    _pack_join = None          # stream the UNet forward has to wait for (the side stream that group's re-pack was launched on)

    _mp_serial = 0             # bumped whenever the pack tables are rebuilt (operand buffers may have moved): opt_pack_rows is stale then

    def opt_pack_rows(self, tensors):
        """For FusedAdamWEMA.set_pack_rows: per optimiser tensor (matched by address) the forward operand the update kernel should also
        write -- the [Cout][taps][Cin] pack of a conv with more than one tap (+ the stem's channel-window pack), the slice of the
        concatenated FiLM operand of a FiLM weight / bias -- or zeros; in the 16-bit MFMA modes also the 16-bit twin of the forward operand.
        Returns (rows, serial)."""
        self.refresh_packs()                          # allocates every operand; fixes the addresses the rows point at
        from ._lib import lib
        bf16 = lib.v2a_get_precision() == 1
        f16 = 1 if (bf16 and ops.POLICY_HALF[0] is torch.float16) else 0
        by_ptr = {}
        for c in self._convs.values():
            taps = c.kh * c.kw
            tw = c._pf_h.data_ptr() if (bf16 and c._pf_h is not None) else 0
            if not bf16 and c._pf_p3 is not None and self._p3_mode():      # side copy = the three bf16 planes of the forward operand (format 2)
                tw = c._pf_p3.data_ptr()
                fmt = 2
                if taps > 1:
                    by_ptr[c.w.data_ptr()] = (c._pf.data_ptr(), c.ci, taps, 0, tw, fmt)
                else:
                    by_ptr[c.w.data_ptr()] = (0, c.w.numel(), 1, 0, tw, fmt)
                continue
            if taps > 1:
                by_ptr[c.w.data_ptr()] = (c._pf.data_ptr(), c.ci, taps, c._pw.data_ptr() if c.window else 0, tw, f16)
            elif tw:                                    # 1 x 1 / linear weights are their own fp32 operand: twin only
                by_ptr[c.w.data_ptr()] = (0, c.w.numel(), 1, 0, tw, f16)
        if self.batch_film:
            for r in self.film:
                w, b = r["ce"].w.detach(), r["ce"].b.detach()
                o = r["film_off"]
                tw = by_ptr.get(w.data_ptr(), (0, 0, 0, 0, 0, 0))[4]     # (its own 1 x 1 twin, if the 16-bit mode keeps one: same plain-copy indexing)
                by_ptr[w.data_ptr()] = (self._film_w.data_ptr() + 4 * o * self.film_gd, w.numel(), 1, 0, tw, f16)
                by_ptr[b.data_ptr()] = (self._film_b.data_ptr() + 4 * o, b.numel(), 1, 0, 0, 0)
        return [by_ptr.get(t.data_ptr(), (0, 1, 1, 0, 0, 0)) for t in tensors], self._mp_serial

    def refresh_packs(self, which="all", skip_fwd=False):
        """Unconditionally re-pack conv weights (call once per train step after the optimiser; capturable).  which = "all" | "enc" |
        "unet": the image encoders' (`obs_encoder.*`) or the ConditionalUnet1D's (`model.*`, 75 % of the parameters) operands only --
        the trainer packs the encoders right after the optimiser and the UNet at the START of the next step on a side stream, under
        the encoder forward that does not read them (0.33 of the 0.45 ms of pack launches leave the serial tail); until then
        `_packs_pending` makes any other reader of a UNet operand refresh it first.
        Two multi-tensor launches per group write everything: the forward packs (gather within a filter) and the flipped
        data-gradient packs (tap-reversed transposes through LDS; odd-channel layers always, every layer in the bf16 precision mode).
        In that mode both launches also write the bf16 twin of each operand, which ops.conv2d picks up for the layers the LDS-DMA
        bf16 kernel can take."""
        from ._lib import lib, check
        bf16 = lib.v2a_get_precision() == 1
        p3 = self._p3_mode()
        if getattr(self, "_mp", None) is not None and (self._mp["bf16"] != bf16 or self._mp.get("half") is not ops.POLICY_HALF[0]
                                                       or self._mp.get("p3") != p3):
            self._mp = None                            # precision mode, 16-bit format or fp32 conv mode changed: new twins / planes
        if getattr(self, "_mp", None) is None:
            rows = []
            ch0, ch1 = {"enc": [], "unet": []}, {"enc": [], "unet": []}
            ce = lib.v2a_pack_chunk_elems()
            for c in self._convs.values():
                w = c.w.detach()
                taps = c.kh * c.kw
                if bf16 and (c._pf_h is None or c._pf_h.dtype is not ops.POLICY_HALF[0]):
                    c._pf_h = torch.empty(w.numel(), dtype=ops.POLICY_HALF[0], device=self.device)
                    c._pd_h = torch.empty(w.numel(), dtype=ops.POLICY_HALF[0], device=self.device)
                if taps == 1:
                    c._pf = w
                elif c._pf is None or c._pf.data_ptr() == w.data_ptr():
                    c._pf = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
                cp3 = p3 and c.want_p3 and not bf16
                if cp3 and c._pf_p3 is None:
                    c._pf_p3 = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=self.device)
                    c._pd_p3 = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=self.device)
                if not cp3:
                    c._pf_p3 = c._pd_p3 = None
                if taps > 1 or bf16 or cp3:            # forward operand (1x1 weights are their own fp32 operand: twin / planes only)
                    rows.append([w.data_ptr(), c._pf.data_ptr() if taps > 1 else 0, c.co, c.ci, taps, 256 if cp3 else 0,
                                 c._pf_p3.data_ptr() if cp3 else (c._pf_h.data_ptr() if bf16 else 0)])
                    ch0[c.group] += [[len(rows) - 1, s0] for s0 in range(0, w.numel(), ce)]
                if c.window:                           # RGB stem: the channel-window pack next to the plain one (mode 2 of the same launch)
                    if c._pw is None:
                        c._pw = torch.zeros(c.co * c.kh * (c.kw + 1) * (c.ci + 1), dtype=torch.float32, device=self.device)
                    rows.append([w.data_ptr(), c._pw.data_ptr(), c.co, c.ci, taps, 2, 0])
                    ch0[c.group] += [[len(rows) - 1, s0] for s0 in range(0, w.numel(), ce)]
                if not c.nmaj or bf16 or self.flip_dgrad:
                    if c._pd is None:
                        c._pd = torch.empty(w.numel(), dtype=torch.float32, device=self.device)
                    rows.append([w.data_ptr(), c._pd.data_ptr(), c.co, c.ci, taps, 1 | (256 if cp3 else 0),
                                 c._pd_p3.data_ptr() if cp3 else (c._pd_h.data_ptr() if bf16 else 0)])
                    ntile = -(-c.co // 64) * -(-(c.ci * taps) // 64)
                    ch1[c.group] += [[len(rows) - 1, t] for t in range(ntile)]
                if bf16:
                    ops.register_h_twin(c._pf, c._pf_h)
                    ops.register_h_twin(c._pd, c._pd_h)
                if cp3:
                    ops.register_p3(c._pf, c._pf_p3)
                    ops.register_p3(c._pd, c._pd_p3)
            if self.batch_film:                        # (FiLM tables: ConditionalUnet1D operands)
                f0, f1 = self._film_rows()
                for row in f0:
                    rows.append(row)
                    ch0["unet"] += [[len(rows) - 1, s0] for s0 in range(0, row[2] * row[3], ce)]
                for row in f1:
                    rows.append(row)
                    ch1["unet"] += [[len(rows) - 1, t_] for t_ in range(-(-row[2] // 64) * -(-row[3] // 64))]
            dev = self.device
            t = lambda a, dt: torch.tensor(a, dtype=dt).to(dev) if a else None
            self._mp_serial += 1
            self._mp = dict(tab=t(rows, torch.int64), ptrs=[c.w.data_ptr() for c in self._convs.values()], bf16=bf16, half=ops.POLICY_HALF[0], p3=p3,
                            ch0={g: t(v, torch.int32) for g, v in ch0.items()}, n0={g: len(v) for g, v in ch0.items()},
                            ch1={g: t(v, torch.int32) for g, v in ch1.items()}, n1={g: len(v) for g, v in ch1.items()})
        mp = self._mp
        if mp["ptrs"] != [c.w.data_ptr() for c in self._convs.values()]:      # parameters were re-allocated (.to(), load): rebuild
            self._mp = None
            return self.refresh_packs(which, skip_fwd)
        groups = ("enc", "unet") if which == "all" else (which,)
        for g in groups:
            # skip_fwd: the optimiser's update kernel wrote the forward packs itself (FusedAdamWEMA.step(packs=True))
            if mp["n0"][g] and not skip_fwd:
                check(lib.v2a_pack_weights_multi(mp["tab"].data_ptr(), mp["ch0"][g].data_ptr(), mp["n0"][g], 0, ops._stream()), "pack_weights_multi")
            if mp["n1"][g]:
                check(lib.v2a_pack_weights_multi(mp["tab"].data_ptr(), mp["ch1"][g].data_ptr(), mp["n1"][g], 1, ops._stream()), "pack_weights_multi_t")
        if self._packs_pending in groups:
            self._packs_pending = None
        for c in self._convs.values():
            if c.group in groups:
                c._ver = (c.w.data_ptr(), c.w._version)
        if self.batch_film and "unet" in groups:
            self._film_ver = self._film_key()

    # ------------------------------------------------------------------ encoder
    def _gn(self, x4, pre, G, act, residual=None, film=None, slabs=None, post=None, post_slabs=None):
        """slabs: x4 is the not-yet-reduced output of a conv launched with defer=True (ops.Slabs); the GroupNorm launch finishes it."""
        N = x4.shape[0]
        C = x4.shape[-1]
        x3 = x4.view(N, -1, C)
        r3 = residual.view(N, -1, C) if residual is not None else None
        tw = [] if (C % 64 == 0 and ops.lib.v2a_get_precision() == 1) else None      # bf16-MFMA mode: the consuming conv's operand twin
        pl = [] if (self._p3_active and tw is None and C % 64 == 0) else None        # fp32 three-plane mode: its pre-split operand
        y, mean, rstd = ops.groupnorm_fwd(x3, self.P[pre + ".weight"], self.P[pre + ".bias"], G, act, residual=r3, film=film, slabs=slabs,
                                          twin_out=tw, post=None if post is None else post.view(N, -1, C), post_slabs=post_slabs,
                                          planes_out=pl)
        if pl:      # (the entry keeps y alive until the pass ends: in an inference pass nothing else does, and a freed address would be handed
            self._p3[y.data_ptr()] = (pl[0], y.numel(), y)      # to some other tensor that must not inherit these planes)
        self._set_tw(tw[0] if tw else None, y)       # taken by the caller right after (x_h of the next conv): no cast launch
        return y.view(x4.shape), (x3, mean, rstd, r3, film, pre, G, act)

    def _defer_ok(self, rows, C, G):
        """May a conv whose [N, rows, C] output feeds GroupNorm(G) directly leave its split-K reduce to that launch?"""
        return ops.gn_takes_slabs(rows, C, G)

    def _gn_bwd(self, saved, dout4, grads, want_dres=False, want_dfilm=False, dfilm_out=None, dslabs=None, keep_dout=False, want_twin=False):
        """dslabs: dout4 is the not-yet-reduced output of a data-gradient conv (ops.Slabs); keep_dout: other launches read dout4 later,
        so the GroupNorm launch also stores the finished sum into it."""
        x3, mean, rstd, r3, film, pre, G, act = saved
        d3 = dout4.view(x3.shape)
        N, _, C = x3.shape
        kw = dict(dout_slabs=dslabs, dout_sum=d3 if (dslabs is not None and keep_dout) else None)
        tw = [] if (want_twin and C % 64 == 0 and ops.lib.v2a_get_precision() == 1) else None
        kw["twin_out"] = tw
        pl = [] if (self._p3_active and tw is None and C % 64 == 0) else None
        kw["planes_out"] = pl
        chain = self._gn_chain
        if chain is None:            # outside a chain (stand-alone use): reduce this layer's parameter gradients right away
            dx, _, _, dres, dfilm = ops.groupnorm_bwd(x3, self.P[pre + ".weight"], self.P[pre + ".bias"], G, d3, mean, rstd, act,
                                                      residual=r3, film=film, want_dres=want_dres, want_dfilm=want_dfilm,
                                                      dgamma=grads[pre + ".weight"], dbeta=grads[pre + ".bias"], dfilm_out=dfilm_out, **kw)
        else:
            cs = self._gn_cs.get((pre, N, C))
            if cs is None:
                cs = torch.empty((N, 2, C), dtype=torch.float32, device=self.device)
                self._gn_cs[(pre, N, C)] = cs
            chain.append((pre, N, C))
            dx, _, _, dres, dfilm = ops.groupnorm_bwd(x3, self.P[pre + ".weight"], self.P[pre + ".bias"], G, d3, mean, rstd, act,
                                                      residual=r3, film=film, want_dres=want_dres, want_dfilm=want_dfilm,
                                                      dfilm_out=dfilm_out, colsum=cs, defer_params=True, **kw)
        if pl:
            self._p3[dx.data_ptr()] = (pl[0], dx.numel(), dx)
        self._set_tw(tw[0] if tw else None, dx)      # bf16 twin of dx (want_twin): operand of the data / weight gradients that follow
        return dx.view(dout4.shape), (dres.view(dout4.shape) if dres is not None else None), dfilm

    def _gn_begin(self):
        """Start collecting GroupNorm layers of one backward chain (returns the token _gn_flush takes)."""
        prev, self._gn_chain = self._gn_chain, []
        return prev

    def _gn_flush(self, prev, grads):
        """dgamma / dbeta of every GroupNorm layer the chain visited, in one launch.  The pointer table is cached per (layer list,
        destination addresses): the captured train step re-uses the table built by its eager warm-up steps."""
        chain, self._gn_chain = self._gn_chain, prev
        if not chain:
            return
        key = (tuple(chain), tuple(grads[ent[0] + ".weight"].data_ptr() for ent in chain))
        ent = self._gn_tables.get(key)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("GroupNorm gradient table missing during graph capture; run one eager step first")
            rows, work = [], []
            for i, ent in enumerate(chain):
                pre, N, C = ent          # the layer's own [N, 2, C] column-sum buffer
                rows.append([self._gn_cs[(pre, N, C)].data_ptr(), grads[pre + ".weight"].data_ptr(), grads[pre + ".bias"].data_ptr(), N, C])
                work += [[i, b] for b in range((C + 63) // 64)]
            for k in list(self._gn_tables):              # evict oldest-first, never a table a captured hipGraph points at
                if len(self._gn_tables) <= 64:
                    break
                if k not in self._gn_pinned:
                    del self._gn_tables[k]
            ent = (torch.tensor(rows, dtype=torch.int64).to(self.device), torch.tensor(work, dtype=torch.int32).to(self.device), len(work))
        else:
            del self._gn_tables[key]                     # re-insert: most recently used last
        self._gn_tables[key] = ent
        if torch.cuda.is_current_stream_capturing():
            self._gn_pinned.add(key)
        ops.gn_param_grads_multi(*ent)

    def _stem_fwd(self, key, img_nchw, conv1, w0):
        """RGB stem conv 7x7 / stride 2 / pad 3 (torchvision resnet18.conv1 behind vision_nets.py:29-39).  fp32 three-plane mode: the image
        goes into a persistent zero-bordered [N, H + 6, W + 6, 4] buffer and the conv runs as a channel-window conv on the vector loader
        (ops.conv2d_window: one aligned 128-B line per output pixel and filter row); the same buffer is the weight gradient's input.
        Otherwise: [N, H, W, 3] and the scalar-gather kernel.  Returns (saved input, conv output)."""
        N, C, H, W = img_nchw.shape
        # (also in the 16-bit MFMA modes: the 3-channel stem never ran on the 16-bit kernels, it is an fp32 conv there too)
        if (conv1.window and C == 3 and H % 2 == 0 and W % 2 == 0 and ops.lib.v2a_get_f32_conv_mode() == 1):
            bk = (key, N, H, W)
            xp = self._stem_buf.get(bk)
            if xp is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("stem input buffer missing during graph capture; run one eager step first")
                xp = self._stem_buf[bk] = torch.zeros((N, H + 6, W + 6, 4), dtype=torch.float32, device=img_nchw.device)
            self._stem_ver[bk] = self._stem_ver.get(bk, 0) + 1        # (a backward over an overwritten buffer is refused, see _encode_bwd)
            ops.nchw_to_nhwc4p(img_nchw, xp, 3, normalize=True)
            c1 = ops.conv2d_window(xp, conv1.pw(), w0, 7, 1, (2, 1), (H // 2, W // 2), xpitch=8, C=32)
            return xp, c1
        x0 = ops.nchw_to_nhwc(img_nchw, normalize=True)
        return x0, ops.conv2d(x0, conv1.pf(), None, w0, 7, 7, (2, 2), (3, 3))

    def encode_fwd(self, key, img_nchw, save):
        """img [B,3,H,W] in [0,1] (float or uint8) -> feature [B, feature_dim].  save: list collecting backward state (or None)."""
        e = self.enc[key]
        cfg = self.cfg
        w0 = cfg.widths[0]
        ops.tstamp(f"enc_fwd[{key}] begin")
        x0, c1 = self._stem_fwd(key, img_nchw, e["conv1"], w0)
        a1, s_gn1 = self._gn(c1, e["bb"] + ".1", w0 // 16, "relu")
        h, pidx = ops.maxpool_fwd(a1)
        ops.tstamp_fine(f"enc_fwd[{key}] stem done")
        dec = self.debug_decisions
        if dec is not None:
            dec[e["bb"] + ".relu0"] = a1 > 0
            dec[e["bb"] + ".pool"] = pidx.clone()
        st = dict(x0=x0, x0_ver=self._stem_ver.get((key,) + tuple(img_nchw.shape[:1]) + tuple(img_nchw.shape[2:]), 0), gn1=s_gn1, pidx=pidx, a1_shape=tuple(a1.shape), blocks=[])
        h_tw = None                                  # bf16 twin of the running activation (emitted by the GroupNorm launches)
        for blk in e["blocks"]:
            s, co = blk["stride"], blk["cout"]
            g = co // 16
            inp = h
            k1, k2 = [], []                          # bf16 twins of the conv inputs (bf16-MFMA mode): reused by the weight gradients
            # every conv is followed at once by the GroupNorm that consumes it: on the deep stages that launch also sums the conv's
            # split-K slabs (no reduce launch of its own), so nothing else may touch the scratch lane in between
            oh, ow = inp.shape[1] // s, inp.shape[2] // s
            dfr = self._defer_ok(oh * ow, co, g)
            o1, sl = ops.conv2d(inp, blk["conv1"].pf(), None, co, 3, 3, (s, s), (1, 1), keep_h=k1, x_h=h_tw, defer=dfr) if dfr else \
                (ops.conv2d(inp, blk["conv1"].pf(), None, co, 3, 3, (s, s), (1, 1), keep_h=k1, x_h=h_tw), None)
            a, s1 = self._gn(o1, blk["pre"] + ".bn1", g, "relu", slabs=sl)
            a_tw = self._take_tw(a)
            sd = None
            if blk["down"] is not None:
                idn, sl = ops.conv2d(inp, blk["down"].pf(), None, co, 1, 1, (s, s), (0, 0), x_h=k1[0] if k1 else None, defer=dfr) if dfr else \
                    (ops.conv2d(inp, blk["down"].pf(), None, co, 1, 1, (s, s), (0, 0), x_h=k1[0] if k1 else None), None)
                idn, sd = self._gn(idn, blk["pre"] + ".downsample.1", g, "none", slabs=sl)
            else:
                idn = inp
            o2, sl = ops.conv2d(a, blk["conv2"].pf(), None, co, 3, 3, (1, 1), (1, 1), keep_h=k2, x_h=a_tw, defer=dfr) if dfr else \
                (ops.conv2d(a, blk["conv2"].pf(), None, co, 3, 3, (1, 1), (1, 1), keep_h=k2, x_h=a_tw), None)
            h, s2 = self._gn(o2, blk["pre"] + ".bn2", g, "relu", residual=idn, slabs=sl)
            h_tw = self._take_tw(h)
            if dec is not None:
                dec[blk["pre"] + ".relu1"] = a > 0
                dec[blk["pre"] + ".relu2"] = h > 0
            st["blocks"].append(dict(inp=inp, a=a, s1=s1, s2=s2, sd=sd, inp_h=k1[0] if k1 else None, a_h=k2[0] if k2 else None))
            ops.tstamp_fine(f"enc_fwd[{key}] block {len(st['blocks']) - 1} done")
        feat = h
        B, FH, FW, FC = feat.shape
        kl = ops.conv2d(feat, e["pool"].pf(), e["pool"].b, cfg.num_kp, 1, 1)
        kp, att = ops.spatial_softmax_fwd(kl)
        f = ops.linear(kp, e["fc"].pf(), e["fc"].b)
        st.update(feat=feat, kp=kp, att=att)
        if save is not None:
            save[key] = st
        ops.tstamp(f"enc_fwd[{key}] end")
        return f

    def encode_bwd(self, key, df, st, grads):
        tok = self._gn_begin()
        self._wg_begin()
        self._wgb_begin()
        ops.tstamp(f"enc_bwd[{key}] begin")
        try:
            self._encode_bwd(key, df, st, grads)
        finally:
            ops.tstamp(f"enc_bwd[{key}] chain done")
            self._wgb_end()
            self._gn_flush(tok, grads)
            self._wg_flush()
            ops.tstamp(f"enc_bwd[{key}] end")

    def _encode_bwd(self, key, df, st, grads):
        e = self.enc[key]
        cfg = self.cfg
        B = df.shape[0]
        fc, pool = e["fc"], e["pool"]
        self._wg(st["kp"].view(1, 1, B, -1), df.view(1, 1, B, -1), fc.shape, 1, 1, dw=grads[fc.wname], dbias=grads[fc.bname])
        dkp = _dgrad(df.view(1, 1, B, -1), fc, None, fc.ci, 1, 1, (1, 1), (0, 0)).view(B, -1)
        dkl = ops.spatial_softmax_bwd(st["att"], st["kp"], dkp)
        feat = st["feat"]
        self._wg(feat, dkl, pool.shape, 1, 1, dw=grads[pool.wname], dbias=grads[pool.bname])
        dh = _dgrad(dkl, pool, None, pool.ci, 1, 1)
        ops.tstamp_fine(f"enc_bwd[{key}] head done")
        dh_sl = None
        nblk = len(e["blocks"])
        for bi, (blk, bs) in enumerate(zip(reversed(e["blocks"]), reversed(st["blocks"]))):
            s, co, ci = blk["stride"], blk["cout"], blk["cin"]
            inp = bs["inp"]
            g = co // 16
            dfr = self._defer_ok(bs["a"].shape[1] * bs["a"].shape[2], co, g)              # consumer: this block's bn1
            # consumer of this block's input gradient: bn2 of the block before (none for the first block: maxpool)
            dfr_in = bi + 1 < nblk and self._defer_ok(inp.shape[1] * inp.shape[2], ci, ci // 16)
            do2, didn, _ = self._gn_bwd(bs["s2"], dh, grads, want_dres=True, dslabs=dh_sl, want_twin=bs["a_h"] is not None and co >= 64)
            th = self._twin_dy                       # one bf16 rounding of a gradient serves its data and weight gradient
            do2h = th(do2, bs["a_h"], co)
            self._wg(bs["a"], do2, blk["conv2"].shape, 3, 3, (1, 1), (1, 1), dw=grads[blk["conv2"].wname], x_h=bs["a_h"], dy_h=do2h)
            da, sl = _dgrad(do2, blk["conv2"], None, co, 3, 3, (1, 1), (1, 1), x_h=do2h, defer=True) if dfr else \
                (_dgrad(do2, blk["conv2"], None, co, 3, 3, (1, 1), (1, 1), x_h=do2h), None)
            do1, _, _ = self._gn_bwd(bs["s1"], da, grads, dslabs=sl, want_twin=bs["inp_h"] is not None and co >= 64)
            do1h = th(do1, bs["inp_h"], co)
            self._wg(inp, do1, blk["conv1"].shape, 3, 3, (s, s), (1, 1), dw=grads[blk["conv1"].wname], x_h=bs["inp_h"], dy_h=do1h)
            ih, iw = inp.shape[1], inp.shape[2]
            if blk["down"] is not None:
                didn_raw, _, _ = self._gn_bwd(bs["sd"], didn, grads)
                ddh = th(didn_raw, bs["inp_h"], co)
                self._wg(inp, didn_raw, blk["down"].shape, 1, 1, (s, s), (0, 0), dw=grads[blk["down"].wname], x_h=bs["inp_h"], dy_h=ddh)
                d1 = _dgrad(didn_raw, blk["down"], None, ci, 1, 1, (1, 1), (0, 0), idil=s, out_hw=(ih, iw), x_h=ddh)
                res_in = d1
            else:
                res_in = didn
            if dfr_in:
                dh, dh_sl = _dgrad(do1, blk["conv1"], None, ci, 3, 3, (1, 1), (1, 1), idil=s, out_hw=(ih, iw), residual=res_in, x_h=do1h, defer=True)
            else:
                dh, dh_sl = _dgrad(do1, blk["conv1"], None, ci, 3, 3, (1, 1), (1, 1), idil=s, out_hw=(ih, iw), residual=res_in, x_h=do1h), None
            if bi % 2 == 1:
                ops.tstamp(f"enc_bwd[{key}] stage {3 - bi // 2} dgrad done")
            else:
                ops.tstamp_fine(f"enc_bwd[{key}] block {nblk - 1 - bi} dgrad done")
        da1 = ops.maxpool_bwd(dh, st["pidx"], st["a1_shape"])
        dc1, _, _ = self._gn_bwd(st["gn1"], da1, grads)
        ops.tstamp_fine(f"enc_bwd[{key}] stem gn done")
        # RGB stem: 3 input channels make every 16-B piece of the gathered operand straddle pixels (scalar-gather kernel, 0.4 ms
        # per encoder).  Pad the saved input to 4 channels once (zeros in the 4th), take the gradient of the 4-channel filter on
        # the vector / LDS-DMA kernel and keep its first three input channels.
        x0 = st["x0"]
        c1 = e["conv1"]
        if x0.shape[-1] == 4:                        # zero-bordered 4-channel stem input (_stem_fwd): no padding, no copy
            if st.get("x0_ver") != self._stem_ver.get((key, x0.shape[0], x0.shape[1] - 6, x0.shape[2] - 6)):
                raise RuntimeError("the stem input of this forward was overwritten by a later forward of the same encoder and batch shape")
            dw4 = torch.empty((c1.co, 4, 7, 7), dtype=torch.float32, device=x0.device)
            self._wg(x0, dc1, (c1.co, 4, 7, 7), 7, 7, (2, 2), (0, 0), dw=dw4, immediate=True)
            ops.copy2d(dw4, grads[c1.wname], c1.co, 3 * 49, 4 * 49, 3 * 49)
        elif x0.shape[-1] == 3:
            N0, H0, W0, _ = x0.shape
            xp = self._stem_pad.get(key)
            if xp is None or xp.shape[:3] != x0.shape[:3]:
                xp = torch.zeros((N0, H0, W0, 4), dtype=torch.float32, device=x0.device)
                self._stem_pad[key] = xp
            ops.copy2d(x0, xp, N0 * H0 * W0, 3, 3, 4)
            dw4 = torch.empty((c1.co, 4, 7, 7), dtype=torch.float32, device=x0.device)
            # its result is repacked right below: finished at once, not with the chain's postponed reduces
            self._wg(xp, dc1, (c1.co, 4, 7, 7), 7, 7, (2, 2), (3, 3), dw=dw4, immediate=True)
            ops.copy2d(dw4, grads[c1.wname], c1.co, 3 * 49, 4 * 49, 3 * 49)
        else:
            self._wg(x0, dc1, c1.shape, 7, 7, (2, 2), (3, 3), dw=grads[c1.wname])

    # ------------------------------------------------------------------ ConditionalUnet1D
    def _c1d(self, x, cv, k, x2=None, residual=None, stride=1, pad=None, keep_h=None, defer=False, x_h=None):
        """Conv1d on [B,T,C] (channels-last) via the (1 x k) view.  defer: returns (y, ops.Slabs | None), see ops.conv2d."""
        B, T, C = x.shape
        pad = k // 2 if pad is None else pad
        x3, x23 = self._planes_of(x), self._planes_of(x2)
        if x3 is not None and (x2 is None or x23 is not None):      # both sources arrive pre-split: the pure LDS-DMA kernel (ops.conv2d)
            x3 = x3.view(3, B, 1, T, C)
            x23 = None if x2 is None else x23.view(3, B, 1, T, -1)
        else:
            x3 = x23 = None
        y = ops.conv2d(x.view(B, 1, T, C), cv.pf(), cv.b, cv.co, 1, k, (1, stride), (0, pad),
                       x2=None if x2 is None else x2.view(B, 1, T, -1),
                       residual=None if residual is None else residual.view(B, 1, -1, cv.co), keep_h=keep_h, defer=defer, x_h=x_h,
                       x_p3=x3, x2_p3=x23)
        if defer:
            return y[0].view(B, -1, cv.co), y[1]
        return y.view(B, -1, cv.co)

    def _res_fwd(self, r, x, mgf, save, x2=None):
        cfg = self.cfg
        k, G = cfg.kernel_size, cfg.n_groups
        B, T, _ = x.shape
        co = r["cout"]
        kx = []
        if self._film_all is not None:
            film = self._film_all[:, r["film_off"]:r["film_off"] + 2 * co]         # column slice of the batched projection
        else:
            film = ops.linear(mgf, r["ce"].pf(), r["ce"].b)                       # [B, 2*co] == [B][2][co]
        dfr = self._defer_ok(T, co, G)        # the GroupNorm launch right behind each conv also sums its split-K slabs
        c0, sl = self._c1d(x, r["c0"], k, x2=x2, keep_h=kx, defer=True) if dfr else (self._c1d(x, r["c0"], k, x2=x2, keep_h=kx), None)
        a0, s0 = self._gn(c0.view(B, 1, T, co), r["pre"] + ".blocks.0.block.1", G, "mish", film=film, slabs=sl)
        a0 = a0.view(B, T, co)
        a0_tw = self._take_tw(a0)
        ka = []
        # the residual branch rides on the second GroupNorm launch (out = mish(gn(c1)) + residual_conv(x), conditional_unet1d.py:62-65): the
        # identity branch as a dense addend, the 1 x 1 conv branch as its split-K slabs left on a scratch lane of their own -- no add / reduce
        # launch on the serial chain (bit-equal: the kernel adds in the reduce kernel's order).  fp32 mode, float4 wave GroupNorm only.
        fuse = ops.lib.v2a_get_precision() == 0 and ops.gn_takes_post(T, co, G)
        p_dense, p_slabs = None, None
        if fuse:
            if r["rc"] is None:
                p_dense = x
            else:
                with ops.ws_lane(_LANE_RC):
                    rcy, p_slabs = self._c1d(x, r["rc"], 1, x2=x2, pad=0, defer=True)
                if p_slabs is None:
                    p_dense = rcy                                  # (the plan did not split: a finished tensor)
        c1, sl = self._c1d(a0, r["c1"], k, keep_h=ka, defer=True, x_h=a0_tw) if dfr else (self._c1d(a0, r["c1"], k, keep_h=ka, x_h=a0_tw), None)
        a1, s1 = self._gn(c1.view(B, 1, T, co), r["pre"] + ".blocks.1.block.1", G, "mish", slabs=sl,
                          post=None if p_dense is None else p_dense.reshape(B, 1, T, co), post_slabs=p_slabs)
        a1 = a1.view(B, T, co)
        x_h = kx[0] if kx else None                              # the twins also serve rc's and both weight gradients
        x2_h = kx[1] if (len(kx) > 1 and x2 is not None) else None
        if x2 is not None and x2_h is None:
            x_h = None
        if fuse:
            out = a1
        elif r["rc"] is not None:
            out = self._c1d(x, r["rc"], 1, x2=x2, residual=a1, pad=0)
        else:
            out = ops.axpy(a1, x)
        if save is not None:
            save.append(dict(r=r, x=x, x2=x2, a0=a0, s0=s0, s1=s1, a0_h=ka[0] if ka else None, x_h=x_h, x2_h=x2_h))
        ops.tstamp_fine(f"unet_fwd {r['pre'][6:]} done")
        return out

    def _res_bwd(self, st, dout, grads, dmgf, extra=None, need_dx=True, dslabs=None, defer_dx=False):
        """Returns (dx, dx2, dmgf).  dx2 only for two-source (concat) inputs.  `extra` is added to dx.
        dslabs: `dout` still is the split-K slabs of the conv that produced it (ops.Slabs): the first GroupNorm backward sums them.
        defer_dx: the caller's next consumer of dx is a GroupNorm backward, so dx may come back unreduced: then the return value is
        ((dx, slabs), dx2, dmgf)."""
        cfg = self.cfg
        r, x, x2 = st["r"], st["x"], st["x2"]
        k = cfg.kernel_size
        B, T, _ = x.shape
        co, ci = r["cout"], r["cin"]
        C1 = x.shape[-1]
        x4 = x.view(B, 1, T, C1)
        x24 = None if x2 is None else x2.view(B, 1, T, -1)
        d4 = dout.view(B, 1, T, co)
        dc1, _, _ = self._gn_bwd(st["s1"], d4, grads, dslabs=dslabs, keep_dout=True,      # d4 is read again below (rc, residual)
                                 want_twin=st.get("a0_h") is not None and co >= 64)
        c1v, c0v, cev = r["c1"], r["c0"], r["ce"]
        dc1h = self._twin_dy(dc1, st.get("a0_h"), co)
        self._wg(st["a0"].view(B, 1, T, co), dc1, c1v.shape, 1, k, (1, 1), (0, k // 2), dw=grads[c1v.wname], dbias=grads[c1v.bname],
                 x_h=st.get("a0_h"), dy_h=dc1h)
        dfr = self._defer_ok(T, co, cfg.n_groups)
        da0, sl = _dgrad(dc1, c1v, None, co, 1, k, (1, 1), (0, k // 2), x_h=dc1h, defer=True) if dfr else \
            (_dgrad(dc1, c1v, None, co, 1, k, (1, 1), (0, k // 2), x_h=dc1h), None)
        if self._dfilm_all is not None:          # batched FiLM: the gradient rows go into this block's columns of [B, NF]
            o = r["film_off"]
            dc0, _, _ = self._gn_bwd(st["s0"], da0, grads, want_dfilm=True, dfilm_out=self._dfilm_all[:, o:o + 2 * co], dslabs=sl,
                                     want_twin=st.get("x_h") is not None and co >= 64)
            dc0_tw = self._take_tw(dc0)
        else:
            dc0, _, dfilm = self._gn_bwd(st["s0"], da0, grads, want_dfilm=True, dslabs=sl)
            dc0_tw = None
            df2 = dfilm.view(B, 2 * co)
            self._wg(self._mgf.view(1, 1, B, -1), df2.view(1, 1, B, -1), cev.shape, 1, 1, dw=grads[cev.wname], dbias=grads[cev.bname])
            dmgf = _dgrad(df2.view(1, 1, B, -1), cev, None, cev.ci, 1, 1, (1, 1), (0, 0),
                          residual=None if dmgf is None else dmgf.view(1, 1, B, -1)).view(B, -1)
        xh, x2h = st.get("x_h"), st.get("x2_h")
        self._set_tw(dc0_tw, dc0)
        dc0h = self._twin_dy(dc0, xh, co)
        self._wg(x4, dc0, c0v.shape, 1, k, (1, 1), (0, k // 2), x2=x24, dw=grads[c0v.wname], dbias=grads[c0v.bname], x_h=xh, dy_h=dc0h,
                 x2_h=x2h)
        rc = r["rc"]
        d4h = None
        if rc is not None:
            d4h = self._twin_dy(d4, xh, co)
            self._wg(x4, d4, rc.shape, 1, 1, x2=x24, dw=grads[rc.wname], dbias=grads[rc.bname], x_h=xh, dy_h=d4h, x2_h=x2h)
        if not need_dx:
            return None, None, dmgf
        if rc is not None:
            first = _dgrad(dc0, c0v, None, ci, 1, k, (1, 1), (0, k // 2),
                               residual=None if extra is None else extra.view(B, 1, T, ci), x_h=dc0h)
            if x2 is None:
                if defer_dx:
                    dx, sl = _dgrad(d4, rc, None, ci, 1, 1, residual=first, x_h=d4h, defer=True)
                    return (dx.view(B, T, ci), sl), None, dmgf
                dx = _dgrad(d4, rc, None, ci, 1, 1, residual=first, x_h=d4h).view(B, T, ci)
                return dx, None, dmgf
            C2 = ci - C1
            dxa = torch.empty((B, T, C1), dtype=torch.float32, device=x.device)
            dxb = torch.empty((B, T, C2), dtype=torch.float32, device=x.device)
            _dgrad(d4, rc, None, ci, 1, 1, residual=first, y=dxa.view(B, 1, T, C1), y2=dxb.view(B, 1, T, C2), csplit=C1)
            return dxa, dxb, dmgf
        res = dout if extra is None else ops.axpy(dout, extra)
        if defer_dx:
            dx, sl = _dgrad(dc0, c0v, None, ci, 1, k, (1, 1), (0, k // 2), residual=res.view(B, 1, T, ci), x_h=dc0h, defer=True)
            return (dx.view(B, T, ci), sl), None, dmgf
        dx = _dgrad(dc0, c0v, None, ci, 1, k, (1, 1), (0, k // 2), residual=res.view(B, 1, T, ci), x_h=dc0h).view(B, T, ci)
        return dx, None, dmgf

    def unet_fwd(self, sample, t_long, global_cond, save=None):
        """sample [B,T,Da] fp32, t_long [B] int64, global_cond [B,G] -> eps prediction [B,T,Da]."""
        cfg = self.cfg
        B, T, Da = sample.shape
        ops.tstamp("unet_fwd begin")
        self._p3 = {}
        self._p3_active = self._p3_mode()
        try:
            return self._unet_fwd(sample, t_long, global_cond, save)
        finally:
            self._p3_active = False
            self._p3 = {}

    def _unet_fwd(self, sample, t_long, global_cond, save):
        cfg = self.cfg
        B, T, Da = sample.shape
        temb = ops.sincos_embed(t_long, cfg.dsed, 0)
        e1 = ops.linear(temb, self.step1.pf(), self.step1.b)
        m1 = ops.act_fwd(e1, "mish")
        e2 = ops.linear(m1, self.step3.pf(), self.step3.b)
        Gd = global_cond.shape[1]
        gf = torch.empty((B, cfg.dsed + Gd), dtype=torch.float32, device=sample.device)
        ops.copy2d(e2, gf, B, cfg.dsed, cfg.dsed, cfg.dsed + Gd)
        ops.copy2d(global_cond, gf, B, Gd, Gd, cfg.dsed + Gd, dst_off=cfg.dsed)
        mgf = ops.act_fwd(gf, "mish")
        self._mgf = mgf
        self._film_all = None
        if self.batch_film:
            if self._film_ver != self._film_key():
                self._refresh_film()
            self._film_all = ops.linear(mgf, self._film_w, self._film_b)          # [B, NF]: every block's (scale | shift) rows
        tape = [] if save is not None else None
        x = sample
        hs = []
        n = len(self.down)
        for i, lvl in enumerate(self.down):
            x = self._res_fwd(lvl["r0"], x, mgf, tape)
            x = self._res_fwd(lvl["r1"], x, mgf, tape)
            hs.append(x)
            if lvl["ds"] is not None:
                xin = x
                x = self._c1d(x, lvl["ds"], 3, stride=2, pad=1)
                if tape is not None:
                    tape.append(dict(ds=lvl["ds"], x=xin))
        for r in self.mid:
            x = self._res_fwd(r, x, mgf, tape)
        for lvl in self.up:
            skip = hs.pop()
            x = self._res_fwd(lvl["r0"], x, mgf, tape, x2=skip)
            x = self._res_fwd(lvl["r1"], x, mgf, tape)
            xin = x
            us = lvl["us"]
            Bx, Tx, Cx = x.shape
            x = _dgrad(x.view(Bx, 1, Tx, Cx), us, us.b, us.ci, 1, 4, (1, 1), (0, 2), idil=2, out_hw=(1, 2 * Tx)).view(Bx, 2 * Tx, us.ci)
            if tape is not None:
                tape.append(dict(us=us, x=xin))
        k = cfg.kernel_size
        dfr = self._defer_ok(x.shape[1], self.fin0.co, 8)
        c, sl = self._c1d(x, self.fin0, k, defer=True) if dfr else (self._c1d(x, self.fin0, k), None)
        a, sf = self._gn(c.view(B, 1, -1, self.fin0.co), "model.final_conv.0.block.1", 8, "mish", slabs=sl)
        a = a.view(B, -1, self.fin0.co)
        pred = self._c1d(a, self.fin1, 1, pad=0)
        if save is not None:
            save.update(tape=tape, temb=temb, e1=e1, m1=m1, gf=gf, fin_x=x, fin_a=a, fin_s=sf, n_levels=n)
        return pred

    def unet_bwd(self, dpred, save, grads):
        """Returns d(global_cond) [B,G]."""
        self._p3 = {}
        self._p3_active = self._p3_mode()
        try:
            return self._unet_bwd(dpred, save, grads)
        finally:
            self._p3_active = False
            self._p3 = {}

    def _unet_bwd(self, dpred, save, grads):
        cfg = self.cfg
        k = cfg.kernel_size
        B, T, Da = dpred.shape
        f1, f0 = self.fin1, self.fin0
        a, x = save["fin_a"], save["fin_x"]
        d4 = dpred.view(B, 1, T, Da)
        self._wg(a.view(B, 1, T, -1), d4, f1.shape, 1, 1, dw=grads[f1.wname], dbias=grads[f1.bname])
        da = _dgrad(d4, f1, None, f1.ci, 1, 1)
        dc, _, _ = self._gn_bwd(save["fin_s"], da, grads)
        self._wg(x.view(B, 1, T, -1), dc, f0.shape, 1, k, (1, 1), (0, k // 2), dw=grads[f0.wname], dbias=grads[f0.bname])
        dx = _dgrad(dc, f0, None, f0.ci, 1, k, (1, 1), (0, k // 2)).view(B, T, f0.ci)
        tape = save["tape"]
        dmgf = None
        self._dfilm_all = None
        if self._film_all is not None and self._film_grads_contiguous(grads):
            self._dfilm_all = torch.empty((B, self.film_nf), dtype=torch.float32, device=dpred.device)
        dx_sl = None               # set when dx still is the split-K slabs of its conv (the next residual block's GroupNorm sums them)
        pending_skip = []          # gradients flowing into hs entries from the up path (LIFO order of use)
        # walk the tape backwards
        i = len(tape) - 1
        n_res_seen = 0
        total_res = sum(1 for e in tape if "r" in e)
        skip_for_level = {}
        while i >= 0:
            e = tape[i]
            if "us" in e:
                us, xin = e["us"], e["x"]
                Bx, Tx, Cx = xin.shape
                dy4 = dx.view(Bx, 1, 2 * Tx, us.ci)
                x4 = xin.view(Bx, 1, Tx, Cx)
                self._wg(dy4, x4, us.shape, 1, 4, (1, 2), (0, 1), dw=grads[us.wname])
                ops.colsum(dx.view(-1, us.ci), out=grads[us.bname])
                dx = ops.conv2d(dy4, us.pf(), None, Cx, 1, 4, (1, 2), (0, 1)).view(Bx, Tx, Cx)
            elif "ds" in e:
                ds, xin = e["ds"], e["x"]
                Bx, Tx, Cx = xin.shape
                dy4 = dx.view(Bx, 1, -1, ds.co)
                x4 = xin.view(Bx, 1, Tx, Cx)
                self._wg(x4, dy4, ds.shape, 1, 3, (1, 2), (0, 1), dw=grads[ds.wname], dbias=grads[ds.bname])
                skip = pending_skip.pop() if pending_skip else None
                dx = _dgrad(dy4, ds, None, Cx, 1, 3, (1, 1), (0, 1), idil=2, out_hw=(1, Tx),
                                residual=None if skip is None else skip.view(Bx, 1, Tx, Cx)).view(Bx, Tx, Cx)
            else:
                n_res_seen += 1
                first_block = (n_res_seen == total_res)          # down_modules.0.0: its input is data -> no dx
                extra = None
                # the first mid block consumes hs[-1] directly (Identity downsample on the last level)
                if e["r"] is self.mid[0]:
                    extra = pending_skip.pop()
                # dx may stay unreduced when the next consumer is the GroupNorm backward of another residual block on the wave path
                nxt = tape[i - 1] if i > 0 else None
                defer_dx = (not first_block and nxt is not None and "r" in nxt and e["x2"] is None
                            and self._defer_ok(dx.shape[1], e["r"]["cin"], cfg.n_groups))
                dx, dx2, dmgf = self._res_bwd(e, dx, grads, dmgf, extra=extra, need_dx=not first_block, dslabs=dx_sl, defer_dx=defer_dx)
                ops.tstamp_fine(f"unet_bwd {e['r']['pre'][6:]} done")
                dx_sl = None
                if isinstance(dx, tuple):
                    dx, dx_sl = dx
                if dx2 is not None:
                    pending_skip.append(dx2)
            i -= 1
        # pending_skip is pushed in up-path order (deepest level first) and popped by the matching consumers
        # in reverse tape order: mid[0] pops the deepest, then each downsample pops the next shallower one.
        if self._dfilm_all is not None:
            # the 16 FiLM projections' gradients as two GEMMs over the concatenated operands: dW_cat / db_cat land directly in the
            # arena (their 32 slices are contiguous in film order: trainable_names), d Mish(cond) = dFiLM_all @ W_cat
            df4 = self._dfilm_all.view(1, 1, B, self.film_nf)
            w0, b0 = self.film[0]["ce"].wname, self.film[0]["ce"].bname
            self._wg(self._mgf.view(1, 1, B, -1), df4, (self.film_nf, self.film_gd), 1, 1, dw=grads[w0], dbias=grads[b0])
            dmgf = ops.conv2d(df4, self._film_wd, None, self.film_gd, 1, 1).view(B, -1)
            self._dfilm_all = None
        # --- step encoder
        gf = save["gf"]
        dgf = ops.act_bwd(gf, dmgf, "mish")
        Gd = gf.shape[1] - cfg.dsed
        de2 = torch.empty((B, cfg.dsed), dtype=torch.float32, device=gf.device)
        dgc = torch.empty((B, Gd), dtype=torch.float32, device=gf.device)
        ops.copy2d(dgf, de2, B, cfg.dsed, gf.shape[1], cfg.dsed)
        ops.copy2d(dgf, dgc, B, Gd, gf.shape[1], Gd, src_off=cfg.dsed)
        s3, s1 = self.step3, self.step1
        self._wg(save["m1"].view(1, 1, B, -1), de2.view(1, 1, B, -1), s3.shape, 1, 1, dw=grads[s3.wname], dbias=grads[s3.bname])
        dm1 = _dgrad(de2.view(1, 1, B, -1), s3, None, s3.ci, 1, 1, (1, 1), (0, 0)).view(B, -1)
        de1 = ops.act_bwd(save["e1"], dm1, "mish")
        self._wg(save["temb"].view(1, 1, B, -1), de1.view(1, 1, B, -1), s1.shape, 1, 1, dw=grads[s1.wname], dbias=grads[s1.bname])
        return dgc

    # ------------------------------------------------------------------ policy level
    def _enc_parallel(self, fns):
        """Run the per-camera encoder chains (independent: separate weights, no shared state) as parallel branches: the first on
        the current stream, the others on side streams with their own scratch lanes; joined before returning."""
        if not self.enc_streams or len(fns) < 2 or self._cur_batch < 8:      # B = 1 rollouts: fork / join costs more than it hides
            return [f() for f in fns]
        main = torch.cuda.current_stream()
        while len(self._enc_side) < len(fns) - 1:
            self._enc_side.append(torch.cuda.Stream(device=self.device))
        out = [None] * len(fns)
        for i, f in enumerate(fns[1:], 1):
            st = self._enc_side[i - 1]
            st.wait_stream(main)
            with torch.cuda.stream(st), ops.ws_lane(_LANE_ENC0 + i):
                out[i] = f()
        out[0] = fns[0]()
        for st in self._enc_side[:len(fns) - 1]:
            main.wait_stream(st)
        return out

    def global_cond(self, imgs: dict, save=None):
        keys = list(self.cfg.rgb_keys)
        self._cur_batch = imgs[keys[0]].shape[0]
        feats = self._enc_parallel([(lambda k=k: self.encode_fwd(k, imgs[k], save)) for k in self.cfg.rgb_keys])
        B = feats[0].shape[0]
        fd = feats[0].shape[1]
        gc = torch.empty((B, fd * len(feats)), dtype=torch.float32, device=feats[0].device)
        for i, f in enumerate(feats):
            ops.copy2d(f, gc, B, fd, fd, fd * len(feats), dst_off=i * fd)
        return gc

    def grad_layout(self, names):
        """Flat-arena layout for the gradients of `names` (canonical parameter names, optimiser order)."""
        offs, total = {}, 0
        for n in names:
            offs[n] = total
            total += self.P[n].numel()
        return offs, total

    def grad_views(self, arena, names):
        offs, total = self.grad_layout(names)
        assert arena.numel() >= total
        return {n: arena[offs[n]:offs[n] + self.P[n].numel()].view(self.P[n].shape) for n in names}

    def loss_fwd_bwd(self, imgs: dict, action, noise, timesteps, need_grad=True, names=None, arena=None):
        """compute_loss (+ backward).  imgs[key] [B,3,H,W]; action/noise [B,T,Da]; timesteps [B] int64.
        Gradients (torch layout) are written into views of one flat fp32 arena (a fresh zeroed one unless given): the same
        buffer feeds the RCCL all-reduce and the fused optimiser.  Returns (loss[1], {name: grad view}, arena)."""
        if not need_grad:
            gc = self.global_cond(imgs, None)
            noisy = ops.add_noise(action, noise, timesteps, self.ac, self.act_limits)
            pred = self.unet_fwd(noisy, timesteps, gc, None)
            loss, _ = ops.mse_loss(pred, noise, want_grad=False)
            return loss, None, None
        st = self.backward_phase1(imgs, action, noise, timesteps, names=names, arena=arena)
        self.backward_phase2(st)
        return st["loss"], st["grads"], st["arena"]

    def backward_phase1(self, imgs, action, noise, timesteps, names=None, arena=None):
        """forward + loss + ConditionalUnet1D backward: afterwards the `model.*` slice of the arena is final (its all-reduce can
        start while phase 2 runs).  Returns the state phase 2 needs."""
        save_enc = {}
        gc = self.global_cond(imgs, save_enc)
        noisy = ops.add_noise(action, noise, timesteps, self.ac, self.act_limits)
        save = {}
        if self._pack_join is not None:             # (trainer: the ConditionalUnet1D operands were re-packed on a side stream under the encoders)
            torch.cuda.current_stream().wait_stream(self._pack_join)
            self._pack_join = None
        pred = self.unet_fwd(noisy, timesteps, gc, save)
        loss, dpred = ops.mse_loss(pred, noise, want_grad=True, grad_scale_ptr=self.loss_scale_ptr)
        ops.tstamp("unet_bwd begin")
        names = list(names) if names is not None else self.trainable_names()
        if arena is None:
            arena = torch.zeros(self.grad_layout(names)[1], dtype=torch.float32, device=self.device)   # GN param grads accumulate
        grads = self.grad_views(arena, names)
        self._dw_names = {v.data_ptr(): n for n, v in grads.items()}
        self._collect_wg = True
        tok = self._gn_begin()
        self._wg_begin()
        if not self.defer_unet_wgrad:          # (weight gradients inside phase 1: grouped launches at its end)
            self._wgb_begin()
        try:
            dgc = self.unet_bwd(dpred, save, grads)
        finally:
            self._collect_wg = False
            self._wgb_end()
            self._gn_flush(tok, grads)
            self._wg_flush()
        ops.tstamp("unet_bwd end")
        return dict(loss=loss, grads=grads, arena=arena, dgc=dgc, save_enc=save_enc, keep=save)

    def backward_phase2(self, st):
        """image-encoder backward (the `obs_encoder.*` slice of the arena)."""
        dgc, grads = st["dgc"], st["grads"]
        B = dgc.shape[0]
        fd = self.cfg.feature_dim
        nk = len(self.cfg.rgb_keys)
        def one(i, key):
            df = torch.empty((B, fd), dtype=torch.float32, device=dgc.device)
            ops.copy2d(dgc, df, B, fd, fd * nk, fd, src_off=i * fd)
            self.encode_bwd(key, df, st["save_enc"][key], grads)
            return df

        main = torch.cuda.current_stream()
        deferred = []
        if not self.split_deferred:                    # (data parallel: the trainer runs them itself -- run_deferred_wgrads -- on the stream
            deferred, self._deferred = self._deferred, []      # its first slice all-reduce then leaves from)
        if deferred:                                   # third branch: every ConditionalUnet1D weight gradient
            if self._wg_stream is None:
                self._wg_stream = torch.cuda.Stream(device=self.device)
            self._wg_stream.wait_stream(main)
            with torch.cuda.stream(self._wg_stream), ops.ws_lane(_LANE_WG):
                self._launch_deferred(deferred)
                if self.on_unet_wgrads_done is not None:       # (trainer: the model.* slice of the arena is final -> its gradient-norm partial sums)
                    self.on_unet_wgrads_done()
        self._in_enc = True
        try:
            self._enc_parallel([(lambda i=i, key=key: one(i, key)) for i, key in enumerate(self.cfg.rgb_keys)])
        finally:
            self._in_enc = False
        if deferred:
            main.wait_stream(self._wg_stream)

    def _launch_deferred(self, deferred):
        ops.tstamp("unet_wgrad begin")
        col = self._collector() if self._wgc_on else None
        batch = ops.WgradBatch(col) if col is not None else None
        for a, k in deferred:
            kk = dict(k, slab_key=k.get("slab_key") or self._slab_key(k.get("dw")))
            if batch is not None and batch.add(*a, **kk):
                continue
            ops.conv2d_wgrad(*a, **dict(kk, collector=col))
        if batch is not None:
            batch.launch()
        if col is not None:
            col.flush()
        ops.tstamp("unet_wgrad end")

    def run_deferred_wgrads(self):
        """Data-parallel step (split_deferred): the ConditionalUnet1D weight gradients collected by backward_phase1, launched on the CURRENT
        stream (the trainer's side stream, as a graph of their own) -- after them the `model.*` arena slice is final and its all-reduce
        starts from that stream, while the encoder backward runs on the main one."""
        deferred, self._deferred = self._deferred, []
        if deferred:
            with ops.ws_lane(_LANE_WG):
                self._launch_deferred(deferred)
        return deferred        # the CALLER keeps the operands alive until this stream has been joined (the encoder backward allocates meanwhile)

    def arena_slices(self, names):
        """(start, end) element ranges of the `model.*` (ConditionalUnet1D) and the remaining (encoder) gradients in the arena."""
        offs, total = self.grad_layout(names)
        m = [n for n in names if n.startswith("model.")]
        if not m:
            return (0, 0), (0, total)
        lo = offs[m[0]]
        hi = offs[m[-1]] + self.P[m[-1]].numel()
        assert hi - lo == sum(self.P[n].numel() for n in m), "model.* gradients must be contiguous in the arena"
        assert lo == 0 or hi == total
        return (lo, hi), ((0, lo) if hi == total else (hi, total))

    def _film_grads_contiguous(self, grads):
        """True when the gradient views of the 16 FiLM weights (then the 16 biases) are adjacent in film order (trainable_names)."""
        ws = [grads.get(r["ce"].wname) for r in self.film]
        bs = [grads.get(r["ce"].bname) for r in self.film]
        if any(t is None for t in ws + bs):
            return False
        for seq in (ws, bs):
            for a, b in zip(seq[:-1], seq[1:]):
                if a.data_ptr() + 4 * a.numel() != b.data_ptr():
                    return False
        return True

    def trainable_names(self):
        """Every parameter the backward produces a gradient for (the reference trains all of these), in arena order: the
        `model.*` group first with its 16 FiLM weights, then its 16 FiLM biases, at the end of the group in film order (their
        gradients are written by ONE batched GEMM), then the encoders."""
        skip = ("_dummy_variable", "temperature")
        names = [n for n, p in self.P.items() if torch.is_floating_point(p) and p.dim() > 0 and p.numel() > 0
                 and not any(n.endswith(s) for s in skip) and not n.endswith("pos_x") and not n.endswith("pos_y")]
        fw = [r["ce"].wname for r in self.film]
        fb = [r["ce"].bname for r in self.film]
        film = set(fw + fb)
        model = [n for n in names if n.startswith("model.") and n not in film]
        rest = [n for n in names if not n.startswith("model.")]
        first_is_model = bool(names) and names[0].startswith("model.")
        grp = model + [n for n in fw if n in names] + [n for n in fb if n in names]
        return grp + rest if first_is_model else rest + grp
