"""Training step of the AVDC video diffusion model on the MI355X.

Reference: GoalGaussianDiffusion.forward / p_losses (flowdiffusion/flowdiffusion/goal_diffusion.py:690-724) differentiated by autograd,
driven by Trainer.train (:953-990: clip_grad_norm_(1.0) -> Adam.step -> zero_grad -> EMA.update).  Here:

  q_sample (+ 2x-1)          one elementwise kernel on the 'b (f c) h w' tensors, then the channels-last input pack of the sampler
  UNet forward               UNetTrainEngine.forward_train_tokens (keeps a tape)
  loss                       per-sample mean * loss_weight[t] -> batch mean, fixed summation order
  backward                   hand-written (unet_train.py): every parameter gradient of the UNet and of the text branch
  optimiser tail             FusedAdamWEMA (clip -> Adam -> zero -> EMA with ema_pytorch's warm-up), one multi-tensor launch set

Two entry points share the same kernels:
  * `diffusion_loss(diffusion, img, cond, tokens)` returns a scalar that participates in torch autograd (`loss.backward()` fills `.grad`):
    the drop-in for user code written against the reference module;
  * `VideoTrainStep` is what flowdiffusion's Trainer drives: gradients land in one arena (one RCCL all-reduce when world > 1), the fused
    optimiser consumes it, packed operands are refreshed afterwards.
fp32 throughout (the reference trains under fp16 autocast; fp32 is the parity configuration)."""
import sys
import os
import torch
from . import ops
from .unet_train import UNetTrainEngine


def _drop_graph(diffusion):
    from flowdiffusion.flowdiffusion.goal_diffusion import drop_sampler_graph
    drop_sampler_graph(diffusion)


def _train_engine(model):
    """UNetTrainEngine of a _HipUnetWrapper, rebuilt when the module moved; packs refreshed when torch changed any parameter."""
    params = dict(model.named_parameters())
    dev = next(iter(params.values())).device
    if dev.type != "cuda":
        raise RuntimeError("video-model training runs on a HIP device only: call .to('cuda') first (no CPU fallback)")
    eng = model.__dict__.get("_train_eng")
    if eng is None or eng.device != dev or any(eng.P[n] is not p for n, p in params.items()):
        eng = UNetTrainEngine(model.unet.engine_cfg(), params, prefix="unet.")
        model.__dict__["_train_eng"] = eng
        eng._versions = None
    ver = sum(p._version for p in params.values())
    if eng._versions != ver:                       # an optimiser that goes through torch bumped the version counters
        eng.refresh_packs()
        eng._versions = ver
    return eng


class _GradArena:
    """One flat fp32 buffer with a view per trainable parameter (torch layout), in named_parameters() order."""

    def __init__(self, params: dict):
        dev = next(iter(params.values())).device
        self.names = list(params.keys())
        total = sum(p.numel() for p in params.values())
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = {}, 0
        for n, p in params.items():
            self.views[n] = self.flat[off:off + p.numel()].view(p.shape)
            off += p.numel()


def gradient_ready_slices(names, numels):
    """Arena element ranges in the order the hand-written backward finishes them (UNetTrainEngine.backward walks out.* and
    output_blocks.* first, then middle_block / input_blocks / time_embed / the text branch): [(decoder slice), (everything before it)].
    named_parameters() registers output_blocks and out last (guided_diffusion/unet.py:435-632), so the decoder's gradients -- 58 % of
    Unet_Libero's 201 M -- are one contiguous tail that can travel while the encoder half of the backward still runs."""
    total = sum(numels)
    off, first = 0, None
    for n, k in zip(names, numels):
        is_dec = ".output_blocks." in n or ".out." in n
        if is_dec and first is None:
            first = off
        elif not is_dec and first is not None:
            return [(0, total)]                          # decoder parameters are not a contiguous tail here: one slice
        off += k
    if first is None or first == 0:
        return [(0, total)]
    return [(first, total), (0, first)]


def _loss_and_tape(diffusion, eng, img, cond, tokens, t, noise, normalize):
    B, C, H, W = img.shape
    ci = getattr(diffusion.model, "frame_channels", 3)
    f = C // ci
    x = ops.video_qsample(img, noise, t, diffusion.sqrt_alphas_cumprod, diffusion.sqrt_one_minus_alphas_cumprod, normalize)
    xin = ops.video_pack2(x, cond, f, H, W, ci)
    out, tape = eng.forward_train_tokens(xin, t, tokens)
    loss = ops.video_loss_fwd(out, img, noise, t, diffusion.sqrt_alphas_cumprod, diffusion.sqrt_one_minus_alphas_cumprod, diffusion.loss_weight,
                              diffusion.objective, diffusion.loss_type, normalize)
    return loss, out, tape


def _backward(diffusion, eng, tape, out, img, noise, t, normalize, grads, gscale=None, on_decoder_done=None):
    dout = ops.video_loss_bwd(out, img, noise, t, diffusion.sqrt_alphas_cumprod, diffusion.sqrt_one_minus_alphas_cumprod, diffusion.loss_weight,
                              diffusion.objective, diffusion.loss_type, normalize, gscale)
    eng.backward(tape, dout, grads, on_decoder_done=on_decoder_done)


class _VideoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, diffusion, img, cond, tokens, t, noise, normalize, *params):
        eng = _train_engine(diffusion.model)
        loss, out, tape = _loss_and_tape(diffusion, eng, img, cond, tokens, t, noise, normalize)
        ctx.pack = (diffusion, eng, tape, out, img, noise, t, normalize)
        return loss

    @staticmethod
    def backward(ctx, g):
        diffusion, eng, tape, out, img, noise, t, normalize = ctx.pack
        ctx.pack = None
        arena = _GradArena(dict(diffusion.model.named_parameters()))
        _backward(diffusion, eng, tape, out, img, noise, t, normalize, arena.views, g.float().contiguous())
        return (None,) * 7 + tuple(arena.views[n] for n in arena.names)


def _prep(diffusion, img, cond, tokens, t, noise):
    dev = diffusion.betas.device
    if dev.type != "cuda":
        raise RuntimeError("GoalGaussianDiffusion.forward runs on a HIP device only (no CPU fallback)")
    img = img.to(dev).float().contiguous()
    cond = cond.to(dev).float().contiguous()
    tokens = tokens.to(dev).float().contiguous()
    t = t.to(dev).long().contiguous()
    if noise is None:
        noise = diffusion._noise(tuple(img.shape), dev)
    return img, cond, tokens, t, noise.to(dev).float().contiguous()


def diffusion_loss(diffusion, img, cond, tokens, t, noise=None, normalize=False):
    """p_losses as an autograd scalar; `img` is x_start (already in [-1,1]) unless `normalize`."""
    img, cond, tokens, t, noise = _prep(diffusion, img, cond, tokens, t, noise)
    params = [p for _, p in diffusion.model.named_parameters()]
    return _VideoLossFn.apply(diffusion, img, cond, tokens, t, noise, bool(normalize), *params)


class VideoTrainStep:
    """One optimisation step per call: loss -> hand-written backward into the arena -> (all-reduce) -> clip + Adam + zero + EMA.
    `ema_model`: a second GoalGaussianDiffusion whose UNet parameters receive the moving average (None = no EMA)."""

    def __init__(self, diffusion, ema_model=None, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, max_norm=1.0, ema_beta=0.995,
                 ema_update_every=10, ema_update_after_step=100, ema_inv_gamma=1.0, ema_power=2.0 / 3.0, process_group=None):
        from .optim import FusedAdamWEMA
        self.diffusion, self.ema_model = diffusion, ema_model
        self.params = dict(diffusion.model.named_parameters())
        self.arena = _GradArena(self.params)
        ema_params = None
        if ema_model is not None:
            ep = dict(ema_model.model.named_parameters())
            ema_params = [ep[n].data for n in self.arena.names]
        self.opt = FusedAdamWEMA([self.params[n].data for n in self.arena.names], [self.arena.views[n] for n in self.arena.names], ema_params,
                                 lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_norm=max_norm, ema_inv_gamma=ema_inv_gamma,
                                 ema_power=ema_power, ema_min_value=0.0, ema_beta=ema_beta, ema_update_after_step=ema_update_after_step,
                                 ema_update_every=ema_update_every)
        self.pg = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        self._acc = None
        # data parallel: the one collective of the step as two asynchronous slice all-reduces in gradient-ready order (v2a_hip.dp):
        # the decoder slice leaves as soon as the backward walk reaches the middle block and travels under the encoder half
        self.reducer = None
        if self.world > 1:
            from .dp import GradReducer
            numels = [self.params[n].numel() for n in self.arena.names]
            slices = gradient_ready_slices(self.arena.names, numels)
            # a parameter re-ordering that loses the decoder / encoder split would silently lose the overlap: say how many slices there are
            self.n_gradient_slices = len(slices)
            if len(slices) < 2:
                print(f"[v2a_hip.video_train] gradient_ready_slices found {len(slices)} slice(s): the decoder slice's early all-reduce is OFF",
                      file=sys.stderr)
            self.reducer = GradReducer(self.arena.flat, slices, process_group, self.world)

    def loss_and_grads(self, img, cond, tokens, t=None, noise=None, normalize=True, accumulate=False, scale=1.0, last=True):
        """Forward + backward of one micro-batch; gradients are written to (or, with `accumulate`, added into) the arena times `scale`.
        last: no further micro-batch follows before apply() (then a data-parallel run starts the decoder slice's all-reduce mid-backward)."""
        d = self.diffusion
        if t is None:
            t = d._draw_t(img.shape[0])
        img, cond, tokens, t, noise = _prep(d, img, cond, tokens, t, noise)
        eng = _train_engine(d.model)
        loss, out, tape = _loss_and_tape(d, eng, img, cond, tokens, t, noise, normalize)
        g = None if scale == 1.0 else torch.full((1,), float(scale), dtype=torch.float32, device=img.device)
        if accumulate:
            if self._acc is None:
                self._acc = _GradArena(self.params)
            _backward(d, eng, tape, out, img, noise, t, normalize, self._acc.views, g)
            ops.axpy(self._acc.flat, self.arena.flat, 1.0, out=self.arena.flat)
        else:
            early = None
            if self.reducer is not None and last and len(self.reducer.slices) > 1:
                early = lambda: self.reducer.launch(0)
            _backward(d, eng, tape, out, img, noise, t, normalize, self.arena.views, g, on_decoder_done=early)
        return loss

    def abandon(self):
        """Drop the gradients of the step in flight instead of apply() (e.g. a non-finite loss; in a data-parallel run EVERY rank must
        take the same decision): outstanding slice all-reduces are completed and forgotten, the arena is zeroed."""
        if self.reducer is not None:
            self.reducer.abort()
        self.arena.flat.zero_()

    def apply(self):
        """All-reduce (mean over ranks) -> fused clip / Adam / zero / EMA -> refresh the packed operands."""
        if self.reducer is not None:
            done = self.reducer.pending()
            for i in range(len(self.reducer.slices)):
                if i not in done:
                    self.reducer.launch(i)
            self.reducer.finish(self.opt.scale_grads)
        self.opt.step(zero_grad=True)
        eng = self.diffusion.model.__dict__.get("_train_eng")
        if eng is not None:
            eng.refresh_packs()
        if self.ema_model is not None:                      # the sampler engine of the averaged copy caches packed weights too
            seng = self.ema_model.model.__dict__.get("_eng")
            if seng is not None:
                seng.packs._c.clear()
            _drop_graph(self.ema_model)                     # ... and its captured sampler step points at those packs
        _drop_graph(self.diffusion)

    def step(self, img, cond, tokens, t=None, noise=None, normalize=True):
        loss = self.loss_and_grads(img, cond, tokens, t, noise, normalize)
        self.apply()
        return loss
