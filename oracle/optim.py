"""CPU restatement of the optimiser segment of the train step.  TEST INFRASTRUCTURE.

Follows the call order of /root/reference/diffuser/libero/lb_online_trainer_v7.py:604-624:
  clip_grad_norm_(params, 1.0) -> AdamW.step -> zero_grad -> EMA.update
with hyper-parameters from /root/reference/config/libero/lb_tk8_65to72.py:138-153.

THIRD-PARTY arithmetic: torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW are present in this image and
tests/test_optim_oracle.py pins this file against them; ema_pytorch 0.2.3 is absent (requirements.txt:22) ->
its warm-up/decay rule is restated from the published code and cross-checked against the in-tree statement of
the same formula, diffuser/diffusion_policy/model/ema_model.py:44-54 (PARITY UNPINNED for ema_pytorch).
"""
import math
import torch


def global_grad_norm(grads):
    return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float(), 2.0) for g in grads]), 2.0)


def clip_coef(total_norm, max_norm=1.0):
    return torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)


def adamw_step(p, g, m, v, step, lr=1e-4, b1=0.95, b2=0.999, eps=1e-8, wd=1e-6):
    """In place, `step` is 1-based (value AFTER increment, as torch uses it)."""
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_decay(step_after_increment, update_after_step=0, inv_gamma=1.0, power=0.75, min_value=0.0, beta=0.9999):
    epoch = max(step_after_increment - update_after_step - 1, 0.0)
    if epoch <= 0:
        return 0.0
    value = 1 - (1 + epoch / inv_gamma) ** -power
    return min(max(value, min_value), beta)


class EmaState:
    """ema_pytorch.EMA.update() state machine for (update_after_step=0, update_every=1)."""

    def __init__(self, update_after_step=0, update_every=1, **kw):
        self.step = 0
        self.initted = False
        self.update_after_step = update_after_step
        self.update_every = update_every
        self.kw = kw

    def next(self):
        """Returns ('skip'|'copy'|'lerp', decay) for this update() call."""
        step = self.step
        self.step += 1
        if step % self.update_every != 0:
            return "skip", None
        if step <= self.update_after_step:
            return "copy", None
        if not self.initted:
            self.initted = True
            return "copy+lerp", ema_decay(self.step, self.update_after_step, **self.kw)
        return "lerp", ema_decay(self.step, self.update_after_step, **self.kw)


def ema_apply(ema_p, p, mode, decay):
    if mode == "skip":
        return
    if "copy" in mode:
        ema_p.copy_(p)
    if "lerp" in mode:
        d = ema_p - p
        d.mul_(1.0 - decay)
        ema_p.sub_(d)


def train_tail(params, grads, ms, vs, emas, step, ema_state: EmaState, max_norm=1.0, **opt):
    """clip -> AdamW -> (zero) -> EMA over lists of tensors, all in place.  Returns the pre-clip norm."""
    tn = global_grad_norm(grads)
    c = clip_coef(tn, max_norm)
    for g in grads:
        g.mul_(c)
    for p, g, m, v in zip(params, grads, ms, vs):
        adamw_step(p, g, m, v, step, **opt)
    mode, decay = ema_state.next()
    for e, p in zip(emas, params):
        ema_apply(e, p, mode, decay)
    return tn
