"""Fused multi-tensor optimiser tail: global-norm clip -> AdamW -> zero grads -> EMA, three kernel launches, hipGraph-safe.
Host wrapper over v2a_opt_step (csrc/optim.hip).  Replaces lb_online_trainer_v7.py:604-624."""
import ctypes
import torch
from ._lib import lib, check
from . import ops


class FusedAdamWEMA:
    def __init__(self, params, grads, ema_params=None, lr=1e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6, max_norm=1.0,
                 ema_inv_gamma=1.0, ema_power=0.75, ema_min_value=0.0, ema_beta=0.9999, ema_update_after_step=0, ema_update_every=1):
        """params / grads / ema_params: equally long lists of fp32 CUDA tensors (grads typically views of one arena)."""
        assert len(params) == len(grads)
        self.device = params[0].device
        self.params, self.grads, self.ema = list(params), list(grads), (list(ema_params) if ema_params is not None else None)
        total = sum(p.numel() for p in params)
        self.m = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.v = torch.zeros(total, dtype=torch.float32, device=self.device)
        chunk = lib.v2a_opt_chunk_elems()
        rows, chunks, off = [], [], 0
        for i, p in enumerate(params):
            n = p.numel()
            assert p.is_contiguous() and grads[i].is_contiguous() and grads[i].numel() == n
            e = self.ema[i].data_ptr() if self.ema is not None else 0
            rows.append([p.data_ptr(), grads[i].data_ptr(), self.m.data_ptr() + 4 * off, self.v.data_ptr() + 4 * off, e, n])
            for s in range(0, n, chunk):
                chunks.append([i, s])
            off += n
        self.table = torch.tensor(rows, dtype=torch.int64).to(self.device)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).to(self.device)
        self.nchunks = len(chunks)
        self.partial = torch.zeros(self.nchunks, dtype=torch.float64, device=self.device)
        nb = lib.v2a_opt_state_bytes()
        host = (ctypes.c_ubyte * nb)()
        check(lib.v2a_opt_state_init(ctypes.addressof(host), lr, betas[0], betas[1], eps, weight_decay, max_norm, ema_inv_gamma,
                                     ema_power, ema_min_value, ema_beta, ema_update_after_step, ema_update_every), "opt_state_init")
        self.state = torch.frombuffer(bytearray(host), dtype=torch.uint8).clone().to(self.device)
        self._nb = nb
        self.hyper = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self._offsets = [0]
        for p in params:
            self._offsets.append(self._offsets[-1] + p.numel())

    # ------------------------------------------------------------------ dynamic loss scaling (fp16 mode)
    def enable_loss_scaling(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, growth_tracker=0):
        """torch.cuda.amp.GradScaler's defaults and contract, inside the fused tail (csrc/optim.hip): call before the first step (or to
        restore a checkpointed scaler: growth_tracker = its `_growth_tracker`).  init_scale <= 0 switches it off.  Returns the device
        address of the scale (the loss-gradient kernel multiplies by it)."""
        host = self.state.cpu().numpy().tobytes()
        buf = ctypes.create_string_buffer(host, len(host))
        check(lib.v2a_opt_state_set_scaler(ctypes.addressof(buf), float(init_scale), float(growth_factor), float(backoff_factor),
                                           int(growth_interval), int(growth_tracker)), "opt_state_set_scaler")
        self.state.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
        return self.loss_scale_ptr() if init_scale > 0 else 0

    def loss_scale_ptr(self):
        return self.state.data_ptr() + lib.v2a_opt_state_loss_scale_offset()

    def scaler(self):
        """(loss scale, growth tracker, last step skipped?, skipped steps so far) -- synchronises."""
        host = self.state.cpu().numpy().tobytes()
        buf = ctypes.create_string_buffer(host, len(host))
        ls, gt, sk, n = ctypes.c_float(), ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        check(lib.v2a_opt_state_scaler(ctypes.addressof(buf), ctypes.byref(ls), ctypes.byref(gt), ctypes.byref(sk), ctypes.byref(n),
                                       None, None, None, None), "opt_state_scaler")
        return ls.value, gt.value, bool(sk.value), n.value

    def scaler_state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict()'s layout ({'scale', 'growth_factor', 'backoff_factor', 'growth_interval',
        '_growth_tracker'}: what the reference checkpoints as accelerator.scaler.state_dict(), lb_online_trainer_v7.py:377), or None
        when loss scaling is off -- synchronises."""
        host = self.state.cpu().numpy().tobytes()
        buf = ctypes.create_string_buffer(host, len(host))
        ls, gt, gf, bf = ctypes.c_float(), ctypes.c_int(), ctypes.c_float(), ctypes.c_float()
        gi, on = ctypes.c_int(), ctypes.c_int()
        check(lib.v2a_opt_state_scaler(ctypes.addressof(buf), ctypes.byref(ls), ctypes.byref(gt), None, None, ctypes.byref(gf),
                                       ctypes.byref(bf), ctypes.byref(gi), ctypes.byref(on)), "opt_state_scaler")
        if not on.value:
            return None
        return {"scale": float(ls.value), "growth_factor": float(gf.value), "backoff_factor": float(bf.value),
                "growth_interval": int(gi.value), "_growth_tracker": int(gt.value)}

    def load_scaler_state_dict(self, sd):
        """Restore a GradScaler state dict (see scaler_state_dict); None / empty: leave the scaler as it is.  Only meaningful when loss
        scaling is on (fp16 mode); returns the device address of the scale."""
        if not sd:
            return 0
        return self.enable_loss_scaling(float(sd["scale"]), float(sd.get("growth_factor", 2.0)), float(sd.get("backoff_factor", 0.5)),
                                        int(sd.get("growth_interval", 2000)), int(sd.get("_growth_tracker", 0)))

    def set_pack_rows(self, rows):
        """rows[i] = (forward-pack address or 0, Cin, taps, channel-window-pack address or 0, 16-bit twin address or 0, twin is fp16) of parameter i (PolicyEngine.opt_pack_rows):
        step(packs=True) then writes those conv operands from the update kernel itself.  None clears."""
        if rows is None:
            self.pack_table = None
            return
        assert len(rows) == len(self.params)
        self.pack_table = torch.tensor([[int(r[0]), max(int(r[1]), 1), max(int(r[2]), 1), int(r[3]), int(r[4]), int(r[5])] for r in rows],
                                       dtype=torch.int64).to(self.device)

    pack_table = None

    def step(self, zero_grad=True, packs=False, presum=(0, 0)):
        """presum = (first, count): those chunks' gradient sums of squares were taken by presum(first, count) earlier in THIS step (a
        stream ordered before this call); the step sums only the rest.  (0, 0): everything here."""
        pk = self.pack_table.data_ptr() if (packs and self.pack_table is not None) else None
        check(lib.v2a_opt_step_packed(self.table.data_ptr(), self.chunks.data_ptr(), self.nchunks, self.state.data_ptr(),
                                      self.partial.data_ptr(), 1 if zero_grad else 0, pk, int(presum[0]), int(presum[1]), ops._stream()),
              "opt_step")

    def chunk_range(self, t0, t1):
        """(first chunk, number of chunks) covering tensors t0 .. t1 - 1 of the optimiser's list."""
        chunk = lib.v2a_opt_chunk_elems()
        nch = [-(-p.numel() // chunk) for p in self.params]
        return sum(nch[:t0]), sum(nch[t0:t1])

    def presum(self, first, count):
        """Sum the squares of the gradients of chunks [first, first + count) now (current stream); hand the same range to step(presum=...)."""
        if count > 0:
            check(lib.v2a_opt_presum(self.table.data_ptr(), self.chunks.data_ptr(), int(first), int(count), self.partial.data_ptr(), ops._stream()),
                  "opt_presum")

    def scale_grads(self, scale: float):
        check(lib.v2a_opt_scale_grads(self.table.data_ptr(), self.chunks.data_ptr(), self.nchunks, float(scale), ops._stream()),
              "opt_scale_grads")

    def peek(self):
        """(grad_norm before clipping, clip coefficient, step, ema decay) of the last step -- synchronises."""
        host = self.state.cpu().numpy().tobytes()
        buf = ctypes.create_string_buffer(host, len(host))
        gn, cc, dec = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        st = ctypes.c_longlong()
        lib.v2a_opt_state_peek(ctypes.addressof(buf), ctypes.byref(gn), ctypes.byref(cc), ctypes.byref(st), ctypes.byref(dec))
        return gn.value, cc.value, st.value, dec.value

    # ------------------------------------------------------------------ checkpointing (lb_online_trainer_v7.py:367-408)
    def counters(self):
        """(AdamW step, EMA step, EMA initted) -- synchronises."""
        host = self.state.cpu().numpy().tobytes()
        buf = ctypes.create_string_buffer(host, len(host))
        st, es, ini = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
        check(lib.v2a_opt_state_counters(ctypes.addressof(buf), ctypes.byref(st), ctypes.byref(es), ctypes.byref(ini)), "opt_state_counters")
        return st.value, es.value, bool(ini.value)

    def set_counters(self, step, ema_step, ema_initted, lr=0.0):
        host = self.state.cpu().numpy().tobytes()
        buf = ctypes.create_string_buffer(host, len(host))
        check(lib.v2a_opt_state_set_counters(ctypes.addressof(buf), int(step), int(ema_step), 1 if ema_initted else 0, float(lr)),
              "opt_state_set_counters")
        self.state.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))

    def state_dict(self, order=None):
        """torch.optim.AdamW's layout ({'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}).  `order[i]` = index
        into this optimiser's tensor list of the i-th entry of the checkpoint's parameter list, or -1 for a parameter the optimiser
        was given but never updates (no gradient: torch keeps no state for those either).  Default: identity."""
        step = self.counters()[0]
        order = list(range(len(self.params))) if order is None else list(order)
        state = {}
        if step > 0:
            for i, j in enumerate(order):
                if j < 0:
                    continue
                lo, hi = self._offsets[j], self._offsets[j + 1]
                shp = self.params[j].shape
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": self.m[lo:hi].view(shp).clone(),
                            "exp_avg_sq": self.v[lo:hi].view(shp).clone()}
        group = dict(lr=self.hyper["lr"], betas=self.hyper["betas"], eps=self.hyper["eps"], weight_decay=self.hyper["weight_decay"],
                     amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(order))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, order=None, ema_step=None, ema_initted=None):
        order = list(range(len(self.params))) if order is None else list(order)
        step = 0
        for i, j in enumerate(order):
            if j < 0:
                continue
            ent = sd["state"].get(i)
            lo, hi = self._offsets[j], self._offsets[j + 1]
            if ent is None:
                self.m[lo:hi].zero_()
                self.v[lo:hi].zero_()
                continue
            self.m[lo:hi].copy_(ent["exp_avg"].reshape(-1))
            self.v[lo:hi].copy_(ent["exp_avg_sq"].reshape(-1))
            step = int(float(ent["step"]))
        _, es, ini = self.counters()
        lr = float(sd["param_groups"][0]["lr"])
        self.hyper["lr"] = lr
        self.set_counters(step, es if ema_step is None else ema_step, ini if ema_initted is None else ema_initted, lr=lr)
