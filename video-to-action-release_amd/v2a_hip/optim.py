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

    def step(self, zero_grad=True):
        check(lib.v2a_opt_step(self.table.data_ptr(), self.chunks.data_ptr(), self.nchunks, self.state.data_ptr(),
                               self.partial.data_ptr(), 1 if zero_grad else 0, ops._stream()), "opt_step")

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
