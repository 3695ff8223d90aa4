"""Host side of the persistent predict_action kernel (csrc/policy_persist.hip): turns the ConditionalUnet1D of a policy into the op list the
kernel walks -- once per scheduler configuration -- and launches it.  Reference: DiffusionUnetImagePolicy.conditional_sample
(diffusion_policy/diffusion_unet_image_policy.py:88-133) over ConditionalUnet1D.forward (model/conditional_unet1d.py:186-246).

The program holds raw pointers: to the engine's fp32 forward packs of the convs ([Cout][k][Cin], the operands the training step's kernels
read; PolicyEngine keeps them current -- launch() asks each for freshness like every layer-by-layer call does) and to the live parameters for
everything else (biases, GroupNorm affine, Linear and ConvTranspose1d weights).  A re-allocated parameter needs a new PersistentDenoiser
(same contract as a captured graph)."""
import ctypes
import torch
from ._lib import lib, check
from . import ops
from .policy_sched import ddim_coeffs, ddpm_coeffs

P, I = ctypes.c_void_p, ctypes.c_int
SRC_NONE, SRC_PLAIN, SRC_MISH, SRC_GN_MISH, SRC_SINCOS = 0, 1, 2, 3, 4
CONV, CONVT = 0, 1


class PPSrc(ctypes.Structure):
    _fields_ = [("a", P), ("gamma", P), ("beta", P), ("film", P), ("addend", P), ("store", P), ("tsteps", P),
                ("kind", I), ("C", I), ("groups", I), ("tmod", I), ("rows_per_step", I), ("row0", I), ("pad1", I), ("pad2", I)]


class PPOp(ctypes.Structure):
    _fields_ = [("src", PPSrc * 2), ("w", P), ("bias", P), ("out", P),
                ("type", I), ("B", I), ("Tin", I), ("Tout", I), ("Cin", I), ("Cout", I), ("K", I), ("stride", I), ("pad", I), ("ksplit", I),
                ("barrier_after", I), ("sched", I), ("pad0", I), ("pad1", I)]


class PPArgs(ctypes.Structure):
    _fields_ = [("prologue", P), ("step_ops", P), ("coef", P), ("noise", P), ("traj", P), ("amin", P), ("amax", P), ("action", P),
                ("barrier", P), ("init", P), ("err", P), ("trace", P), ("n_prologue", I), ("n_step", I), ("nsteps", I), ("mode", I), ("B", I), ("T", I), ("Da", I), ("pad0", I)]


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class PersistentDenoiser:
    MAX_B = 2              # the largest layer input (2048 channels x (4 + 2 * 2) rows) fits the LDS for two samples

    def __init__(self, eng, batch_size, timesteps, use_ddim, num_inference_steps, init, step_noise=None, nwg=None):
        """eng: PolicyEngine; timesteps: the scheduler's timestep list (descending); init: [B, T, Da] fp32 buffer the caller fills with the
        initial noise before every launch (read only); step_noise: [nsteps, B, T, Da] for the ancestral (DDPM) update.  self.traj holds the
        normalised sample afterwards, self.action the un-normalised one."""
        traj = torch.zeros_like(init)
        if ctypes.sizeof(PPOp) != lib.v2a_policy_persist_op_bytes() or ctypes.sizeof(PPArgs) != lib.v2a_policy_persist_args_bytes():
            raise RuntimeError("policy_persist: the host mirror of the op structs does not match the library")
        cfg = eng.cfg
        if batch_size > self.MAX_B:
            raise ValueError(f"the persistent denoiser holds a layer's whole input in LDS: batch <= {self.MAX_B}, got {batch_size}")
        self.eng, self.B = eng, int(batch_size)
        self.dev = traj.device
        self.T, self.Da = int(traj.shape[1]), int(traj.shape[2])
        self.nsteps = len(timesteps)
        self.traj, self.init = traj, init
        self.step_noise = step_noise
        # workgroups: half the CUs by default -- a grid barrier costs by the number of workgroups (3.7 us at 256, 2.4 at 128, measured), the
        # product needs waves (8 per workgroup), not workgroups
        self.nwg = int(nwg or max(1, torch.cuda.get_device_properties(self.dev).multi_processor_count // 2))
        self._keep = []                                       # every buffer the program points into
        self._convs = []                                      # the convs whose forward packs the program reads
        self.named = {}                                       # intermediate tensors by name (tests compare them layer by layer)
        B, T, R = self.B, self.T, self.nsteps * self.B
        f = self._buf
        Ttr = cfg.num_train_timesteps
        rows = []
        for t in timesteps:
            c = list(ddim_coeffs(eng.ac_host, t, Ttr, num_inference_steps)) if use_ddim else list(ddpm_coeffs(eng.ac_host, t, Ttr))
            if not use_ddim and t == 0:
                c[4] = 0.0                                    # no noise on the last ancestral step (scheduling_ddpm.py: `if t > 0`)
            rows.append(c)
        self.coef = torch.tensor(rows, dtype=torch.float32, device=self.dev)
        self.tsteps = torch.tensor(list(timesteps), dtype=torch.int32, device=self.dev)
        assert 32 % B == 0                                    # (row chunks of the prologue start on a sample boundary)
        self.gcond = f(B, eng.film_gd - cfg.dsed)             # the encoders' feature vector is copied here before every launch
        self.action = f(B, T, self.Da)
        self.barrier = torch.zeros(1024, dtype=torch.int32, device=self.dev)      # one flag per workgroup
        self.mode = 1 if use_ddim else 0

        # ---- prologue: step embedding and every block's FiLM rows for ALL scheduler steps at once (rows = steps x samples; a Linear is a
        # 1 x 1 conv over the row axis).  conditional_unet1d.py:206-219 (diffusion_step_encoder, cat with global_cond), :62-71 (cond_encoder)
        pro = []
        dsed = cfg.dsed
        e1, e2 = f(R, 4 * dsed), f(R, dsed)
        Gd = eng.film_gd - dsed
        RC = 32                                               # rows per op (a layer's rows must fit the LDS: 100 ancestral steps are 100-200 rows)
        chunks = [(lo, min(R, lo + RC)) for lo in range(0, R, RC)]

        def layer(make_srcs, cv, out, last_of_phase):
            for ci, (lo, hi) in enumerate(chunks):
                n = hi - lo
                pro.append(self._op(make_srcs(lo, n), cv, out[lo:hi], 1, n, n, K=1, pad=0,
                                    barrier=1 if (last_of_phase and ci == len(chunks) - 1) else 0))

        layer(lambda lo, n: [dict(kind=SRC_SINCOS, C=dsed, tsteps=self.tsteps, rows_per_step=B, row0=lo)], eng.step1, e1, True)
        layer(lambda lo, n: [dict(kind=SRC_MISH, a=e1[lo:lo + n], C=4 * dsed)], eng.step3, e2, True)
        film_of = {}                                          # residual block (by its parameter prefix) -> its FiLM rows [R, 2 * Cout]
        for j, r in enumerate(eng.film):
            film_of[r["pre"]] = f(R, 2 * r["cout"])
            layer(lambda lo, n: [dict(kind=SRC_MISH, a=e2[lo:lo + n], C=dsed), dict(kind=SRC_MISH, a=self.gcond, C=Gd, tmod=B)], r["ce"],
                  film_of[r["pre"]], j == len(eng.film) - 1)

        # ---- one scheduler step
        st = []
        k = cfg.kernel_size
        G = cfg.n_groups

        def plain(t, C):
            return dict(kind=SRC_PLAIN, a=t, C=C)

        def comp_src(c, first):
            """A residual block's output mish(gn(raw1)) + residual as a loader source; the first consumer also writes it out."""
            if first[0]:
                first[0] = False
                return dict(kind=SRC_GN_MISH, a=c["raw"], gamma=c["gamma"], beta=c["beta"], groups=G, addend=c["addend"], store=c["plain"], C=c["C"])
            return dict(kind=SRC_GN_MISH, a=c["raw"], gamma=c["gamma"], beta=c["beta"], groups=G, addend=c["addend"], C=c["C"])

        def as_sources(xs, first):
            return [comp_src(x, first) if "raw" in x else plain(x["t"], x["C"]) for x in xs]

        def block(r, xs, Tc):
            """xs: one or two inputs, each {'t': plain tensor, 'C'} or a composite {'raw', 'gamma', 'beta', 'addend', 'plain', 'C'}."""
            co = r["cout"]
            raw0, raw1 = f(B, Tc, co), f(B, Tc, co)
            first = [True]
            has_rc = r["rc"] is not None
            st.append(self._op(as_sources(xs, first), r["c0"], raw0, B, Tc, Tc, K=k, pad=k // 2, barrier=0 if has_rc else 1))
            if has_rc:
                res = f(B, Tc, co)
                st.append(self._op(as_sources(xs, first), r["rc"], res, B, Tc, Tc, K=1, pad=0, barrier=1))
            else:
                assert len(xs) == 1 and xs[0]["C"] == co
                res = xs[0]["plain"] if "raw" in xs[0] else xs[0]["t"]        # identity residual: the block's input as a plain tensor
            g0, g1 = r["pre"] + ".blocks.0.block.1", r["pre"] + ".blocks.1.block.1"
            st.append(self._op([dict(kind=SRC_GN_MISH, a=raw0, gamma=eng.P[g0 + ".weight"], beta=eng.P[g0 + ".bias"], groups=G, film=film_of[r["pre"]],
                                     C=co)], r["c1"], raw1, B, Tc, Tc, K=k, pad=k // 2, barrier=1))
            out = dict(raw=raw1, gamma=eng.P[g1 + ".weight"], beta=eng.P[g1 + ".bias"], addend=res, plain=f(B, Tc, co), C=co)
            self.named.update({r["pre"] + ".raw0": raw0, r["pre"] + ".raw1": raw1, r["pre"] + ".out": out["plain"]})
            if has_rc:
                self.named[r["pre"] + ".res"] = res
            return out

        x = dict(t=self.traj, C=self.Da)
        Tc = T
        hs = []
        for lvl in eng.down:
            x = block(lvl["r0"], [x], Tc)
            x = block(lvl["r1"], [x], Tc)
            hs.append(x)
            if lvl["ds"] is not None:
                To = (Tc + 2 - 3) // 2 + 1
                y = f(B, To, x["C"])
                st.append(self._op(as_sources([x], [True]), lvl["ds"], y, B, Tc, To, K=3, stride=2, pad=1, barrier=1))
                self.named[f"down{len(hs) - 1}.ds"] = y
                x, Tc = dict(t=y, C=x["C"]), To
        for r in eng.mid:
            x = block(r, [x], Tc)
        for lvl in eng.up:
            skip = hs.pop()
            x = block(lvl["r0"], [x, dict(t=skip["plain"], C=skip["C"])], Tc)      # (its first consumer stored it phases ago)
            x = block(lvl["r1"], [x], Tc)
            y = f(B, 2 * Tc, x["C"])
            st.append(self._op(as_sources([x], [True]), lvl["us"], y, B, Tc, 2 * Tc, K=4, stride=2, pad=1, barrier=1, type_=CONVT, cout=x["C"]))
            self.named[f"up{len(eng.up) - len(hs)}.us"] = y
            x, Tc = dict(t=y, C=x["C"]), 2 * Tc
        rawf = f(B, Tc, eng.fin0.co)
        st.append(self._op(as_sources([x], [True]), eng.fin0, rawf, B, Tc, Tc, K=k, pad=k // 2, barrier=1))
        gf = "model.final_conv.0.block.1"
        self.eps = f(B, Tc, self.Da)
        self.named.update({"final.raw": rawf, "film": [film_of[r["pre"]] for r in eng.film], "e1": e1, "e2": e2})
        st.append(self._op([dict(kind=SRC_GN_MISH, a=rawf, gamma=eng.P[gf + ".weight"], beta=eng.P[gf + ".bias"], groups=G, C=eng.fin0.co)],
                           eng.fin1, self.eps, B, Tc, Tc, K=1, pad=0, barrier=1, sched=1))
        assert Tc == T
        self.n_barriers = sum(o.barrier_after for o in pro) + self.nsteps * sum(o.barrier_after for o in st)
        self.lds = 0
        for o in pro + st:
            need = lib.v2a_policy_persist_lds_bytes(o.B, o.Tin, o.Tout, o.Cin, o.K, o.stride, o.pad, o.type)
            if need == 0:
                raise ValueError(f"policy_persist: a layer does not fit the kernel (B {o.B}, T {o.Tin}->{o.Tout}, Cin {o.Cin}, k {o.K}, stride {o.stride})")
            self.lds = max(self.lds, need)
        self._pro_dev = self._upload(pro)
        self._st_dev = self._upload(st)
        word = ctypes.c_void_p()
        check(lib.v2a_dp_errword_alloc(ctypes.byref(word)), "errword_alloc")       # pinned host ints the kernel can raise (shared helper)
        self._err = word.value
        lim = eng.act_limits
        self.args = PPArgs(prologue=_ptr(self._pro_dev), step_ops=_ptr(self._st_dev), coef=_ptr(self.coef), noise=_ptr(step_noise), traj=_ptr(traj),
                           amin=_ptr(lim[0]) if lim else 0, amax=_ptr(lim[1]) if lim else 0, action=_ptr(self.action), barrier=_ptr(self.barrier), init=_ptr(init), err=self._err, trace=0,
                           n_prologue=len(pro), n_step=len(st), nsteps=self.nsteps, mode=self.mode, B=B, T=T, Da=self.Da)
        self.n_ops = (len(pro), len(st))
        self._pack_ptrs = {id(cv): cv.pf().data_ptr() for cv in self._convs}

    def _buf(self, *shape):
        t = torch.zeros(shape, dtype=torch.float32, device=self.dev)
        self._keep.append(t)
        return t

    def _op(self, srcs, cv, out, B, Tin, Tout, K, pad, barrier, stride=1, type_=CONV, sched=0, cout=None):
        op = PPOp()
        cin = 0
        for i, s in enumerate(srcs):
            d = op.src[i]
            d.kind, d.C = s["kind"], s["C"]
            d.a, d.gamma, d.beta = _ptr(s.get("a")), _ptr(s.get("gamma")), _ptr(s.get("beta"))
            d.film, d.addend, d.store, d.tsteps = _ptr(s.get("film")), _ptr(s.get("addend")), _ptr(s.get("store")), _ptr(s.get("tsteps"))
            d.groups, d.tmod, d.rows_per_step, d.row0 = s.get("groups", 0), s.get("tmod", 0), s.get("rows_per_step", 1), s.get("row0", 0)
            for key in ("a", "gamma", "beta", "film", "addend", "store", "tsteps"):
                if s.get(key) is not None:
                    assert s[key].is_contiguous()
                    self._keep.append(s[key])
            cin += s["C"]
        w, b = cv.w, cv.b
        assert w.is_contiguous() and w.dtype == torch.float32
        co = cout if cout is not None else w.shape[0]
        if type_ == CONV:
            assert w.shape[0] == co and w.shape[1] == cin and (w.dim() == 2 or w.shape[2] == K), (tuple(w.shape), cin, K)
            wt = cv.pf()                                      # forward pack [Cout][K][Cin] (the parameter itself when K == 1), kept current by
            assert wt.numel() == w.numel()                    # the engine: PolicyEngine.refresh_packs / the optimiser's fused pack writes
            self._convs.append(cv)
        else:
            assert w.shape[0] == cin and w.shape[1] == co and w.shape[2] == K, (tuple(w.shape), cin, co)
            wt = w                                            # ConvTranspose1d: the parameter, [Cin][Cout][4]
        self._keep += [wt, b, out]
        op.w, op.bias, op.out = _ptr(wt), _ptr(b), _ptr(out)
        op.type, op.B, op.Tin, op.Tout, op.Cin, op.Cout, op.K, op.stride, op.pad = type_, B, Tin, Tout, cin, co, K, stride, pad
        waves = self.nwg * lib.v2a_policy_persist_waves_per_wg()
        ks = 1
        while ks < 4 and co * ks * 2 <= waves and cin // (ks * 2) >= 64:
            ks *= 2
        op.ksplit, op.barrier_after, op.sched = ks, barrier, sched
        return op

    def _upload(self, ops_):
        arr = (PPOp * len(ops_))(*ops_)
        t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        self._keep.append(t)
        return t

    def trace(self, global_cond):
        """One launch with workgroup 0's timeline recorded: [(op start, loader done, product done, barrier passed) in microseconds] per
        executed op, prologue first."""
        n = self.n_ops[0] + self.nsteps * self.n_ops[1]
        buf = torch.zeros((n, 4), dtype=torch.int64, device=self.dev)
        self.args.trace = buf.data_ptr()
        try:
            self.launch(global_cond)
            torch.cuda.synchronize(self.dev)
        finally:
            self.args.trace = 0
        t = buf.cpu().double()
        return ((t - t[0, 0]) / 100.0).tolist()

    def check(self):
        """Raise if a launch abandoned a grid barrier (a workgroup never became resident within 2 s: the kernel terminates instead of hanging
        the GPU, its results are garbage).  Read without synchronising; launch() looks at it before every launch."""
        v = ctypes.c_int.from_address(self._err).value
        if v:
            ctypes.c_int.from_address(self._err).value = 0          # the instance stays usable: the NEXT abandoned barrier raises again
            raise RuntimeError(f"persistent denoiser: a grid barrier (phase {v}) was abandoned after 2 s -- {self.nwg} workgroups of 512 threads "
                               f"with {self.lds} bytes of LDS were not all resident (another kernel holding the CUs?)")

    def launch(self, global_cond):
        """global_cond [B, G] (the image encoders' output).  Runs every scheduler step; afterwards self.traj holds the normalised sample and
        self.action the un-normalised one."""
        self.check()
        for cv in self._convs:
            if cv.pf().data_ptr() != self._pack_ptrs[id(cv)]:
                raise RuntimeError("a forward pack of the policy was re-allocated: build a new PersistentDenoiser")
        self.gcond.copy_(global_cond)
        check(lib.v2a_policy_persist_launch(ctypes.byref(self.args), self.nwg, self.lds, ops._stream()), "policy_persist_launch")
        return self.action
