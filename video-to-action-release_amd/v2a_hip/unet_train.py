"""Forward-with-tape and hand-written backward of the AVDC video UNet (video-model *training*, SURVEY.md section 8f rank 4:
GoalGaussianDiffusion.forward / p_losses, reference goal_diffusion.py:690-724 over UNetModel.forward, unet.py:650-684, which
the reference differentiates with autograd).  fp32 (the parity configuration); same kernels as the policy backward:

  Conv3d            spatial conv + (3x1) temporal conv            -> data gradients as convs over flipped packs (input dilation for the
                                                                     stride-2 Downsample, 2x2 sum-pool after the folded Upsample),
                                                                     weight / bias gradients by the LDS-DMA weight-gradient kernel
  GroupNorm32+SiLU  fused forward, saved (mean, rstd)              -> fused backward (dx, dgamma, dbeta); the decoder's channel concat
                                                                     is materialised in training so that one backward covers it
  ResBlock emb add  row vector in the conv epilogue                -> per-sample column sums of the conv's output gradient
  AttentionBlock    GroupNorm -> qkv -> per-frame attention -> proj -> deterministic attention backward kernel
  time embedding    Linear-SiLU-Linear on sinusoids                -> linear backward; d(label embedding) is returned to the caller

Gradients are written into caller-provided tensors `grads[name]` (torch layout).  Packed operands are rebuilt by `refresh_packs()`
(call after every optimiser step: the fused optimiser updates parameters behind torch's version counters)."""
import torch
from . import ops
from .unet_engine import UNetEngine


class UNetTrainEngine(UNetEngine):
    def __init__(self, cfg, params: dict, prefix="unet."):
        super().__init__(cfg, params, prefix)
        self._mp = None             # multi-pack table: forward / flipped operands (+ bf16 twins) of every weight, two launches per refresh
        self._fw, self._fl = {}, {}

    # ------------------------------------------------------------------ operands
    def refresh_packs(self):
        """Re-pack every conv / linear weight after the parameters changed: two multi-tensor launches write all forward packs
        ([Cout][taps][Cin]) and all data-gradient packs ([Cin][taps reversed][Cout]) into persistent buffers -- and, in the bf16 MFMA
        mode (v2a_hip.set_precision('bf16')), the bf16 twin of each, which ops.conv2d picks up to run the layer on the LDS-DMA bf16
        kernel (fp32 tensors stay in HBM, activations are rounded by a cast launch)."""
        from ._lib import lib, check
        self.packs._c.clear()
        bf16 = lib.v2a_get_precision() == 1
        ptrs = [p.data_ptr() for p in self.P.values()]
        mp = self._mp
        if mp is None or mp["bf16"] != bf16 or mp["ptrs"] != ptrs:
            rows, ch0, ch1 = [], [], []
            ce = lib.v2a_pack_chunk_elems()
            self._fw, self._fl, keep = {}, {}, []
            for full, p in self.P.items():
                if p.dim() < 2 or not full.endswith(".weight") or "pos_emb" in full:
                    continue
                w = p.detach()
                assert w.is_contiguous(), full
                co, ci = w.shape[0], w.shape[1]
                taps = w.numel() // (co * ci)
                twin = bf16 and w.numel() >= 4096
                fw = torch.empty(w.numel(), dtype=torch.float32, device=self.device) if taps > 1 else w
                fl = torch.empty(w.numel(), dtype=torch.float32, device=self.device)
                fwh = torch.empty(w.numel(), dtype=torch.bfloat16, device=self.device) if twin else None
                flh = torch.empty(w.numel(), dtype=torch.bfloat16, device=self.device) if twin else None
                if taps > 1 or twin:
                    rows.append([w.data_ptr(), fw.data_ptr() if taps > 1 else 0, co, ci, taps, 0, fwh.data_ptr() if twin else 0])
                    ch0 += [[len(rows) - 1, s0] for s0 in range(0, w.numel(), ce)]
                rows.append([w.data_ptr(), fl.data_ptr(), co, ci, taps, 1, flh.data_ptr() if twin else 0])
                ch1 += [[len(rows) - 1, t] for t in range(-(-co // 64) * -(-(ci * taps) // 64))]
                if twin:
                    ops.register_h_twin(fw, fwh)
                    ops.register_h_twin(fl, flh)
                self._fw[full], self._fl[full] = fw, fl
                keep += [fwh, flh]
            t = lambda a, dt: torch.tensor(a, dtype=dt).to(self.device)
            mp = self._mp = dict(tab=t(rows, torch.int64), ch0=t(ch0, torch.int32), n0=len(ch0), ch1=t(ch1, torch.int32), n1=len(ch1),
                                 ptrs=ptrs, bf16=bf16, keep=keep)
        check(lib.v2a_pack_weights_multi(mp["tab"].data_ptr(), mp["ch0"].data_ptr(), mp["n0"], 0, ops._stream()), "pack_weights_multi")
        check(lib.v2a_pack_weights_multi(mp["tab"].data_ptr(), mp["ch1"].data_ptr(), mp["n1"], 1, ops._stream()), "pack_weights_multi_t")

    def w(self, name, half=False):
        if half:
            return super().w(name, half)
        if self._mp is None:
            self.refresh_packs()
        return self._fw[self.pre + name]

    def wflip(self, name):
        """K-contiguous operand of the conv that computes the data gradient of `name` ([Cin][taps reversed][Cout])."""
        if self._mp is None:
            self.refresh_packs()
        return self._fl[self.pre + name]

    # ------------------------------------------------------------------ Conv3d
    def conv3d_fwd(self, x, name, cout, stride=1, ups=False, rowvec=None, residual=None):
        """Like UNetEngine.conv3d, additionally returning what the backward needs."""
        B, Fr, H, W, C = x.shape
        k = self.p(name + ".spatial_conv.weight").shape[-1]
        has_t = self.has(name + ".temporal_conv.weight")
        x4 = x.view(B * Fr, H, W, C)
        kx, ky = [], []                                          # bf16 twins of the conv inputs (bf16-MFMA mode), kept for the weight gradients
        y = ops.conv2d(x4, self.w(name + ".spatial_conv.weight"), self.p(name + ".spatial_conv.bias"), cout, k, k, (stride, stride),
                       (k // 2, k // 2), ups=ups, rowvec=None if has_t else rowvec, rows_per_batch=1,
                       residual=None if (has_t or residual is None) else residual.view(B * Fr, residual.shape[2], residual.shape[3], cout),
                       keep_h=kx, x_h=getattr(x, "_h", None))
        OH, OW = y.shape[1], y.shape[2]
        st = dict(name=name, x=x, y=y, k=k, stride=stride, ups=ups, has_t=has_t, cout=cout, has_row=rowvec is not None,
                  has_res=residual is not None, xh=kx[0] if kx else None, yh=None)
        if not has_t:
            assert rowvec is None
            return y.view(B, Fr, OH, OW, cout), st
        z = ops.conv2d(y.view(B, Fr, OH * OW, cout), self.w(name + ".temporal_conv.weight"), self.p(name + ".temporal_conv.bias"), cout,
                       3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=Fr * OH * OW,
                       residual=None if residual is None else residual.view(B, Fr, OH * OW, cout), keep_h=ky)
        st["yh"] = ky[0] if ky else None
        return z.view(B, Fr, OH, OW, cout), st

    def conv3d_bwd(self, st, dz, grads, need_dx=True):
        """dz [B,F,OH,OW,cout] -> (dx or None, d(rowvec) [B,cout] or None).  The residual's gradient is dz itself."""
        name, x, y, k, stride, ups, cout = st["name"], st["x"], st["y"], st["k"], st["stride"], st["ups"], st["cout"]
        B, Fr, H, W, C = x.shape
        OH, OW = y.shape[1], y.shape[2]
        pre = self.pre + name
        drow = None
        if st["has_row"]:
            drow = ops.colsum_batched(dz.view(B, Fr * OH * OW, cout))
        twins = ops.lib.v2a_get_precision() == 1
        if st["has_t"]:
            dz4 = dz.view(B, Fr, OH * OW, cout)
            y4 = y.view(B, Fr, OH * OW, cout)
            dzh = None                                                                 # one rounding serves data and weight gradient
            if twins and st["yh"] is not None:
                dzh = getattr(dz, "_h", None)
                dzh = ops.cast_h(dz4) if dzh is None else dzh
            ops.conv2d_wgrad(y4, dz4, (cout, cout, 3, 1), 3, 1, (1, 1), (1, 0), dw=grads[pre + ".temporal_conv.weight"],
                             dbias=grads[pre + ".temporal_conv.bias"], x_h=st["yh"], dy_h=dzh)
            dy = ops.conv2d(dz4, self.wflip(name + ".temporal_conv.weight"), None, cout, 3, 1, (1, 1), (1, 0), x_h=dzh).view(B * Fr, OH, OW, cout)
        else:
            dy = dz.view(B * Fr, OH, OW, cout)
        x4 = x.view(B * Fr, H, W, C)
        dyh = None
        if twins and st["xh"] is not None:
            dyh = getattr(dz, "_h", None) if not st["has_t"] else None
            dyh = ops.cast_h(dy) if dyh is None else dyh
        ops.conv2d_wgrad(x4, dy, (cout, C, k, k), k, k, (stride, stride), (k // 2, k // 2), ups=ups, dw=grads[pre + ".spatial_conv.weight"],
                         dbias=grads[pre + ".spatial_conv.bias"], x_h=st["xh"], dy_h=dyh)
        if not need_dx:
            return None, drow
        wf = self.wflip(name + ".spatial_conv.weight")
        if ups:
            du = ops.conv2d(dy, wf, None, C, k, k, (1, 1), (k // 2, k // 2), x_h=dyh)         # gradient at the upsampled resolution
            dx = ops.sumpool2x2(du)
        elif stride > 1:
            dx = ops.conv2d(dy, wf, None, C, k, k, (1, 1), (k // 2, k // 2), idil=stride, out_hw=(H, W), x_h=dyh)
        else:
            dx = ops.conv2d(dy, wf, None, C, k, k, (1, 1), (k // 2, k // 2), x_h=dyh)
        return dx.view(B, Fr, H, W, C), drow

    # ------------------------------------------------------------------ GroupNorm32 (+ SiLU)
    def gn_fwd(self, x, name, act="silu", frames_separate=False):
        B, Fr, H, W, C = x.shape
        N, S = (B * Fr, H * W) if frames_separate else (B, Fr * H * W)
        x3 = x.view(N, S, C)
        tw = [] if (ops.lib.v2a_get_precision() == 1 and C % 64 == 0) else None      # bf16-MFMA mode: the consuming conv reads the twin
        y, mean, rstd = ops.groupnorm_fwd(x3, self.p(name + ".weight"), self.p(name + ".bias"), 32, act, twin_out=tw)
        y = y.view(B, Fr, H, W, C)
        y._h = tw[0] if tw else None
        return y, dict(name=name, x3=x3, mean=mean, rstd=rstd, act=act, shape=tuple(x.shape))

    def gn_bwd(self, st, dy, grads):
        pre = self.pre + st["name"]
        x3 = st["x3"]
        tw = [] if (ops.lib.v2a_get_precision() == 1 and x3.shape[-1] % 64 == 0) else None
        dx, _, _, _, _ = ops.groupnorm_bwd(x3, self.p(st["name"] + ".weight"), self.p(st["name"] + ".bias"), 32, dy.view(x3.shape),
                                           st["mean"], st["rstd"], st["act"], dgamma=grads[pre + ".weight"], dbeta=grads[pre + ".bias"],
                                           twin_out=tw)
        dx = dx.view(st["shape"])
        dx._h = tw[0] if tw else None
        return dx

    # ------------------------------------------------------------------ Linear
    def lin_fwd(self, x2d, name):
        return ops.linear(x2d, self.p(name + ".weight"), self.p(name + ".bias"))

    def lin_bwd(self, x2d, name, dy2d, grads, need_dx=True):
        pre = self.pre + name
        w = self.p(name + ".weight")
        M = x2d.shape[0]
        ops.conv2d_wgrad(x2d.view(1, 1, M, -1), dy2d.view(1, 1, M, -1), tuple(w.shape), 1, 1, dw=grads[pre + ".weight"],
                         dbias=grads.get(pre + ".bias") if self.has(name + ".bias") else None)
        if not need_dx:
            return None
        return ops.conv2d(dy2d.view(1, 1, M, -1), self.wflip(name + ".weight"), None, w.shape[1], 1, 1).view(M, w.shape[1])

    # ------------------------------------------------------------------ blocks
    def res_fwd(self, x, name, cout, semb):
        a, s_gn0 = self.gn_fwd(x, name + ".in_layers.0")
        eo = self.lin_fwd(semb, name + ".emb_layers.1")
        h, s_c0 = self.conv3d_fwd(a, name + ".in_layers.2", cout, rowvec=eo)
        a2, s_gn1 = self.gn_fwd(h, name + ".out_layers.0")
        s_skip = None
        if self.has(name + ".skip_connection.spatial_conv.weight"):
            xs, s_skip = self.conv3d_fwd(x, name + ".skip_connection", cout)
        else:
            xs = x
        out, s_c1 = self.conv3d_fwd(a2, name + ".out_layers.3", cout, residual=xs)
        return out, dict(kind="res", name=name, gn0=s_gn0, c0=s_c0, gn1=s_gn1, skip=s_skip, c1=s_c1, semb=semb)

    def res_bwd(self, st, dout, grads, dsemb):
        """Returns dx; accumulates d(SiLU(emb)) into dsemb [B, 4*mc]."""
        name = st["name"]
        da2, _ = self.conv3d_bwd(st["c1"], dout, grads)
        dh = self.gn_bwd(st["gn1"], da2, grads)
        da, deo = self.conv3d_bwd(st["c0"], dh, grads)
        d_semb = self.lin_bwd(st["semb"], name + ".emb_layers.1", deo, grads)
        ops.axpy(d_semb, dsemb, out=dsemb)
        dx = self.gn_bwd(st["gn0"], da, grads)
        if st["skip"] is not None:
            dxs, _ = self.conv3d_bwd(st["skip"], dout, grads)
        else:
            dxs = dout
        return ops.axpy(dx, dxs)

    def attn_fwd(self, x, name, C):
        B, Fr, H, W, _ = x.shape
        N, L = B * Fr, H * W
        hc = self.cfg.num_head_channels
        heads = C // hc
        n, s_gn = self.gn_fwd(x, name + ".norm", act="none", frames_separate=True)
        n2 = n.view(N * L, C)
        wq = self.p(name + ".qkv.weight").view(3 * C, C)
        qkv = ops.linear(n2, wq, self.p(name + ".qkv.bias"))
        a = ops.attention(qkv, N, L, heads, hc)
        wo = self.p(name + ".proj_out.weight").view(C, C)
        out = ops.linear(a, wo, self.p(name + ".proj_out.bias"), residual=x.view(N * L, C))
        return out.view(B, Fr, H, W, C), dict(kind="attn", name=name, gn=s_gn, n2=n2, qkv=qkv, a=a, dims=(N, L, heads, hc, C),
                                              shape=tuple(x.shape))

    def attn_bwd(self, st, dout, grads):
        name = st["name"]
        N, L, heads, hc, C = st["dims"]
        pre = self.pre + name
        d2 = dout.view(N * L, C)
        M = N * L
        ops.conv2d_wgrad(st["a"].view(1, 1, M, C), d2.view(1, 1, M, C), (C, C, 1, 1), 1, 1, dw=grads[pre + ".proj_out.weight"],
                         dbias=grads[pre + ".proj_out.bias"])
        da = ops.conv2d(d2.view(1, 1, M, C), self.wflip(name + ".proj_out.weight"), None, C, 1, 1).view(M, C)
        dqkv = ops.attention_bwd(st["qkv"], st["a"], da, N, L, heads, hc)
        ops.conv2d_wgrad(st["n2"].view(1, 1, M, C), dqkv.view(1, 1, M, 3 * C), (3 * C, C, 1, 1), 1, 1, dw=grads[pre + ".qkv.weight"],
                         dbias=grads[pre + ".qkv.bias"])
        dn = ops.conv2d(dqkv.view(1, 1, M, 3 * C), self.wflip(name + ".qkv.weight"), None, C, 1, 1).view(st["shape"])
        dx = self.gn_bwd(st["gn"], dn, grads)
        return ops.axpy(dx, dout)

    # ------------------------------------------------------------------ text branch (PerceiverResampler + Linear + mean)
    def _ln_fwd(self, x2d, wname, bname=None):
        return ops.layernorm(x2d, self.p(wname), None if bname is None else self.p(bname))

    def _ln_bwd(self, x2d, wname, bname, dy2d, grads):
        dx, dg, db = ops.layernorm_bwd(x2d, self.p(wname), dy2d)
        grads[self.pre + wname].copy_(dg.view_as(grads[self.pre + wname]))
        if bname is not None:
            grads[self.pre + bname].copy_(db.view_as(grads[self.pre + bname]))
        return dx

    def label_embedding_train(self, y):
        """Same arithmetic as UNetEngine.label_embedding, keeping what the backward needs."""
        cfg = self.cfg
        pre = "task_attnpool.0"
        B, L, D = y.shape
        y = y.float().contiguous()
        pos = self.p(pre + ".pos_emb.weight")
        xp = torch.empty_like(y)
        for b in range(B):
            ops.axpy(y[b], pos[:L], 1.0, out=xp[b])
        n_lat, n_mp = cfg.pr_num_latents, cfg.pr_num_mean_pooled
        NL = n_lat + n_mp
        lat = torch.empty((B, NL, D), dtype=torch.float32, device=y.device)
        lp = self.p(pre + ".latents")
        for b in range(B):
            ops.copy2d(lp, lat[b], n_lat, D, D, D, dst_off=n_mp * D)
        tape = dict(B=B, L=L, D=D, NL=NL, n_mp=n_mp, n_lat=n_lat, layers=[])
        if n_mp > 0:
            mp0 = ops.mean_rows(y)
            mp1 = self._ln_fwd(mp0, pre + ".to_latents_from_mean_pooled_seq.0.g")
            mp = self.lin_fwd(mp1, pre + ".to_latents_from_mean_pooled_seq.1")
            ops.copy2d(mp, lat, B, n_mp * D, n_mp * D, NL * D)
            tape.update(mp0=mp0, mp1=mp1)
        H, dh = cfg.pr_heads, cfg.pr_dim_head
        xp2 = xp.view(B * L, D)
        for li in range(cfg.pr_depth):
            a = f"{pre}.layers.{li}.0"
            f = f"{pre}.layers.{li}.1"
            lat_in = lat.view(B * NL, D)
            xn = self._ln_fwd(xp2, a + ".norm.weight", a + ".norm.bias")
            ln = self._ln_fwd(lat_in, a + ".norm_latents.weight", a + ".norm_latents.bias")
            q = ops.linear(ln, self.p(a + ".to_q.weight"))
            kvin = torch.empty((B, L + NL, D), dtype=torch.float32, device=y.device)
            ops.copy2d(xn, kvin, B, L * D, L * D, (L + NL) * D)
            ops.copy2d(ln, kvin, B, NL * D, NL * D, (L + NL) * D, dst_off=L * D)
            kv = ops.linear(kvin.view(B * (L + NL), D), self.p(a + ".to_kv.weight"))
            o0 = ops.perceiver_attention(q, kv, self.p(a + ".q_scale"), self.p(a + ".k_scale"), B, NL, L + NL, H, dh, 8.0)
            o1 = ops.linear(o0.view(B * NL, H * dh), self.p(a + ".to_out.0.weight"))
            o2 = self._ln_fwd(o1, a + ".to_out.1.weight", a + ".to_out.1.bias")
            lat_mid = ops.axpy(o2, lat_in)
            f0 = self._ln_fwd(lat_mid, f + ".0.g")
            f1 = ops.linear(f0, self.p(f + ".1.weight"))
            f2 = ops.act_fwd(f1, "gelu")
            f3 = self._ln_fwd(f2, f + ".3.g")
            f4 = ops.linear(f3, self.p(f + ".4.weight"))
            lat = ops.axpy(f4, lat_mid).view(B, NL, D)
            tape["layers"].append(dict(a=a, f=f, lat_in=lat_in, xn=xn, ln=ln, q=q, kvin=kvin, kv=kv, o0=o0, o1=o1, lat_mid=lat_mid, f0=f0,
                                       f1=f1, f2=f2, f3=f3))
        lat2 = lat.view(B * NL, D)
        z = self.lin_fwd(lat2, "task_attnpool.1")
        tape.update(xp2=xp2, lat_out=lat2, H=H, dh=dh)
        return ops.mean_rows(z.view(B, NL, -1)), tape

    def label_embedding_bwd(self, tape, dlab, grads):
        """dlab [B, 4mc] -> parameter gradients of task_attnpool.* (the token features themselves are inputs: no gradient)."""
        B, L, D, NL, n_mp, n_lat, H, dh = (tape[k] for k in ("B", "L", "D", "NL", "n_mp", "n_lat", "H", "dh"))
        pre = "task_attnpool.0"
        dz = ops.bcast_rows(dlab, NL, 1.0 / NL).view(B * NL, -1)
        dlat = self.lin_bwd(tape["lat_out"], "task_attnpool.1", dz, grads)                # [B*NL, D]
        dxp = torch.zeros((B * L, D), dtype=torch.float32, device=dlab.device)
        for st in reversed(tape["layers"]):
            a, f = st["a"], st["f"]
            # feed-forward: lat = lat_mid + W4 ln_g(gelu(W1 ln_g(lat_mid)))
            df3 = self._lin_nobias_bwd(st["f3"], f + ".4.weight", dlat, grads)
            df2 = self._ln_bwd(st["f2"], f + ".3.g", None, df3, grads)
            df1 = ops.act_bwd(st["f1"], df2, "gelu")
            df0 = self._lin_nobias_bwd(st["f0"], f + ".1.weight", df1, grads)
            dmid = ops.axpy(self._ln_bwd(st["lat_mid"], f + ".0.g", None, df0, grads), dlat)
            # attention: lat_mid = lat_in + LN(Wo attn(Wq LN(lat_in), Wkv [LN(x) ; LN(lat_in)]))
            do1 = self._ln_bwd(st["o1"], a + ".to_out.1.weight", a + ".to_out.1.bias", dmid, grads)
            do0 = self._lin_nobias_bwd(st["o0"].view(B * NL, H * dh), a + ".to_out.0.weight", do1, grads)
            dq, dkv, dqs, dks = ops.perceiver_attention_bwd(st["q"], st["kv"], self.p(a + ".q_scale"), self.p(a + ".k_scale"), st["o0"],
                                                            do0.view(B, NL, H * dh), B, NL, L + NL, H, dh, 8.0)
            grads[self.pre + a + ".q_scale"].copy_(dqs)
            grads[self.pre + a + ".k_scale"].copy_(dks)
            dkvin = self._lin_nobias_bwd(st["kvin"].view(B * (L + NL), D), a + ".to_kv.weight", dkv.view(B * (L + NL), -1), grads)
            dln = self._lin_nobias_bwd(st["ln"], a + ".to_q.weight", dq.view(B * NL, -1), grads)
            dkvin3 = dkvin.view(B, L + NL, D)
            dxn = torch.empty((B, L, D), dtype=torch.float32, device=dlab.device)
            dln_kv = torch.empty((B, NL, D), dtype=torch.float32, device=dlab.device)
            ops.copy2d(dkvin3, dxn, B, L * D, (L + NL) * D, L * D)
            ops.copy2d(dkvin3, dln_kv, B, NL * D, (L + NL) * D, NL * D, src_off=L * D)
            dln = ops.axpy(dln, dln_kv.view(B * NL, D))
            dlat_in = self._ln_bwd(st["lat_in"], a + ".norm_latents.weight", a + ".norm_latents.bias", dln, grads)
            dxp_l = self._ln_bwd(tape["xp2"], a + ".norm.weight", a + ".norm.bias", dxn.view(B * L, D), grads)
            ops.axpy(dxp_l, dxp, out=dxp)
            dlat = ops.axpy(dlat_in, dmid)
        # initial latents: [mean-pooled branch | learned latents], positional embedding
        dlat3 = dlat.view(B, NL, D)
        gl = grads[self.pre + pre + ".latents"]
        tmp = torch.empty((B, n_lat * D), dtype=torch.float32, device=dlab.device)
        ops.copy2d(dlat3, tmp, B, n_lat * D, NL * D, n_lat * D, src_off=n_mp * D)
        gl.copy_(ops.colsum(tmp).view_as(gl))
        if n_mp > 0:
            dmp = torch.empty((B, n_mp * D), dtype=torch.float32, device=dlab.device)
            ops.copy2d(dlat3, dmp, B, n_mp * D, NL * D, n_mp * D)
            dmp1 = self.lin_bwd(tape["mp1"], pre + ".to_latents_from_mean_pooled_seq.1", dmp, grads)
            self._ln_bwd(tape["mp0"], pre + ".to_latents_from_mean_pooled_seq.0.g", None, dmp1, grads)
        gp = grads[self.pre + pre + ".pos_emb.weight"]
        gp.zero_()
        dpos = ops.colsum(dxp.view(B, L * D))                         # sum over the batch of d(x + pos)
        gp[:L].copy_(dpos.view(L, D))

    def _lin_nobias_bwd(self, x2d, wname, dy2d, grads):
        """Linear without bias given by its weight name; returns dx."""
        w = self.p(wname)
        M = x2d.shape[0]
        ops.conv2d_wgrad(x2d.view(1, 1, M, -1), dy2d.view(1, 1, M, -1), tuple(w.shape), 1, 1, dw=grads[self.pre + wname])
        return ops.conv2d(dy2d.view(1, 1, M, -1), self.wflip(wname), None, w.shape[1], 1, 1).view(M, w.shape[1])

    # ------------------------------------------------------------------ whole model
    def _run_fwd(self, blk, h, semb, tape, skip=None):
        for op in blk:
            kind, name = op[0], op[1]
            if kind == "conv":
                h, st = self.conv3d_fwd(h, name, op[3])
                tape.append(dict(kind="conv", c=st))
            elif kind == "res":
                csplit = None
                if skip is not None:                             # decoder: normalise / convolve cat[h, skip]: materialised here
                    B, Fr, H, W, C1 = h.shape
                    C2 = skip.shape[-1]
                    cat = torch.empty((B, Fr, H, W, C1 + C2), dtype=torch.float32, device=h.device)
                    rows = B * Fr * H * W
                    ops.copy2d(h, cat, rows, C1, C1, C1 + C2)
                    ops.copy2d(skip, cat, rows, C2, C2, C1 + C2, dst_off=C1)
                    h, csplit, skip = cat, (C1, C2), None
                h, st = self.res_fwd(h, name, op[3], semb)
                st["csplit"] = csplit
                tape.append(st)
            elif kind == "attn":
                h, st = self.attn_fwd(h, name, op[2])
                tape.append(st)
            elif kind == "down":
                h, st = self.conv3d_fwd(h, name + ".op", op[2], stride=2)
                tape.append(dict(kind="conv", c=st))
            elif kind == "up":
                h, st = self.conv3d_fwd(h, name + ".conv", op[2], ups=True)
                tape.append(dict(kind="conv", c=st))
        return h

    def forward_train(self, xin, t_long, label_emb):
        """xin [B,F,H,W,Cin] channels-last fp32, t [B] int64, label_emb [B,4mc] -> (out [B,F,H,W,Cout], tape)."""
        cfg = self.cfg
        e0 = ops.sincos_embed(t_long, cfg.model_channels, 1)
        e1 = self.lin_fwd(e0, "time_embed.0")
        e1a = ops.act_fwd(e1, "silu")
        e2 = self.lin_fwd(e1a, "time_embed.2")
        emb = ops.axpy(e2, label_emb)
        semb = ops.act_fwd(emb, "silu")
        tape = dict(e0=e0, e1=e1, e1a=e1a, emb=emb, blocks=[], marks=[])
        hs = []
        h = xin
        for blk in self.inp:
            h = self._run_fwd(blk, h, semb, tape["blocks"])
            hs.append(h)
            tape["marks"].append(("push", len(tape["blocks"])))
        h = self._run_fwd(self.mid, h, semb, tape["blocks"])
        for blk in self.out:
            tape["marks"].append(("pop", len(tape["blocks"])))
            h = self._run_fwd(blk, h, semb, tape["blocks"], skip=hs.pop())
        a, s_gn = self.gn_fwd(h, "out.0")
        out, s_c = self.conv3d_fwd(a, "out.2", cfg.out_channels)
        tape.update(out_gn=s_gn, out_c=s_c, semb_shape=tuple(semb.shape))
        return out, tape

    def forward_train_tokens(self, xin, t_long, tokens):
        """forward_train with the text branch inside the tape: tokens [B,L,512] (CLIP features)."""
        lab, ttape = self.label_embedding_train(tokens)
        out, tape = self.forward_train(xin, t_long, lab)
        tape["text"] = ttape
        return out, tape

    def backward(self, tape, dout, grads, on_decoder_done=None):
        """dout [B,F,H,W,Cout] -> d(label_emb) [B,4mc]; parameter gradients land in grads[full parameter name] (including the
        text branch's when the tape came from forward_train_tokens)."""
        da, _ = self.conv3d_bwd(tape["out_c"], dout, grads)
        dh = self.gn_bwd(tape["out_gn"], da, grads)
        dsemb = torch.zeros(tape["semb_shape"], dtype=torch.float32, device=dout.device)
        blocks = tape["blocks"]
        # skip gradients: every decoder block pops one encoder activation; its gradient joins that activation's gradient when the
        # backward walk reaches the point where it was pushed (marks record tape positions of pushes / pops in forward order)
        pops = [pos for kind, pos in tape["marks"] if kind == "pop"]
        pushes = [pos for kind, pos in tape["marks"] if kind == "push"]
        pending = []                                     # skip gradients in the order the backward produces them
        first_dec = min(pops) if pops else len(blocks)  # tape position of the first decoder block
        i = len(blocks) - 1
        while i >= 0:
            if i == first_dec - 1 and on_decoder_done is not None:
                on_decoder_done()                        # out.* and output_blocks.* gradients are final (data parallel: their slice leaves now)
                on_decoder_done = None
            st = blocks[i]
            if st["kind"] == "conv":
                first = (i == 0)
                dh, _ = self.conv3d_bwd(st["c"], dh, grads, need_dx=not first)
            elif st["kind"] == "attn":
                dh = self.attn_bwd(st, dh, grads)
            else:
                dh = self.res_bwd(st, dh, grads, dsemb)
                if st["csplit"] is not None:
                    C1, C2 = st["csplit"]
                    B, Fr, H, W, C = dh.shape
                    rows = B * Fr * H * W
                    d1 = torch.empty((B, Fr, H, W, C1), dtype=torch.float32, device=dh.device)
                    d2 = torch.empty((B, Fr, H, W, C2), dtype=torch.float32, device=dh.device)
                    ops.copy2d(dh, d1, rows, C1, C, C1)
                    ops.copy2d(dh, d2, rows, C2, C, C2, src_off=C1)
                    dh = d1
                    pending.append(d2)
            # an encoder activation was pushed right after block i-1 .. i => add the matching skip gradient (LIFO: the last pushed
            # activation is consumed by the first decoder block, whose gradient was produced last)
            while pushes and pushes[-1] == i:
                pushes.pop()
                if dh is not None:
                    dh = ops.axpy(dh, pending.pop())
                else:
                    pending.pop()
            i -= 1
        assert not pending and not pushes, (len(pending), pushes)
        demb = ops.act_bwd(tape["emb"], dsemb, "silu")
        de1a = self.lin_bwd(tape["e1a"], "time_embed.2", demb, grads)
        de1 = ops.act_bwd(tape["e1"], de1a, "silu")
        self.lin_bwd(tape["e0"], "time_embed.0", de1, grads, need_dx=False)
        if "text" in tape:
            self.label_embedding_bwd(tape["text"], demb, grads)
        return demb
