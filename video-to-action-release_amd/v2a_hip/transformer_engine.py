"""TransformerForDiffusion on HIP kernels: forward with tape + hand-written backward (the Transformer policy
backbone named in north_star; reference flowdiffusion/flowdiffusion/diffusion_policy_baseline/transformer_for_diffusion.py:23-358).

The reference builds the trunk from torch.nn.TransformerEncoder / TransformerDecoder layers (pre-norm, GELU, batch_first, additive
float masks); their arithmetic per layer is

    encoder layer   x += SA(LN1(x));                      x += W2 gelu(W1 LN2(x))
    decoder layer   x += SA(LN1(x), tgt_mask);  x += CA(LN2(x), memory, memory_mask);  x += W2 gelu(W1 LN3(x))
    SA / CA         in_proj (packed q|k|v rows of one [3E,E] weight) -> per-head softmax(q k^T / sqrt(D) + mask) v -> out_proj

Every Linear is the 1x1 case of the conv kernels (bias in the epilogue), LayerNorm / GELU / Mish / attention are their own kernels;
the backward recomputes attention probabilities from q, k, v.  Dropout (training mode, p > 0) uses stateless masks: every dropout
site of a forward call gets its own random stream id, element i is kept iff hash(seed, stream, i) >= p, and the backward re-evaluates
the same decisions -- nothing is stored (torch's generator stream cannot be reproduced on the device either way)."""
import torch
from . import ops


class TransformerEngine:
    def __init__(self, cfg: dict, params: dict, buffers: dict):
        self.cfg, self.P, self.Bf = cfg, params, buffers
        self.device = next(iter(params.values())).device
        self._flip = {}
        self._drop = None           # (p_emb, p_attn, seed) while a training-mode forward with dropout runs, else None
        self._sid = 0               # next random stream id (unique per dropout site per call)

    # ------------------------------------------------------------------ dropout sites
    def _next_sid(self):
        self._sid += 1
        return self._sid

    def drop(self, x, which, st, key):
        """Forward of one dropout site (`which`: 0 = embedding rate, 1 = layer rate); records its stream id in st[key]."""
        if self._drop is None or self._drop[which] <= 0.0:
            st[key] = None
            return x
        sid = self._next_sid()
        st[key] = sid
        return ops.dropout(x, self._drop[which], self._drop[2], sid)

    def drop_bwd(self, dy, which, st, key, dcfg):
        sid = st.get(key)
        if sid is None:
            return dy
        return ops.dropout(dy, dcfg[which], dcfg[2], sid)

    def _attn_drop(self, st):
        """(p, seed, stream id) of an attention-probability dropout site, recorded in st["adrop"]."""
        if self._drop is None or self._drop[1] <= 0.0:
            st["adrop"] = (0.0, 0, 0)
        else:
            st["adrop"] = (self._drop[1], self._drop[2], self._next_sid())
        return st["adrop"]

    def refresh_packs(self):
        self._flip.clear()

    # ------------------------------------------------------------------ Linear on (a row range of) a parameter
    def _w(self, name, rows=None):
        w = self.P[name].detach()
        return w if rows is None else w[rows[0]:rows[1]]

    def _wflip(self, name, rows=None):
        key = (name, rows)
        pk = self._flip.get(key)
        if pk is None:
            w = self._w(name, rows).contiguous()
            pk = ops.pack_weight(w.view(w.shape[0], w.shape[1], 1, 1), 1)
            self._flip[key] = pk
        return pk

    def lin(self, x2d, wname, bname=None, rows=None):
        return ops.linear(x2d, self._w(wname, rows), None if bname is None else self._w(bname, rows))

    def lin_bwd(self, x2d, wname, bname, dy2d, grads, rows=None, need_dx=True):
        w = self._w(wname, rows)
        M = x2d.shape[0]
        gw = grads[wname] if rows is None else grads[wname][rows[0]:rows[1]]
        gb = None
        if bname is not None:
            gb = grads[bname] if rows is None else grads[bname][rows[0]:rows[1]]
        ops.conv2d_wgrad(x2d.view(1, 1, M, -1), dy2d.view(1, 1, M, -1), tuple(w.shape), 1, 1, dw=gw, dbias=gb)
        if not need_dx:
            return None
        return ops.conv2d(dy2d.view(1, 1, M, -1), self._wflip(wname, rows), None, w.shape[1], 1, 1).view(M, w.shape[1])

    # ------------------------------------------------------------------ LayerNorm (weight + bias)
    def ln(self, x2d, pre):
        return ops.layernorm(x2d, self.P[pre + ".weight"].detach(), self.P[pre + ".bias"].detach())

    def ln_bwd(self, x2d, pre, dy2d, grads):
        dx, dg, db = ops.layernorm_bwd(x2d, self.P[pre + ".weight"].detach(), dy2d)
        grads[pre + ".weight"].copy_(dg)
        grads[pre + ".bias"].copy_(db)
        return dx

    # ------------------------------------------------------------------ attention blocks
    def self_attn(self, h2d, pre, B, T, mask):
        E, H = self.cfg["n_emb"], self.cfg["n_head"]
        qkv = self.lin(h2d, pre + ".in_proj_weight", pre + ".in_proj_bias")
        st = dict(h=h2d, qkv=qkv, mask=mask, T=T)
        o = ops.mha_fwd(qkv, qkv, qkv, mask, B, T, T, H, E // H, 0, E, 2 * E, *self._attn_drop(st))
        st["o"] = o
        return self.lin(o, pre + ".out_proj.weight", pre + ".out_proj.bias"), st

    def self_attn_bwd(self, st, pre, dy, grads, B):
        E, H, T = self.cfg["n_emb"], self.cfg["n_head"], st["T"]
        do = self.lin_bwd(st["o"], pre + ".out_proj.weight", pre + ".out_proj.bias", dy, grads)
        dqkv = torch.empty_like(st["qkv"])
        ops.mha_bwd(st["qkv"], st["qkv"], st["qkv"], st["mask"], do, dqkv, dqkv, dqkv, B, T, T, H, E // H, 0, E, 2 * E, *st["adrop"])
        return self.lin_bwd(st["h"], pre + ".in_proj_weight", pre + ".in_proj_bias", dqkv, grads)

    def cross_attn(self, h2d, mem2d, pre, B, T, S, mask):
        E, H = self.cfg["n_emb"], self.cfg["n_head"]
        q = self.lin(h2d, pre + ".in_proj_weight", pre + ".in_proj_bias", rows=(0, E))
        kv = self.lin(mem2d, pre + ".in_proj_weight", pre + ".in_proj_bias", rows=(E, 3 * E))
        st = dict(h=h2d, mem=mem2d, q=q, kv=kv, mask=mask, T=T, S=S)
        o = ops.mha_fwd(q, kv, kv, mask, B, T, S, H, E // H, 0, 0, E, *self._attn_drop(st))
        st["o"] = o
        return self.lin(o, pre + ".out_proj.weight", pre + ".out_proj.bias"), st

    def cross_attn_bwd(self, st, pre, dy, grads, B):
        """-> (d h, d memory)"""
        E, H = self.cfg["n_emb"], self.cfg["n_head"]
        do = self.lin_bwd(st["o"], pre + ".out_proj.weight", pre + ".out_proj.bias", dy, grads)
        dq, dkv = torch.empty_like(st["q"]), torch.empty_like(st["kv"])
        ops.mha_bwd(st["q"], st["kv"], st["kv"], st["mask"], do, dq, dkv, dkv, B, st["T"], st["S"], H, E // H, 0, 0, E, *st["adrop"])
        dh = self.lin_bwd(st["h"], pre + ".in_proj_weight", pre + ".in_proj_bias", dq, grads, rows=(0, E))
        dmem = self.lin_bwd(st["mem"], pre + ".in_proj_weight", pre + ".in_proj_bias", dkv, grads, rows=(E, 3 * E))
        return dh, dmem

    def ff(self, h2d, pre):
        f1 = self.lin(h2d, pre + ".linear1.weight", pre + ".linear1.bias")
        st = dict(h=h2d, f1=f1)
        f2 = self.drop(ops.act_fwd(f1, "gelu"), 1, st, "d_act")
        st["f2"] = f2
        return self.lin(f2, pre + ".linear2.weight", pre + ".linear2.bias"), st

    def ff_bwd(self, st, pre, dy, grads):
        df2 = self.drop_bwd(self.lin_bwd(st["f2"], pre + ".linear2.weight", pre + ".linear2.bias", dy, grads), 1, st, "d_act", self._dcfg)
        df1 = ops.act_bwd(st["f1"], df2, "gelu")
        return self.lin_bwd(st["h"], pre + ".linear1.weight", pre + ".linear1.bias", df1, grads)

    # ------------------------------------------------------------------ layers (norm_first)
    def enc_layer(self, x, pre, B, T, mask):
        h1 = self.ln(x, pre + ".norm1")
        st = dict(x=x)
        a, sa = self.self_attn(h1, pre + ".self_attn", B, T, mask)
        x1 = ops.axpy(self.drop(a, 1, st, "d1"), x)
        h2 = self.ln(x1, pre + ".norm2")
        f, sf = self.ff(h2, pre)
        out = ops.axpy(self.drop(f, 1, st, "d2"), x1)
        st.update(x1=x1, sa=sa, ff=sf)
        return out, st

    def enc_layer_bwd(self, st, pre, dy, grads, B):
        dh2 = self.ff_bwd(st["ff"], pre, self.drop_bwd(dy, 1, st, "d2", self._dcfg), grads)
        dx1 = ops.axpy(self.ln_bwd(st["x1"], pre + ".norm2", dh2, grads), dy)
        dh1 = self.self_attn_bwd(st["sa"], pre + ".self_attn", self.drop_bwd(dx1, 1, st, "d1", self._dcfg), grads, B)
        return ops.axpy(self.ln_bwd(st["x"], pre + ".norm1", dh1, grads), dx1)

    def dec_layer(self, x, mem, pre, B, T, S, mask, mem_mask):
        h1 = self.ln(x, pre + ".norm1")
        st = dict(x=x)
        a, sa = self.self_attn(h1, pre + ".self_attn", B, T, mask)
        x1 = ops.axpy(self.drop(a, 1, st, "d1"), x)
        h2 = self.ln(x1, pre + ".norm2")
        c, ca = self.cross_attn(h2, mem, pre + ".multihead_attn", B, T, S, mem_mask)
        x2 = ops.axpy(self.drop(c, 1, st, "d2"), x1)
        h3 = self.ln(x2, pre + ".norm3")
        f, sf = self.ff(h3, pre)
        out = ops.axpy(self.drop(f, 1, st, "d3"), x2)
        st.update(x1=x1, x2=x2, sa=sa, ca=ca, ff=sf)
        return out, st

    def dec_layer_bwd(self, st, pre, dy, grads, B, dmem):
        dh3 = self.ff_bwd(st["ff"], pre, self.drop_bwd(dy, 1, st, "d3", self._dcfg), grads)
        dx2 = ops.axpy(self.ln_bwd(st["x2"], pre + ".norm3", dh3, grads), dy)
        dh2, dm = self.cross_attn_bwd(st["ca"], pre + ".multihead_attn", self.drop_bwd(dx2, 1, st, "d2", self._dcfg), grads, B)
        ops.axpy(dm, dmem, out=dmem)
        dx1 = ops.axpy(self.ln_bwd(st["x1"], pre + ".norm2", dh2, grads), dx2)
        dh1 = self.self_attn_bwd(st["sa"], pre + ".self_attn", self.drop_bwd(dx1, 1, st, "d1", self._dcfg), grads, B)
        return ops.axpy(self.ln_bwd(st["x"], pre + ".norm1", dh1, grads), dx1)

    # ------------------------------------------------------------------ whole model
    def _add_pos(self, tok3d, pos_name, n):
        B, T, E = tok3d.shape
        ops.copy2d(self.P[pos_name].detach(), tok3d, B, n * E, 0, T * E, accumulate=True)     # broadcast over the batch (ld_src = 0)
        return tok3d

    def forward(self, sample, t_long, cond, drop=None):
        """sample [B,T,input_dim], t [B] int64, cond [B,To,cond_dim] or None -> (out [B,T,output_dim], tape).
        drop = (p_emb, p_attn, seed) in training mode with dropout, else None."""
        c = self.cfg
        self._drop = drop if (drop is not None and max(drop[0], drop[1]) > 0.0) else None
        E = c["n_emb"]
        B, T, Din = sample.shape
        sample = sample.float().contiguous()
        temb = ops.sincos_embed(t_long, E, 0)                                                   # [B, E]
        inp = self.lin(sample.view(B * T, Din), "input_emb.weight", "input_emb.bias")
        mask = self.Bf.get("mask")
        tape = dict(B=B, T=T, sample=sample, layers=[], enc=[], dcfg=self._drop)
        if c["encoder_only"]:
            Tt = T + 1
            x = torch.empty((B, Tt, E), dtype=torch.float32, device=sample.device)
            ops.copy2d(temb, x, B, E, E, Tt * E)
            ops.copy2d(inp, x, B, T * E, T * E, Tt * E, dst_off=E)
            x = self.drop(self._add_pos(x, "pos_emb", Tt).view(B * Tt, E), 0, tape, "d_tok")
            for li in range(c["n_layer"]):
                x, st = self.enc_layer(x, f"encoder.layers.{li}", B, Tt, mask)
                tape["layers"].append(st)
            body = torch.empty((B, T, E), dtype=torch.float32, device=sample.device)
            ops.copy2d(x, body, B, T * E, Tt * E, T * E, src_off=E)
            x = body.view(B * T, E)
        else:
            To = 0 if cond is None else cond.shape[1]
            S = 1 + To
            ce = torch.empty((B, S, E), dtype=torch.float32, device=sample.device)
            ops.copy2d(temb, ce, B, E, E, S * E)
            if To:
                cond = cond.float().contiguous()
                co = self.lin(cond.view(B * To, -1), "cond_obs_emb.weight", "cond_obs_emb.bias")
                ops.copy2d(co, ce, B, To * E, To * E, S * E, dst_off=E)
                tape["cond"] = cond
            m = self.drop(self._add_pos(ce, "cond_pos_emb", S).view(B * S, E), 0, tape, "d_cond")
            if c["n_cond_layers"] > 0:
                for li in range(c["n_cond_layers"]):
                    m, st = self.enc_layer(m, f"encoder.layers.{li}", B, S, None)
                    tape["enc"].append(st)
            else:
                e1 = self.lin(m, "encoder.0.weight", "encoder.0.bias")
                e2 = ops.act_fwd(e1, "mish")
                tape["enc_mlp"] = dict(m=m, e1=e1, e2=e2)
                m = self.lin(e2, "encoder.2.weight", "encoder.2.bias")
            x = self.drop(self._add_pos(inp.view(B, T, E), "pos_emb", T).view(B * T, E), 0, tape, "d_tok")
            mem_mask = self.Bf.get("memory_mask")
            for li in range(c["n_layer"]):
                x, st = self.dec_layer(x, m, f"decoder.layers.{li}", B, T, S, mask, mem_mask)
                tape["layers"].append(st)
            tape.update(S=S, To=To, mem=m)
        hf = self.ln(x, "ln_f")
        out = self.lin(hf, "head.weight", "head.bias")
        tape.update(xf=x, hf=hf)
        self._drop = None
        return out.view(B, T, -1), tape

    def backward(self, tape, dout, grads, need_dsample=True, need_dcond=True):
        """dout [B,T,output_dim] -> (d sample or None, d cond or None); parameter gradients are written into grads[name]."""
        c = self.cfg
        E = c["n_emb"]
        B, T = tape["B"], tape["T"]
        dout = dout.float().contiguous().view(B * T, -1)
        self._dcfg = tape["dcfg"]
        dhf = self.lin_bwd(tape["hf"], "head.weight", "head.bias", dout, grads)
        dx = self.ln_bwd(tape["xf"], "ln_f", dhf, grads)
        dcond = None
        if c["encoder_only"]:
            Tt = T + 1
            full = torch.zeros((B, Tt, E), dtype=torch.float32, device=dout.device)
            ops.copy2d(dx, full, B, T * E, T * E, Tt * E, dst_off=E)
            dx = full.view(B * Tt, E)
            for li in reversed(range(c["n_layer"])):
                dx = self.enc_layer_bwd(tape["layers"][li], f"encoder.layers.{li}", dx, grads, B)
            dx = self.drop_bwd(dx, 0, tape, "d_tok", self._dcfg)
            grads["pos_emb"].zero_()
            grads["pos_emb"].view(-1)[:Tt * E].copy_(ops.colsum(dx.view(B, Tt * E)))
            dinp = torch.empty((B * T, E), dtype=torch.float32, device=dout.device)
            ops.copy2d(dx, dinp, B, T * E, Tt * E, T * E, src_off=E)
        else:
            S, To = tape["S"], tape["To"]
            dmem = torch.zeros((B * S, E), dtype=torch.float32, device=dout.device)
            for li in reversed(range(c["n_layer"])):
                dx = self.dec_layer_bwd(tape["layers"][li], f"decoder.layers.{li}", dx, grads, B, dmem)
            dx = self.drop_bwd(dx, 0, tape, "d_tok", self._dcfg)
            grads["pos_emb"].zero_()
            grads["pos_emb"].view(-1)[:T * E].copy_(ops.colsum(dx.view(B, T * E)))
            dinp = dx
            if c["n_cond_layers"] > 0:
                for li in reversed(range(c["n_cond_layers"])):
                    dmem = self.enc_layer_bwd(tape["enc"][li], f"encoder.layers.{li}", dmem, grads, B)
            else:
                st = tape["enc_mlp"]
                de2 = self.lin_bwd(st["e2"], "encoder.2.weight", "encoder.2.bias", dmem, grads)
                de1 = ops.act_bwd(st["e1"], de2, "mish")
                dmem = self.lin_bwd(st["m"], "encoder.0.weight", "encoder.0.bias", de1, grads)
            dmem = self.drop_bwd(dmem, 0, tape, "d_cond", self._dcfg)
            grads["cond_pos_emb"].zero_()
            grads["cond_pos_emb"].view(-1)[:S * E].copy_(ops.colsum(dmem.view(B, S * E)))
            if To:
                dco = torch.empty((B * To, E), dtype=torch.float32, device=dout.device)
                ops.copy2d(dmem, dco, B, To * E, S * E, To * E, src_off=E)
                dc = self.lin_bwd(tape["cond"].view(B * To, -1), "cond_obs_emb.weight", "cond_obs_emb.bias", dco, grads, need_dx=need_dcond)
                dcond = None if dc is None else dc.view(B, To, -1)
        ds = self.lin_bwd(tape["sample"].view(B * T, -1), "input_emb.weight", "input_emb.bias", dinp, grads, need_dx=need_dsample)
        return (None if ds is None else ds.view(B, T, -1)), dcond
