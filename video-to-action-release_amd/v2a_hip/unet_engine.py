"""Host-side executor of the AVDC pseudo-3D video UNet forward on the HIP kernels.

Mirrors UNetModel.forward (reference guided_diffusion/guided_diffusion/unet.py:650-684) with the Libero wrapper
(flowdiffusion/unet.py:216-222) as an explicit launch sequence over channels-last video tensors [B, F, H, W, C]:
  * Conv3d = implicit-GEMM 3x3 on [(B F), H, W, C] + implicit-GEMM (3x1) over frames on [B, F, (H W), C]  (nn.py:53-87)
    -- zero permute copies (the reference pays two full rearranges per Conv3d);
  * GroupNorm32 + SiLU fused, fp32 statistics (nn.py:26-28); decoder skip concat read in place by two-source loaders
    (unet.py:681), nearest x2 upsample folded into the conv loader (unet.py:105-115);
  * timestep-embedding add folded into the temporal conv epilogue, residual add folded into the last conv of a ResBlock;
  * per-frame spatial attention (unet.py:303-309, 341-358) by the fused HIP attention kernel;
  * the text branch (PerceiverResampler + Linear + mean, imagen.py:254-372) does not depend on t: computed once per
    sample() call and cached.
Parameters are ordinary torch tensors under the reference's names (checkpoints load unchanged).
"""
import math
import os
import torch
from . import ops

GN_SMALL_MAX = 16384
_EMB_BATCH = True       # all ResBlock embedding projections of a forward as one launch (v2a_emb_linear_multi)


def build_program(cfg):
    """Same flattening of UNetModel.__init__ as the oracle's (kept separate: the product never imports oracle/)."""
    mc = cfg.model_channels
    ch = int(cfg.channel_mult[0] * mc)
    inp = [[("conv", "input_blocks.0.0", cfg.in_channels, ch)]]
    chans = [ch]
    ds = 1
    idx = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            blk = [("res", f"input_blocks.{idx}.0", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in cfg.attention_resolutions:
                blk.append(("attn", f"input_blocks.{idx}.1", ch))
            inp.append(blk)
            chans.append(ch)
            idx += 1
        if level != len(cfg.channel_mult) - 1:
            inp.append([("down", f"input_blocks.{idx}.0", ch)])
            chans.append(ch)
            idx += 1
            ds *= 2
    mid = [("res", "middle_block.0", ch, ch), ("attn", "middle_block.1", ch), ("res", "middle_block.2", ch, ch)]
    out = []
    oidx = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            blk = [("res", f"output_blocks.{oidx}.0", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            j = 1
            if ds in cfg.attention_resolutions:
                blk.append(("attn", f"output_blocks.{oidx}.{j}", ch))
                j += 1
            if level and i == cfg.num_res_blocks:
                blk.append(("up", f"output_blocks.{oidx}.{j}", ch))
                ds //= 2
            out.append(blk)
            oidx += 1
    return inp, mid, out, ch


class _Packs:
    """Packed conv operands keyed by parameter name, refreshed when (data_ptr, _version) changes.  `half`: bf16 packs for the
    bf16-storage kernels (csrc/igemm_h.hip) instead of fp32 ones."""

    def __init__(self, P):
        self.P = P
        self._c = {}

    def get_ups4(self, name, half=False):
        """The four class filters of an Upsample + 3x3 conv (ops.pack_weight_ups4: sums of the fp32 taps), refreshed with the weight like
        every other pack.  half: the 16-bit dtype of the copy the 16-bit kernel takes (rounded once, from the fp32 sums)."""
        w = self.P[name]
        key = (w.data_ptr(), w._version)
        ck = (name, "ups4", half)
        ent = self._c.get(ck)
        if ent is None or ent[0] != key:
            pk = ops.pack_weight_ups4(self.get(name), w.shape[0], w.shape[1], None if (ent is None or half) else ent[1])
            if half:
                pk = ops.cast_h(pk, half)
            ent = (key, pk)
            self._c[ck] = ent
        return ent[1]

    def get(self, name, half=False, pad_cin=0):
        """half: False (fp32 pack) or the 16-bit dtype of the pack (torch.bfloat16 / torch.float16).
        pad_cin: 16-bit pack with the input-channel axis zero-padded to `pad_cin` (the 6-channel stem on the 32-channel-chunk kernels)."""
        w = self.P[name]
        key = (w.data_ptr(), w._version)
        if half is True:
            half = torch.bfloat16
        ck = (name, half, pad_cin)
        ent = self._c.get(ck)
        if ent is None or ent[0] != key:
            wd = w.detach()
            if pad_cin:
                assert half and wd.shape[1] <= pad_cin
                wp = torch.zeros((wd.shape[0], pad_cin) + tuple(wd.shape[2:]), dtype=wd.dtype, device=wd.device)
                wp[:, :wd.shape[1]] = wd
                wd = wp
            taps = 1
            for s in wd.shape[2:]:
                taps *= s
            if half:
                pk = ops.pack_weight_h(wd.contiguous(), None if ent is None else ent[1], dtype=half)
            else:
                pk = wd if taps == 1 else ops.pack_weight(wd.contiguous(), 0, None if ent is None else ent[1])
            ent = (key, pk)
            self._c[ck] = ent
        return ent[1]


class _LazyGN:
    """GroupNorm + activation of a [B,F,H,W,C] tensor whose apply pass is postponed (ops.PendingGN): the ResBlock convs that can,
    normalise their input inside the conv kernel; every other consumer calls `materialize()`."""

    def __init__(self, pending, shape, frames_separate):
        self.pending, self.shape, self.frames_separate = pending, tuple(shape), frames_separate
        self.dtype = pending.x.dtype

    def materialize(self):
        return self.pending.apply().view(self.shape)


class _LazyGN32:
    """fp32 counterpart of _LazyGN: GroupNorm + SiLU of a [B,F,H,W,C] fp32 tensor of which only the statistics ran (ops.PendingGN32).  The
    3x3 ResBlock convs that run on conv_patch_x3 normalise their input in the kernel; every other consumer calls `materialize()`."""

    def __init__(self, pending, shape):
        self.pending, self.shape = pending, tuple(shape)
        self.dtype = torch.float32

    def materialize(self):
        return self.pending.apply().view(self.shape)


class UNetEngine:
    def __init__(self, cfg, params: dict, prefix="unet."):
        self.cfg = cfg
        self.P = params
        self.pre = prefix
        self.device = next(iter(params.values())).device
        self.packs = _Packs(params)
        self.inp, self.mid, self.out, self.final_ch = build_program(cfg)
        # "f32": fp32 tensors in HBM (parity configuration; the MFMA input type follows v2a_hip.set_precision).
        # "bf16": bf16 activations + bf16 weight packs, fp32 accumulation / normalisation statistics / softmax -- the counterpart of
        # the reference's fp16-autocast GPU path (lb_online_trainer_v7.py:889); needs every inner width to be a multiple of 64.
        # "fp16": the same kernels instantiated for IEEE half (v_mfma_f32_32x32x16_f16): the reference's own 16-bit type.
        self.storage = "f32"
        self.hdt = torch.bfloat16
        self._eo = None
        self._res_names = None

    def set_storage(self, mode):
        if mode not in ("f32", "bf16", "fp16"):
            raise ValueError(mode)
        self.hdt = torch.float16 if mode == "fp16" else torch.bfloat16
        if mode != "f32":
            c = self.cfg
            widths = [c.model_channels * m for m in c.channel_mult]
            if any(w % 64 for w in widths):
                raise ValueError(f"bf16 storage needs channel widths that are multiples of 64, got {widths}")
        self.storage = mode

    def w(self, name, half=False, pad_cin=0):
        return self.packs.get(self.pre + name, self.hdt if half else False, pad_cin)

    def p(self, name):
        return self.P[self.pre + name]

    def has(self, name):
        return (self.pre + name) in self.P

    # ------------------------------------------------------------------ primitives
    def _conv3d_h(self, x, name, cout, stride, ups, x2, rowvec, residual, out_f32):
        """bf16-storage Conv3d: spatial conv then (k > 1) temporal conv, both on the LDS-DMA kernel.  A cout that the temporal
        kernel cannot take (the 3-channel output head) leaves the spatial result in fp32 and finishes on the fp32 kernels."""
        B, Fr, H, W, C = x.shape
        k = self.p(name + ".spatial_conv.weight").shape[-1]
        has_t = self.has(name + ".temporal_conv.weight")
        t_half = has_t and cout % 64 == 0
        sp_f32 = (has_t and not t_half) or (out_f32 and not has_t)
        if (ups and k == 3 and stride == 1 and x2 is None and t_half and not isinstance(x, _LazyGN) and not sp_f32
                and ops.conv2d_hp_ups4_ok(B * Fr, 2 * H, 2 * W, C, cout)):
            # Upsample + 3x3 as four 2x2 class convs over the source map: 4 of the 9 products per output (csrc/igemm_hp.hip)
            y = ops.conv2d_hp_ups4(x.view(B * Fr, H, W, C), self.packs.get_ups4(self.pre + name + ".spatial_conv.weight", half=self.hdt),
                                   self.p(name + ".spatial_conv.bias"), cout)
            OH, OW = 2 * H, 2 * W
            z = ops.conv2d_h(y.view(B, Fr, OH * OW, cout), self.w(name + ".temporal_conv.weight", half=True),
                             self.p(name + ".temporal_conv.bias"), cout, 3, 1, (1, 1), (1, 0), rowvec=rowvec,
                             rows_per_batch=Fr * OH * OW, residual=None if residual is None else residual.view(B, Fr, OH * OW, cout),
                             out_f32=out_f32, want_stats=not out_f32)
            stats = None
            if not out_f32:
                z, stats = z
            out = z.view(B, Fr, OH, OW, cout)
            out._gn_stats = stats
            return out
        pre_gn = None
        if isinstance(x, _LazyGN):
            assert x2 is None
            if (not sp_f32 and not x.frames_separate
                    and ops.gn_fuse_pays(cout)
                    and ops.gn_fusable(B * Fr, H, W, C, cout, k, k, (stride, stride), (k // 2, k // 2), ups)):
                pre_gn = x.pending                       # normalised inside the halo conv: the GroupNorm apply pass never runs
                x4 = pre_gn.x.view(B * Fr, H, W, pre_gn.C1)
                x24 = None if pre_gn.x2 is None else pre_gn.x2.view(B * Fr, H, W, -1)
            else:
                x = x.materialize()
        if pre_gn is None:
            x4 = x.view(B * Fr, H, W, C)
            x24 = None if x2 is None else x2.view(B * Fr, H, W, -1)
        y = ops.conv2d_h(x4, self.w(name + ".spatial_conv.weight", half=True), self.p(name + ".spatial_conv.bias"), cout, k, k,
                         (stride, stride), (k // 2, k // 2), x2=x24, ups=ups, rowvec=None if has_t else rowvec, rows_per_batch=1,
                         residual=None if (has_t or residual is None) else residual.view(B * Fr, residual.shape[2], residual.shape[3], cout),
                         out_f32=sp_f32, want_stats=not has_t and not sp_f32, pre_gn=pre_gn)
        stats = None
        if not has_t and not sp_f32:
            y, stats = y
        OH, OW = y.shape[1], y.shape[2]
        if not has_t:
            if rowvec is not None:
                raise NotImplementedError("rowvec on a conv without temporal part")
            out = y.view(B, Fr, OH, OW, cout)
            out._gn_stats = stats        # per-64-row sum / sum-of-squares slabs: the next GroupNorm skips its statistics pass
            return out
        if t_half:
            z = ops.conv2d_h(y.view(B, Fr, OH * OW, cout), self.w(name + ".temporal_conv.weight", half=True),
                             self.p(name + ".temporal_conv.bias"), cout, 3, 1, (1, 1), (1, 0), rowvec=rowvec,
                             rows_per_batch=Fr * OH * OW, residual=None if residual is None else residual.view(B, Fr, OH * OW, cout),
                             out_f32=out_f32, want_stats=not out_f32)
            if not out_f32:
                z, stats = z
        else:
            assert residual is None and out_f32
            z = ops.conv2d(y.view(B, Fr, OH * OW, cout), self.w(name + ".temporal_conv.weight"), self.p(name + ".temporal_conv.bias"),
                           cout, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=Fr * OH * OW)
        out = z.view(B, Fr, OH, OW, cout)
        out._gn_stats = stats
        return out

    def conv3d(self, x, name, cout, stride=1, ups=False, x2=None, rowvec=None, residual=None, out_f32=False):
        """x [B,F,H,W,C] (+x2) -> [B,F,OH,OW,cout].  rowvec [B,cout] / residual [B,F,OH,OW,cout] land in the LAST kernel."""
        B, Fr, H, W, C = x.shape
        if x.dtype in ops.HALF_DTYPES:                      # (a _LazyGN reports bf16)
            return self._conv3d_h(x, name, cout, stride, ups, x2, rowvec, residual, out_f32)
        k = self.p(name + ".spatial_conv.weight").shape[-1]
        has_t = self.has(name + ".temporal_conv.weight")
        if isinstance(x, _LazyGN32):
            if (k == 3 and stride == 1 and not ups and x2 is None and has_t and self.storage == "f32" and cout <= 128
                    and ops.conv2d_x3p_gn_ok(B * Fr, H, W, C, cout)):
                # GroupNorm + SiLU applied inside the spatial conv (conv_patch_x3<GN>): the normalised tensor is never written.  Only
                # where ONE 128-channel column tile covers the layer: every column tile normalises the halo again, and measured per
                # layer (gpurun_out/r6m) the in-kernel arithmetic costs 6-7 % of the conv -- less than the apply pass it replaces at
                # cout = 128 (-200 us per 128 x 128 layer), break-even at 256, a loss at 384 / 512
                pg = x.pending
                y = ops.conv2d_x3p_gn(pg, pg.x.view(B * Fr, H, W, C), self.w(name + ".spatial_conv.weight"), self.p(name + ".spatial_conv.bias"),
                                      cout, Fr)
                z, stats = ops.conv2d(y.view(B, Fr, H * W, cout), self.w(name + ".temporal_conv.weight"), self.p(name + ".temporal_conv.bias"),
                                      cout, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=Fr * H * W,
                                      residual=None if residual is None else residual.view(B, Fr, H * W, cout), want_stats=True)
                out = z.view(B, Fr, H, W, cout)
                out._gn_stats = stats
                return out
            x = x.materialize()
        x4 = x.view(B * Fr, H, W, C)
        x24 = None if x2 is None else x2.view(B * Fr, H, W, -1)
        if (self.storage != "f32" and has_t and cout % 128 == 0 and C < 32 and k == 3 and stride == 1 and x2 is None and not ups
                ):
            # stem (Cin = 6) in the bf16-storage configuration: input padded to one 32-channel chunk, so that the spatial conv runs on the
            # halo kernel (135 GFLOP of padded work at ~1 PFLOP/s instead of 25 GFLOP on the scalar-gather fp32 kernel at 33 TFLOP/s) and
            # its output is born bf16 (no cast launch in front of the temporal conv)
            xh = ops.pad_cast_h(x4, 32, self.hdt)
            y = ops.conv2d_h(xh, self.w(name + ".spatial_conv.weight", half=True, pad_cin=32), self.p(name + ".spatial_conv.bias"), cout, 3, 3,
                             (1, 1), (1, 1))
            z, stats = ops.conv2d_h(y.view(B, Fr, H * W, cout), self.w(name + ".temporal_conv.weight", half=True),
                                    self.p(name + ".temporal_conv.bias"), cout, 3, 1, (1, 1), (1, 0), rowvec=rowvec,
                                    rows_per_batch=Fr * H * W, residual=None if residual is None else residual.view(B, Fr, H * W, cout),
                                    want_stats=True)
            out = z.view(B, Fr, H, W, cout)
            out._gn_stats = stats
            return out
        if (ups and k == 3 and stride == 1 and x2 is None and has_t and self.storage == "f32"
                and ops.conv2d_x3p_ups4_ok(B * Fr, 2 * H, 2 * W, C, cout)):
            # Upsample + 3x3 as four 2x2 class convs over the source map: 4 of the 9 products per output (conv_patch_x3<.., 2>)
            y = ops.conv2d_x3p_ups4(x4, self.packs.get_ups4(self.pre + name + ".spatial_conv.weight"), self.p(name + ".spatial_conv.bias"), cout)
            OH, OW = 2 * H, 2 * W
            z, stats = ops.conv2d(y.view(B, Fr, OH * OW, cout), self.w(name + ".temporal_conv.weight"), self.p(name + ".temporal_conv.bias"),
                                  cout, 3, 1, (1, 1), (1, 0), rowvec=rowvec, rows_per_batch=Fr * OH * OW,
                                  residual=None if residual is None else residual.view(B, Fr, OH * OW, cout), want_stats=True)
            out = z.view(B, Fr, OH, OW, cout)
            out._gn_stats = stats
            return out
        wsp = self.w(name + ".spatial_conv.weight")
        # the LAST kernel of the Conv3d also leaves the per-64-row sums GroupNorm needs (fp32 LDS-DMA kernel epilogue)
        y = ops.conv2d(x4, wsp, self.p(name + ".spatial_conv.bias"), cout, k, k, (stride, stride), (k // 2, k // 2), x2=x24, ups=ups,
                       rowvec=None if has_t else rowvec, rows_per_batch=1,
                       residual=None if (has_t or residual is None) else residual.view(B * Fr, residual.shape[2], residual.shape[3], cout),
                       want_stats=not has_t)
        stats = None
        if not has_t:
            y, stats = y
        OH, OW = y.shape[1], y.shape[2]
        if not has_t:
            if rowvec is not None:
                raise NotImplementedError("rowvec on a conv without temporal part")
            out = y.view(B, Fr, OH, OW, cout)
            out._gn_stats = stats
            return out
        if self.storage != "f32" and cout % 64 == 0:      # stem: fp32 spatial conv (Cin = 6), the 128-wide temporal conv on the bf16 kernel
            z, stats = ops.conv2d_h(ops.cast_h(y, self.hdt).view(B, Fr, OH * OW, cout), self.w(name + ".temporal_conv.weight", half=True),
                                    self.p(name + ".temporal_conv.bias"), cout, 3, 1, (1, 1), (1, 0), rowvec=rowvec,
                                    rows_per_batch=Fr * OH * OW, residual=None if residual is None else residual.view(B, Fr, OH * OW, cout),
                                    want_stats=True)
            out = z.view(B, Fr, OH, OW, cout)
            out._gn_stats = stats
            return out
        wt = self.w(name + ".temporal_conv.weight")
        z, stats = ops.conv2d(y.view(B, Fr, OH * OW, cout), wt, self.p(name + ".temporal_conv.bias"), cout, 3, 1, (1, 1), (1, 0),
                              rowvec=rowvec, rows_per_batch=Fr * OH * OW,
                              residual=None if residual is None else residual.view(B, Fr, OH * OW, cout), want_stats=True)
        out = z.view(B, Fr, OH, OW, cout)
        out._gn_stats = stats
        return out

    def gn_silu(self, x, name, act="silu", x2=None, frames_separate=False, lazy=False):
        """GroupNorm32 over (C/32 x F x H x W) per sample (or per frame when frames_separate), fused activation.
        lazy (bf16 storage): only the statistics run; returns a _LazyGN for conv3d, which applies the normalisation inside its
        3x3 kernel when it can and materialises it otherwise."""
        B, Fr, H, W, C1 = x.shape
        C = C1 + (0 if x2 is None else x2.shape[-1])
        N, S = (B * Fr, H * W) if frames_separate else (B, Fr * H * W)
        x3 = x.view(N, S, C1)
        x23 = None if x2 is None else x2.view(N, S, -1)
        if lazy and x.dtype in ops.HALF_DTYPES and ops.GN_FUSE[0]:
            pg = ops.groupnorm_prep_h(x3, self.p(name + ".weight"), self.p(name + ".bias"), 32, act, x2=x23,
                                      stats=getattr(x, "_gn_stats", None), stats2=None if x2 is None else getattr(x2, "_gn_stats", None))
            return _LazyGN(pg, (B, Fr, H, W, C), frames_separate)
        if x.dtype in ops.HALF_DTYPES:
            return ops.groupnorm_fwd_h(x3, self.p(name + ".weight"), self.p(name + ".bias"), 32, act, x2=x23,
                                       stats=getattr(x, "_gn_stats", None),
                                       stats2=None if x2 is None else getattr(x2, "_gn_stats", None)).view(B, Fr, H, W, C)
        if (lazy and x2 is None and not frames_separate and self.storage == "f32" and x.dtype == torch.float32
                and getattr(x, "_gn_stats", None) is not None and S * (C // 32) > GN_SMALL_MAX):
            pg = ops.groupnorm_prep_f32(x3, self.p(name + ".weight"), self.p(name + ".bias"), 32, act, stats=x._gn_stats)
            if pg is not None:
                return _LazyGN32(pg, (B, Fr, H, W, C))
        if x23 is not None and S * (C // 32) <= GN_SMALL_MAX:
            cat = torch.empty((N, S, C), dtype=torch.float32, device=x.device)      # tiny tensors: materialise the concat
            ops.copy2d(x3, cat, N * S, C1, C1, C)
            ops.copy2d(x23, cat, N * S, C - C1, C - C1, C, dst_off=C1)
            x3, x23 = cat, None
        y, _, _ = ops.groupnorm_fwd(x3, self.p(name + ".weight"), self.p(name + ".bias"), 32, act, x2=x23,
                                    stats=getattr(x, "_gn_stats", None), stats2=None if x2 is None else getattr(x2, "_gn_stats", None))
        return y.view(B, Fr, H, W, C)

    def resblock(self, x, name, cin, cout, semb, x2=None):
        B = x.shape[0]
        a = self.gn_silu(x, name + ".in_layers.0", x2=x2, lazy=True)
        eo = self._eo.get(name) if self._eo is not None else None               # [B,cout], from the one-launch batch of forward_cl
        if eo is None:
            eo = ops.linear(semb, self.p(name + ".emb_layers.1.weight"), self.p(name + ".emb_layers.1.bias"))
        h = self.conv3d(a, name + ".in_layers.2", cout, rowvec=eo)
        a2 = self.gn_silu(h, name + ".out_layers.0", lazy=True)
        if self.has(name + ".skip_connection.spatial_conv.weight"):
            xs = self.conv3d(x, name + ".skip_connection", cout, x2=x2)
        else:
            assert x2 is None
            xs = x
        return self.conv3d(a2, name + ".out_layers.3", cout, residual=xs)

    def attention(self, x, name, C):
        B, Fr, H, W, _ = x.shape
        N, L = B * Fr, H * W
        hc = self.cfg.num_head_channels
        heads = C // hc
        n = self.gn_silu(x, name + ".norm", act="none", frames_separate=True)
        half = x.dtype in ops.HALF_DTYPES
        wq = self.w(name + ".qkv.weight", True).view(3 * C, C) if half else self.p(name + ".qkv.weight").view(3 * C, C)
        wo = self.w(name + ".proj_out.weight", True).view(C, C) if half else self.p(name + ".proj_out.weight").view(C, C)
        qkv = ops.linear(n.view(N * L, C), wq, self.p(name + ".qkv.bias"))
        a = ops.attention(qkv, N, L, heads, hc)
        if half:
            out, stats = ops.linear(a, wo, self.p(name + ".proj_out.bias"), residual=x.view(N * L, C), want_stats=True)
            out = out.view(B, Fr, H, W, C)
            out._gn_stats = stats
            return out
        out = ops.linear(a, wo, self.p(name + ".proj_out.bias"), residual=x.view(N * L, C))
        return out.view(B, Fr, H, W, C)

    # ------------------------------------------------------------------ embeddings
    def _ln_g(self, x2d, gname):
        return ops.layernorm(x2d, self.p(gname), None, 1e-5)

    def label_embedding(self, y):
        """task_attnpool(y).mean(1) -> [B, 4*mc]; y [B,L,512] (CLIP token features)."""
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
        if n_mp > 0:
            mp = ops.mean_rows(y)
            mp = self._ln_g(mp, pre + ".to_latents_from_mean_pooled_seq.0.g")
            mp = ops.linear(mp, self.p(pre + ".to_latents_from_mean_pooled_seq.1.weight"), self.p(pre + ".to_latents_from_mean_pooled_seq.1.bias"))
            ops.copy2d(mp, lat, B, n_mp * D, n_mp * D, NL * D)
        H, dh = cfg.pr_heads, cfg.pr_dim_head
        for li in range(cfg.pr_depth):
            a = f"{pre}.layers.{li}.0"
            xn = ops.layernorm(xp.view(B * L, D), self.p(a + ".norm.weight"), self.p(a + ".norm.bias"))
            ln = ops.layernorm(lat.view(B * NL, D), self.p(a + ".norm_latents.weight"), self.p(a + ".norm_latents.bias"))
            q = ops.linear(ln, self.p(a + ".to_q.weight"))
            kvin = torch.empty((B, L + NL, D), dtype=torch.float32, device=y.device)
            ops.copy2d(xn, kvin, B, L * D, L * D, (L + NL) * D)
            ops.copy2d(ln, kvin, B, NL * D, NL * D, (L + NL) * D, dst_off=L * D)
            kv = ops.linear(kvin.view(B * (L + NL), D), self.p(a + ".to_kv.weight"))
            o = ops.perceiver_attention(q, kv, self.p(a + ".q_scale"), self.p(a + ".k_scale"), B, NL, L + NL, H, dh, 8.0)
            o = ops.linear(o.view(B * NL, H * dh), self.p(a + ".to_out.0.weight"))
            o = ops.layernorm(o, self.p(a + ".to_out.1.weight"), self.p(a + ".to_out.1.bias"))
            lat = ops.axpy(o, lat.view(B * NL, D)).view(B, NL, D)
            f = f"{pre}.layers.{li}.1"
            h = ops.linear(self._ln_g(lat.view(B * NL, D), f + ".0.g"), self.p(f + ".1.weight"))
            h = ops.act_fwd(h, "gelu")
            h = ops.linear(self._ln_g(h, f + ".3.g"), self.p(f + ".4.weight"))
            lat = ops.axpy(h, lat.view(B * NL, D)).view(B, NL, D)
        z = ops.linear(lat.view(B * NL, D), self.p("task_attnpool.1.weight"), self.p("task_attnpool.1.bias"))
        return ops.mean_rows(z.view(B, NL, -1))

    def time_embedding(self, t_long):
        e = ops.sincos_embed(t_long, self.cfg.model_channels, 1)
        e = ops.linear(e, self.p("time_embed.0.weight"), self.p("time_embed.0.bias"))
        e = ops.act_fwd(e, "silu")
        return ops.linear(e, self.p("time_embed.2.weight"), self.p("time_embed.2.bias"))

    # ------------------------------------------------------------------ forward
    def _run(self, blk, h, semb, skip=None):
        for op in blk:
            kind, name = op[0], op[1]
            if kind == "conv":
                h = self.conv3d(h, name, op[3])
            elif kind == "res":
                h = self.resblock(h, name, op[2], op[3], semb, x2=skip)
                skip = None
            elif kind == "attn":
                h = self.attention(h, name, op[2])
            elif kind == "down":
                h = self.conv3d(h, name + ".op", op[2], stride=2)
            elif kind == "up":
                h = self.conv3d(h, name + ".conv", op[2], ups=True)
        return h

    def forward_cl(self, xin, t_long, label_emb):
        """xin [B,F,H,W,Cin] channels-last, t [B] int64, label_emb [B,4mc] -> [B,F,H,W,Cout] channels-last."""
        emb = ops.axpy(self.time_embedding(t_long), label_emb)
        semb = ops.act_fwd(emb, "silu")                     # every ResBlock's emb_layers starts with the same SiLU
        # ... so the 27 `emb_layers` Linears (unet.py:204-210,248-257) are ONE launch over the shared input (30 latency-bound GEMM
        # launches per forward before: 0.6 ms at B = 16).  V2A_EMB_BATCH=0: one GEMM per ResBlock.
        self._eo = None
        B, K = semb.shape
        if _EMB_BATCH and B <= 16 and K % 256 == 0 and K <= 1024:
            if self._res_names is None:
                self._res_names = [op[1] for blk in (self.inp + [self.mid] + self.out) for op in blk if op[0] == "res"]
            outs = ops.emb_linear_multi(semb, [self.p(n + ".emb_layers.1.weight") for n in self._res_names],
                                        [self.p(n + ".emb_layers.1.bias") for n in self._res_names])
            self._eo = dict(zip(self._res_names, outs))
        hs = []
        h = xin
        for i, blk in enumerate(self.inp):
            h = self._run(blk, h, semb)
            if i == 0 and self.storage != "f32" and h.dtype not in ops.HALF_DTYPES:
                h = ops.cast_h(h, self.hdt)          # the stem (Cin = 6) runs on the fp32 kernels; everything after it is bf16 in HBM
            hs.append(h)
        h = self._run(self.mid, h, semb)
        for blk in self.out:
            h = self._run(blk, h, semb, skip=hs.pop())
        a = self.gn_silu(h, "out.0")
        return self.conv3d(a, "out.2", self.cfg.out_channels, out_f32=True)

    def forward_libero(self, x, t, task_embed=None, label_emb=None, frame_ch=3):
        """Unet_Libero / UnetMW / UnetThor / UnetBridge (frame_ch 3) and UnetMWFlow (frame_ch 2) forward:
        x [B, f*frame_ch + 3, H, W] -> [B, f*out_channels, H, W] (reference layouts at the boundary)."""
        B, C, H, W = x.shape
        f = (C - 3) // frame_ch
        xin = ops.video_pack(x.float().contiguous(), f, H, W, frame_ch)
        if label_emb is None:
            label_emb = self.label_embedding(task_embed)
        v = self.forward_cl(xin, t.long().contiguous(), label_emb)                      # [B,f,H,W,3]
        return ops.nhwc_to_nchw(v.view(B * f, H, W, self.cfg.out_channels)).view(B, f * self.cfg.out_channels, H, W)
