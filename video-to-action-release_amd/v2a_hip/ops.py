"""Tensor-level wrappers over the C ABI (one function per exported entry point family).

PyTorch is used for device memory and streams only; every arithmetic op below is a HIP kernel of
libv2a_hip.so launched on torch's current stream with raw pointers.
All activations are channels-last fp32: images [N,H,W,C], sequences [N,T,C], video [B,F,H,W,C].
"""
import os
import threading
import weakref
import torch
from ._lib import lib, check

ACT = {"none": 0, "silu": 1, "relu": 2, "mish": 3, "gelu": 4}

_ws = {}
_ws_retired = []        # outgrown scratch buffers stay allocated: captured hipGraphs hold their raw pointers (see workspace)
_tls = threading.local()      # per host thread: the current scratch lane (ws_lane), the 16-bit format last handed to the C side (_set_fmt)


class ws_lane:
    """Context manager: kernels launched inside use scratch buffer `lane` (concurrent streams must not share split-K slabs)."""

    def __init__(self, lane):
        self.lane = lane

    def __enter__(self):
        self.prev = getattr(_tls, "lane", 0)
        _tls.lane = self.lane

    def __exit__(self, *a):
        _tls.lane = self.prev


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of torch's current stream on the current device.  The private accessor costs 0.1 us against 2.8 us for building a
    torch.cuda.Stream object per launch (tools/host_overhead_probe.py) -- a third of the host time of an eager launch."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def workspace(nbytes: int, device=None) -> torch.Tensor:
    """Grow-only scratch buffer per (device, lane); never resized during graph capture (run one eager step first).
    A captured hipGraph (PolicyTrainer's step, GraphedPredictAction) has the buffer's address baked into its split-K / GroupNorm
    nodes, so a buffer that is outgrown later (a bigger batch, the video sampler) is RETIRED, not freed: were it returned to the
    caching allocator, every replay would keep writing partial sums into memory that now belongs to some other tensor."""
    device = torch.device(device if device is not None else torch.cuda.current_device())
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, getattr(_tls, "lane", 0))          # one scratch buffer per (device, lane): a side-stream launch sequence sets lane 1 (see ws_lane)
    cur = _ws.get(key)
    if cur is None or cur.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("v2a workspace would grow during graph capture; run one eager step first")
        if cur is not None:
            _ws_retired.append(cur)
        cur = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=f"cuda:{idx}")
        _ws[key] = cur
    return cur


def reserve_workspace(nbytes: int, device=None):
    workspace(nbytes, device)


def _chk(t, name="tensor"):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), f"{name}: need contiguous fp32 CUDA tensor"
    return t


# ------------------------------------------------------------------------------------------------ conv family
def pack_weight(w: torch.Tensor, mode: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    """torch layout [Cout,Cin,KH,KW] (or [Cout,Cin,K] / [Cout,Cin]) -> packed operand.
    mode 0: [Cout][KH][KW][Cin] (forward);  mode 1: [Cin][KH'][KW'][Cout] flipped (data-gradient / transposed conv)."""
    _chk(w, "weight")
    if w.dim() == 2:
        co, ci, kh, kw = w.shape[0], w.shape[1], 1, 1
    elif w.dim() == 3:
        co, ci, kh, kw = w.shape[0], w.shape[1], 1, w.shape[2]
    else:
        co, ci, kh, kw = w.shape
    if out is None:
        out = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    check(lib.v2a_pack_weight(w.data_ptr(), out.data_ptr(), co, ci, kh, kw, mode, _stream()), "pack_weight")
    return out


_H_ROUTE_MIN_ROWS = [256]
# fp32 convs whose im2col matrix (rows x K) has at least this many elements run on the LDS-DMA kernel (tools/conv_dma_f32_bench.py)
_DMA_F32_MIN_WORK = [300000]
last_kernel = [None]    # rocprof-style name of the contraction kernel the most recent conv2d / conv2d_wgrad / conv2d_h call launched


def _plan_name(fn, prefix, M, Cout, K):
    import ctypes
    bm, bn, sp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    fn(M, Cout, K, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp))
    return f"{prefix}<{bm.value},{bn.value}>"


def _plan_name_h(M, Cout, K, ept, tname):
    tiles128 = -(-M // 128) * -(-Cout // (64 if Cout <= 64 else 128))
    if tname == "float":                          # fp32 instances: three-plane kernel (default) or the pipelined exact-f32 kernel
        x3 = lib.v2a_get_f32_conv_mode() == 1
        base = "conv_igemm_f32x3" if x3 else "conv_igemm_f32p"
        if tiles128 < 128:                        # (mirrors conv_plan_h)
            return f"{base}<64,64>"
        if Cout <= 64:
            big = x3 and M % 256 == 0 and -(-M // 256) >= 200
            return f"{base}<{256 if big else 128},64>"
        return f"{base}<128,128>"
    if tiles128 < 128:                            # mirrors conv_plan_h (csrc/igemm_h.hip)
        return f"conv_igemm_h<64,64,{tname}>"
    return f"conv_igemm_h<128,{64 if Cout <= 64 else 128},{tname}>"


def graph_capture_mode():
    """Keyword arguments for `torch.cuda.graph(...)`: thread-local capture while a process group is live -- the communicator's watchdog
    thread polls the events of earlier collectives during a capture, and in the default GLOBAL mode its hipEventQuery is an illegal call
    that takes the process down (v2a_hip/trainer.py captures its step graphs the same way)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return {"capture_error_mode": "thread_local"}
    except Exception:
        pass
    return {}


_h_twin_regs = 0
_h_twin = {}        # fp32 operand data_ptr -> bf16 twin of the same operand (registered by the engines that keep both fresh)


def register_h_twin(w_f32: torch.Tensor, w_h: torch.Tensor):
    """Declare `w_h` (bf16) to hold the same packed operand as `w_f32`: in the bf16 precision mode conv2d then runs eligible
    layers on the LDS-DMA bf16 kernel (csrc/igemm_h.hip) -- the fp32 activations are rounded to bf16 by a cast launch first."""
    global _h_twin_regs
    _h_twin_regs += 1
    if _h_twin_regs % 64 == 0:                                  # drop twins whose fp32 operand died (rebuilt engines), or they leak
        for k in [k for k, (ref, _) in _h_twin.items() if ref() is None]:
            del _h_twin[k]
    _h_twin[w_f32.data_ptr()] = (weakref.ref(w_f32), w_h)      # validated by identity: a recycled address never matches


def _twin_of(w_packed):
    ent = _h_twin.get(w_packed.data_ptr())
    if ent is None:
        return None
    if ent[0]() is not w_packed:
        if ent[0]() is None:
            del _h_twin[w_packed.data_ptr()]
        return None
    return ent[1]


_p3_reg = {}        # fp32 packed operand data_ptr -> (weakref, its three bf16 planes [3, numel]) (registered by the engine that keeps both fresh)


def register_p3(w_f32: torch.Tensor, planes: torch.Tensor):
    """Declare `planes` ([3, numel] bf16: hi / mid / lo) to hold the split of the packed fp32 operand `w_f32`: conv2d then runs eligible
    layers whose input also arrives pre-split (x_p3) on the pure LDS-DMA kernel conv_p3 -- bit-identical results, no re-splitting."""
    assert planes.dtype == torch.bfloat16 and planes.numel() == 3 * w_f32.numel()
    if len(_p3_reg) % 64 == 63:
        for k in [k for k, (ref, _) in _p3_reg.items() if ref() is None]:
            del _p3_reg[k]
    _p3_reg[w_f32.data_ptr()] = (weakref.ref(w_f32), planes)


def _p3_of(w_packed):
    ent = _p3_reg.get(w_packed.data_ptr())
    if ent is None or ent[0]() is not w_packed:
        return None
    return ent[1]


class Slabs:
    """Split-K partial sums a conv left for its consumer (conv2d(..., defer=True)): the tensor is sum_s ws[s] (+ bias) (+ residual).
    Only valid until the next launch that uses the same scratch lane -- the consumer must be the very next user."""
    __slots__ = ("ws", "n", "stride", "bias", "residual")

    def __init__(self, ws, n, stride, bias, residual):
        self.ws, self.n, self.stride, self.bias, self.residual = ws, n, stride, bias, residual


def gn_takes_slabs(S, C, G):
    return bool(lib.v2a_groupnorm_takes_slabs(S, C, G))


def gn_takes_post(S, C, G):
    return bool(lib.v2a_groupnorm_takes_post(S, C, G))


def _conv2d_dma_f32(x, w_packed, bias, Cout, KH, KW, stride, pad, x2, rowvec, rows_per_batch, residual, idil, ups, out_hw, y, defer=False,
                    want_stats=False):
    """fp32 conv on the LDS-DMA kernel (exact-f32 MFMA): same results as the register-staged kernel up to summation order.
    defer: returns (y, Slabs | None) -- with Slabs the split-K reduce is left to the consuming GroupNorm launch.
    want_stats: returns (y, stats | None) -- per-64-row (sum, sum of squares) blocks of y for groupnorm_fwd(stats=...)."""
    N, H, W, C1 = x.shape
    C2 = x2.shape[-1] if x2 is not None else 0
    sh, sw = stride
    ph, pw = pad
    if out_hw is None:
        HL = 2 * H if ups else ((H - 1) * idil + 1 if idil > 1 else H)
        WL = 2 * W if ups else ((W - 1) * idil + 1 if idil > 1 else W)
        OH = (HL + 2 * ph - KH) // sh + 1
        OW = (WL + 2 * pw - KW) // sw + 1
    else:
        OH, OW = out_hw
    M, K = N * OH * OW, KH * KW * (C1 + C2)
    if y is None:
        y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device)
    wsb = lib.v2a_conv2d_dma_f32_workspace_bytes(M, Cout, K)
    ws = workspace(wsb, x.device) if wsb else None
    last_kernel[0] = _plan_name_h(M, Cout, K, 32, "float")
    # mirrors conv_dma_launch (csrc/igemm_h.hip): 3x3 / stride 1 / pad 1 (optionally behind the x2 upsample) -> the three-plane halo
    # kernel: whole rows of square 4 ... 64-wide maps, 8 x 16 pixel patches of every other map with OH % 8 == 0, OW % 16 == 0
    sq = OH == OW and OW in (4, 8, 16, 32, 64)
    if (lib.v2a_get_f32_conv_mode() == 1 and KH == 3 and KW == 3 and (sh, sw, ph, pw) == (1, 1, 1, 1) and idil == 1
            and x2 is None and (OH, OW) == ((2 * H, 2 * W) if ups else (H, W)) and (sq or (OH % 8 == 0 and OW % 16 == 0))
            and M % 128 == 0 and Cout % 64 == 0 and not want_stats and N * H * W * C1 < 2 ** 31):
        last_kernel[0] = f"conv_halo_x3<{OW}>" if sq else "conv_halo_x3<8x16>"
        if sq and not ups and rowvec is None and lib.v2a_conv2d_x3m_eligible(N, OW, C1, Cout):
            last_kernel[0] = f"conv_maps_x3<{OW}>"
        if rowvec is None and lib.v2a_conv2d_x3p_eligible(N, OH, OW, C1, Cout):
            last_kernel[0] = "conv_patch_x3<256x128>"
    if (lib.v2a_get_f32_conv_mode() == 1 and (KH, KW, sh, sw, ph, pw) == (3, 1, 1, 1, 1, 0) and not ups and idil == 1 and x2 is None
            and (OH, OW) == (H, W) and lib.v2a_conv2d_x3t_eligible(N, H, W, C1, Cout, rows_per_batch, 0 if rowvec is None else 1)):
        last_kernel[0] = "conv_frames_x3<448x128>"
    if defer and wsb and rowvec is None:
        import ctypes
        ns = ctypes.c_int(0)
        check(lib.v2a_conv2d_fwd_dma_f32_d(x.data_ptr(), _p(x2), w_packed.data_ptr(), _p(bias), None, _p(residual), y.data_ptr(),
                                           _zero_line(x.device).data_ptr(), N, H, W, C1, C2, Cout, KH, KW, sh, sw, ph, pw, 1 if ups else 0,
                                           idil, OH, OW, rows_per_batch, ctypes.byref(ns), _p(ws), wsb, _stream()), "conv2d_fwd_dma_f32_d")
        return y, (Slabs(ws, ns.value, M * Cout, bias, residual) if ns.value > 0 else None)
    stats = None
    if want_stats and _FUSED_STATS and lib.v2a_conv2d_dma_f32_can_emit_stats(M, Cout, K):
        stats = torch.empty(((M + 63) // 64, 2, Cout), dtype=torch.float32, device=x.device)
    check(lib.v2a_conv2d_fwd_dma_f32(x.data_ptr(), _p(x2), w_packed.data_ptr(), _p(bias), _p(rowvec), _p(residual), y.data_ptr(),
                                     _zero_line(x.device).data_ptr(), N, H, W, C1, C2, Cout, KH, KW, sh, sw, ph, pw, 1 if ups else 0, idil,
                                     OH, OW, rows_per_batch, _p(stats), _p(ws), wsb, _stream()), "conv2d_fwd_dma_f32")
    if want_stats:
        return y, stats
    return (y, None) if defer else y


def conv2d(x, w_packed, bias, Cout, KH, KW, stride=(1, 1), pad=(0, 0), x2=None, rowvec=None, rows_per_batch=1,
           residual=None, idil=1, ups=False, out_hw=None, y=None, y2=None, csplit=0, bmode=0, x_h=None, keep_h=None, defer=False,
           want_stats=False, x_p3=None, x2_p3=None):
    """Generic channels-last conv.  x [N,H,W,C1] (+ x2 [N,H,W,C2] concatenated along C).  Returns y [N,OH,OW,Cout]; with defer=True
    (y, Slabs | None): when Slabs is returned, y is NOT written yet -- hand both to the GroupNorm that consumes the conv.
    bf16-MFMA mode: `x_h` = an existing bf16 twin of x (skips the cast launch); `keep_h` (a list) receives the twin that was used, so a
    training engine can hand it to conv2d_wgrad later."""
    _chk(x, "x")
    N, H, W, C1 = x.shape
    C2 = 0
    if x2 is not None:
        _chk(x2, "x2")
        C2 = x2.shape[-1]
    # operands that arrive pre-split (x_p3 / x2_p3: the [3, ...] bf16 planes a GroupNorm launch wrote next to x / x2; planes of the packed
    # weight registered by the engine): the pure LDS-DMA three-plane kernel -- same plan, same slabs, bit-identical results
    if (x_p3 is not None and (x2 is None or x2_p3 is not None) and bmode == 0 and y2 is None and not csplit and rowvec is None and not ups
            and idil == 1 and out_hw is None and not want_stats and Cout % 64 == 0 and lib.v2a_get_precision() == 0
            and lib.v2a_get_f32_conv_mode() == 1):
        w3 = _p3_of(w_packed)
        oh_ = (H + 2 * pad[0] - KH) // stride[0] + 1
        ow_ = (W + 2 * pad[1] - KW) // stride[1] + 1
        if w3 is not None and p3_eligible(N * oh_ * ow_, Cout, KH * KW * (C1 + C2), C1, C2):
            assert x_p3.shape[1:] == x.shape and (x2 is None or x2_p3.shape[1:] == x2.shape)
            return conv2d_p3(x_p3, w3, bias, Cout, KH, KW, stride, pad, x2_3=x2_p3 if x2 is not None else None, residual=residual, y=y,
                             defer=defer)
    # (with the three-plane fp32 products every eligible shape takes this route, however small: a layer must not change its
    # arithmetic with the batch size -- the data-parallel equality test compares B rows on one rank with B/2 on two)
    if (bmode == 0 and y2 is None and not csplit and C1 % 32 == 0 and C2 % 32 == 0 and idil in (1, 2) and not (ups and idil > 1)
            and (N * H * W * KH * KW * (C1 + C2) >= _DMA_F32_MIN_WORK[0] or lib.v2a_get_f32_conv_mode() == 1)
            and lib.v2a_get_precision() == 0):
        return _conv2d_dma_f32(x, w_packed, bias, Cout, KH, KW, stride, pad, x2, rowvec, rows_per_batch, residual, idil, ups, out_hw, y,
                               defer=defer, want_stats=want_stats)
    if want_stats:     # only the LDS-DMA fp32 kernel emits statistics
        assert not defer
        return conv2d(x, w_packed, bias, Cout, KH, KW, stride, pad, x2=x2, rowvec=rowvec, rows_per_batch=rows_per_batch, residual=residual,
                      idil=idil, ups=ups, out_hw=out_hw, y=y, y2=y2, csplit=csplit, bmode=bmode, x_h=x_h, keep_h=keep_h), None
    if (_h_twin and bmode == 0 and y2 is None and not csplit and C1 % 64 == 0 and C2 % 64 == 0 and idil in (1, 2)
            and not (ups and idil > 1) and N * H * W >= _H_ROUTE_MIN_ROWS[0] and lib.v2a_get_precision() == 1):
        wh = _twin_of(w_packed)
        if wh is not None:
            xh = x_h.view(x.shape) if x_h is not None else cast_h(x, POLICY_HALF[0])
            x2h = None if x2 is None else cast_h(x2, POLICY_HALF[0])
            if keep_h is not None:
                keep_h.append(xh)
                if x2h is not None:
                    keep_h.append(x2h)
            return conv2d_h(xh, wh, bias, Cout, KH, KW, stride, pad, x2=x2h, rowvec=rowvec,
                            rows_per_batch=rows_per_batch, residual=residual, ups=ups, out_f32=True, idil=idil, out_hw=out_hw, y=y,
                            defer=defer)
    if defer:          # every other route finishes y itself
        return conv2d(x, w_packed, bias, Cout, KH, KW, stride, pad, x2=x2, rowvec=rowvec, rows_per_batch=rows_per_batch, residual=residual,
                      idil=idil, ups=ups, out_hw=out_hw, y=y, y2=y2, csplit=csplit, bmode=bmode, x_h=x_h, keep_h=keep_h), None
    sh, sw = stride
    ph, pw = pad
    if out_hw is None:
        HL = 2 * H if ups else ((H - 1) * idil + 1 if idil > 1 else H)
        WL = 2 * W if ups else ((W - 1) * idil + 1 if idil > 1 else W)
        OH = (HL + 2 * ph - KH) // sh + 1
        OW = (WL + 2 * pw - KW) // sw + 1
    else:
        OH, OW = out_hw
    M = N * OH * OW
    K = KH * KW * (C1 + C2)
    if y is None:
        if y2 is not None or csplit:
            raise ValueError("split output needs explicit y / y2")
        y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device)
    wsb = lib.v2a_conv2d_workspace_bytes(M, Cout, K)
    ws = workspace(wsb, x.device) if wsb else None
    check(lib.v2a_conv2d_fwd(x.data_ptr(), _p(x2), w_packed.data_ptr(), _p(bias), _p(rowvec), _p(residual), y.data_ptr(),
                             _p(y2), csplit, N, H, W, C1, C2, OH, OW, Cout, KH, KW, sh, sw, ph, pw, idil, 1 if ups else 0,
                             rows_per_batch, bmode, _p(ws), wsb, _stream()), "conv2d_fwd")
    last_kernel[0] = _plan_name(lib.v2a_conv2d_plan, "conv_igemm_bf16" if lib.v2a_get_precision() == 1 else "conv_igemm_f32", M, Cout, K)
    # mirrors v2a_conv2d_fwd (csrc/igemm.hip): reductions of <= 64 values the vector loader cannot take -> the direct kernel
    vec = (C1 + C2) % 16 == 0 and C1 % 4 == 0 and x.data_ptr() % 16 == 0 and w_packed.data_ptr() % 16 == 0
    if (not vec and not bmode and x2 is None and y2 is None and idil == 1 and not ups and K <= 64 and M <= (1 << 20)
            and lib.v2a_debug_set_smallk(-1) == 1):
        last_kernel[0] = "conv_smallk<8>" if K <= 8 else "conv_smallk<64>"
    return y


def split3(x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """fp32 tensor -> its three bf16 planes [3, *x.shape] (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)): the operand format
    of conv2d_p3, for tensors no fused producer (GroupNorm launch, optimiser update kernel, pack launch) writes."""
    _chk(x, "x")
    if out is None:
        out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    check(lib.v2a_split3_f32(x.data_ptr(), out.data_ptr(), x.numel(), x.numel(), _stream()), "split3_f32")
    return out


def p3_eligible(M, Cout, K, C1, C2=0):
    return bool(lib.v2a_conv2d_p3_eligible(M, Cout, K, C1, C2))


def conv2d_p3(x3, w3, bias, Cout, KH, KW, stride=(1, 1), pad=(0, 0), x2_3=None, residual=None, y=None, defer=False):
    """fp32 conv over PRE-SPLIT operands (csrc/igemm_h.hip conv_p3): x3 [3, N, H, W, C1] / x2_3 [3, N, H, W, C2] / w3 [3, Cout * K] bf16
    planes.  Bit-identical to conv2d(x, w, ...) on the unsplit tensors in the three-plane mode (same plan, same slabs).  defer: returns
    (y, Slabs | None) like conv2d."""
    assert x3.dtype == torch.bfloat16 and x3.is_contiguous() and x3.shape[0] == 3 and w3.dtype == torch.bfloat16 and w3.is_contiguous()
    _, N, H, W, C1 = x3.shape
    C2 = 0 if x2_3 is None else x2_3.shape[-1]
    sh, sw = stride
    ph, pw = pad
    OH = (H + 2 * ph - KH) // sh + 1
    OW = (W + 2 * pw - KW) // sw + 1
    M, K = N * OH * OW, KH * KW * (C1 + C2)
    assert w3.numel() == 3 * Cout * K
    if y is None:
        y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x3.device)
    wsb = lib.v2a_conv2d_dma_f32_workspace_bytes(M, Cout, K)
    ws = workspace(wsb, x3.device) if wsb else None
    import ctypes
    ns = ctypes.c_int(0)
    last_kernel[0] = "conv_p3<64,64>"
    check(lib.v2a_conv2d_fwd_p3(x3.data_ptr(), x3.numel() // 3, _p(x2_3), 0 if x2_3 is None else x2_3.numel() // 3, w3.data_ptr(), w3.numel() // 3,
                                _p(bias), _p(residual), y.data_ptr(), _zero_line(x3.device).data_ptr(), N, H, W, C1, C2, Cout, KH, KW, sh, sw,
                                ph, pw, OH, OW, ctypes.byref(ns) if defer else None, _p(ws), wsb, _stream()), "conv2d_fwd_p3")
    if defer:
        return y, (Slabs(ws, ns.value, M * Cout, bias, residual) if ns.value > 0 else None)
    return y


class WgradCollector:
    """Weight gradients whose split-K reduce is postponed: every layer keeps its slabs in a scratch buffer of its own (owned here,
    keyed by the gradient's address) and ONE multi-tensor launch (`flush`) finishes all of them -- 40-55 reduce launches per train
    step become three.  The device tables are cached per list of pending reduces, so the captured step re-uses the tables its eager
    warm-up steps built (flush never copies from the host during capture).

    Lifetime rules (a captured hipGraph has the raw addresses of its tables and slabs baked in): an entry that was used while a
    stream was capturing is PINNED and never evicted; other entries are evicted oldest-first, tables beyond MAX_TABLES and slabs beyond
    MAX_SLABS, and a slab is only evicted while no un-flushed pending item points at it."""
    MAX_TABLES, MAX_SLABS = 32, 1024

    def __init__(self, device):
        self.device = torch.device(device)
        self.slabs = {}             # key -> tensor (insertion / last-use order)
        self.pending = []
        self._pending_keys = set()  # slab keys the un-flushed pending items point at
        self.tables = {}            # key -> (items, work, n)
        self._pinned_slabs, self._pinned_tables = set(), set()
        self.item_bytes = lib.v2a_wgrad_item_bytes()

    def _evict(self, table, pinned, limit, busy=()):
        if len(table) <= limit:
            return
        for k in list(table):
            if len(table) <= limit:
                break
            if k not in pinned and k not in busy:
                del table[k]

    def slab(self, key, nbytes):
        capturing = torch.cuda.is_current_stream_capturing()
        t = self.slabs.get(key)
        if t is None or t.numel() < nbytes:
            if capturing:
                raise RuntimeError("weight-gradient slab buffer missing during graph capture; run one eager step first")
            if key in self._pinned_slabs:
                raise RuntimeError("a weight-gradient slab a captured hipGraph points at would have to grow; rebuild the trainer "
                                   "(its graph) for the new shapes instead of re-using this engine")
            if key in self._pending_keys:
                raise RuntimeError("two un-flushed weight gradients share one slab key")
            # callers without stable keys (gradient buffers re-allocated per call) must not hoard: drop the oldest slabs nothing
            # pending and no captured graph refers to
            self._evict(self.slabs, self._pinned_slabs, self.MAX_SLABS, busy=self._pending_keys)
            t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            self.slabs.pop(key, None)
            self.slabs[key] = t          # (cached tables that point at the old buffer can never match a pending list again)
        elif key in self._pending_keys:
            raise RuntimeError("two un-flushed weight gradients share one slab key")
        if capturing:
            self._pinned_slabs.add(key)
        self._pending_keys.add(key)
        return t

    def add(self, item, blocks, form):
        if blocks > 0:
            self.pending.append((bytes(item), blocks, form))

    def flush(self):
        pend, self.pending = self.pending, []
        self._pending_keys = set()
        if not pend:
            return
        key = tuple(pend)
        ent = self.tables.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if ent is None:
            if capturing:
                raise RuntimeError("weight-gradient reduce table missing during graph capture; run one eager step first")
            import numpy as np
            items = np.frombuffer(b"".join(p[0] for p in pend), dtype=np.uint8).copy()
            work = np.array([[i, b, nb, f] for i, (_, nb, f) in enumerate(pend) for b in range(nb)], dtype=np.int32)
            self._evict(self.tables, self._pinned_tables, self.MAX_TABLES)
            ent = (torch.from_numpy(items).to(self.device), torch.from_numpy(work).to(self.device), int(work.shape[0]))
        else:
            del self.tables[key]          # re-insert: most recently used last
        self.tables[key] = ent
        if capturing:
            self._pinned_tables.add(key)
        check(lib.v2a_wgrad_reduce_multi(ent[0].data_ptr(), ent[1].data_ptr(), ent[2], _stream()), "wgrad_reduce_multi")


class WgradBatch:
    """Weight gradients of SEVERAL layers as one launch (csrc/igemm.hip conv_wgrad_multi_kernel; descriptors in the kernel arguments,
    so a captured hipGraph carries them in its kernel node).  `add` records a gradient (False: the grouped kernel cannot take it --
    launch it alone); `launch` plans the reduction splits over the whole group (about TARGET_WG workgroups per launch, every slice at
    least MIN_DEPTH reduction tiles), runs the main kernel(s) and hands the split-K reduces to the collector (one multi-tensor launch
    at the collector's next flush).  Per-element summation order is fixed by the plan: bitwise reproducible."""
    TARGET_WG, MIN_DEPTH, TARGET_WG_HALO, TARGET_WG_X3, TARGET_WG_X3H = 1280, 4, 512, (1024, 512), 512
    WEIGHT = {0: 1.0, 1: 2.0, 2: 1.5, 3: 1.0, 4: 1.0, 5: 1.0, 6: 1.0, 7: 1.0, 8: 1.0, 9: 1.0, 10: 1.0}   # relative cost of one (output tile, reduction tile) step per kernel body

    def __init__(self, collector):
        self.col = collector
        self.calls = []

    def __len__(self):
        return len(self.calls)

    def _describe(self, c, want, slab, slab_bytes, item, ritem):
        import ctypes
        v, t, rt, sp, rb, rf = (ctypes.c_int(0) for _ in range(6))
        x, dy, x2 = c["x"], c["dy"], c["x2"]
        N, H, W, C1 = x.shape
        C2 = x2.shape[-1] if x2 is not None else 0
        _, OH, OW, Cout = dy.shape
        check(lib.v2a_conv2d_wgrad_describe(x.data_ptr(), _p(x2), dy.data_ptr(), _p(c["x_h"]), _p(c["x2_h"]), _p(c["dy_h"]), c["dw"].data_ptr(),
                                            _p(c["dbias"]), N, H, W, C1, C2, OH, OW, Cout, c["KH"], c["KW"], c["stride"][0], c["stride"][1],
                                            c["pad"][0], c["pad"][1], c["idil"], 1 if c["ups"] else 0, 1 if c["accumulate"] else 0, want,
                                            slab, slab_bytes, item, ctypes.byref(v), ctypes.byref(t), ctypes.byref(rt), ctypes.byref(sp),
                                            ritem, ctypes.byref(rb), ctypes.byref(rf)), "conv2d_wgrad_describe")
        return v.value, t.value, rt.value, sp.value, rb.value, rf.value

    def add(self, x, dy, w_shape, KH, KW, stride=(1, 1), pad=(0, 0), x2=None, idil=1, ups=False, dw=None, accumulate=False, dbias=None,
            x_h=None, dy_h=None, x2_h=None, slab_key=None):
        if dw is None:
            return False
        _chk(x, "x"); _chk(dy, "dy")
        twins = x_h is not None and dy_h is not None and (x2 is None or x2_h is not None) and lib.v2a_get_precision() == 1
        c = dict(x=x, dy=dy, x2=x2, x_h=x_h if twins else None, dy_h=dy_h if twins else None, x2_h=x2_h if (twins and x2 is not None) else None,
                 dw=dw, dbias=dbias, KH=KH, KW=KW, stride=stride, pad=pad, idil=idil, ups=ups, accumulate=accumulate, slab_key=slab_key)
        v, t, rt, _, _, _ = self._describe(c, 0, None, 0, None, None)
        if v < 0:
            return False
        c.update(variant=v, tiles=t, rtiles=rt)
        self.calls.append(c)
        return True

    def launch(self):
        import ctypes
        allc, self.calls = self.calls, []
        if not allc:
            return
        # kernel families (csrc/igemm.hip): the 64x64 / twin-fed bodies (variants 0-2), the halo-tile body (3-5), the three-bf16-plane
        # bodies (6: 64 x 64 tiles, 7: 128 x 128 tiles)
        # (8-10: the three-plane halo body)
        for fam, target in ((0, self.TARGET_WG), (1, self.TARGET_WG_HALO), (2, self.TARGET_WG_X3[0]), (3, self.TARGET_WG_X3[1]),
                            (4, self.TARGET_WG_X3H)):
            calls = [c for c in allc if lib.v2a_wgrad_family(c["variant"]) == fam]
            if calls:
                self._launch_family(calls, target)
        last_kernel[0] = "conv_wgrad_multi"

    def _launch_family(self, calls, target_wg):
        import ctypes
        units = sum(c["tiles"] * c["rtiles"] * self.WEIGHT[c["variant"]] for c in calls)
        per_wg = max(units / target_wg, 1.0)
        ib = self.col.item_bytes
        recs = []
        for c in calls:
            want = int(round(c["rtiles"] * self.WEIGHT[c["variant"]] / per_wg))
            want = max(1, min(want, 128, c["rtiles"] // self.MIN_DEPTH if c["rtiles"] >= self.MIN_DEPTH else 1))
            Cout = c["dy"].shape[-1]
            K = c["KH"] * c["KW"] * (c["x"].shape[-1] + (c["x2"].shape[-1] if c["x2"] is not None else 0))
            nbytes = (want * Cout * K + want * Cout) * 4 if want > 1 else 0
            slab = None
            if nbytes:
                key = c["slab_key"] if c["slab_key"] is not None else c["dw"].data_ptr()
                slab = self.col.slab((key, "multi", nbytes), nbytes)      # size in the key: a slab a captured graph points at never has to grow
            item, ritem = ctypes.create_string_buffer(ib), ctypes.create_string_buffer(ib)
            v, t, rt, sp, rb, rf = self._describe(c, want, _p(slab), nbytes, item, ritem)
            recs.append((-(-rt // sp) * self.WEIGHT[v], item.raw, v, t, ritem.raw, rb, rf))
        recs.sort(key=lambda r: -r[0])                     # deepest slices first (they are dispatched first)
        mx = lib.v2a_wgrad_multi_max()
        for i in range(0, len(recs), mx):
            grp = recs[i:i + mx]
            items = b"".join(r[1] for r in grp)
            vs = (ctypes.c_int * len(grp))(*[r[2] for r in grp])
            ts = (ctypes.c_int * len(grp))(*[r[3] for r in grp])
            check(lib.v2a_conv2d_wgrad_multi(items, vs, ts, len(grp), _stream()), "conv2d_wgrad_multi")
        for r in recs:
            self.col.add(r[4], r[5], r[6])


def conv2d_wgrad(x, dy, w_shape, KH, KW, stride=(1, 1), pad=(0, 0), x2=None, idil=1, ups=False, dw=None, accumulate=False, dbias=None,
                 x_h=None, dy_h=None, x2_h=None, collector=None, slab_key=None):
    """dW in torch layout (shape w_shape = [Cout, Cin, ...]) of the conv whose input was x (+x2) and output grad dy [N,OH,OW,Cout].
    bf16-MFMA mode with bf16 twins of both operands at hand (x_h, dy_h): the twin-fed kernel (half the operand traffic)."""
    _chk(x, "x"); _chk(dy, "dy")
    N, H, W, C1 = x.shape
    C2 = x2.shape[-1] if x2 is not None else 0
    _, OH, OW, Cout = dy.shape
    if dw is None:
        dw = torch.empty(w_shape, dtype=torch.float32, device=x.device)
    M = N * OH * OW
    K = KH * KW * (C1 + C2)
    if (x_h is not None and dy_h is not None and (x2 is None or x2_h is not None) and Cout >= 64 and K > 64 and C1 % 8 == 0
            and C2 % 8 == 0 and Cout % 8 == 0
            and lib.v2a_get_precision() == 1):
        _chk_h(x_h, "x_h"); _chk_h(dy_h, "dy_h")
        wsb = lib.v2a_conv2d_wgrad_h_workspace_bytes(M, Cout, K)
        last_kernel[0] = "conv_wgrad_bf16h<128,128>" if Cout > 64 else "conv_wgrad_bf16h<64,128>"
        if collector is not None and wsb:
            import ctypes
            ws = collector.slab(slab_key if slab_key is not None else dw.data_ptr(), wsb)
            item = ctypes.create_string_buffer(collector.item_bytes)
            nb, fm = ctypes.c_int(0), ctypes.c_int(0)
            check(lib.v2a_conv2d_wgrad_h_deferred(x_h.data_ptr(), _p(x2_h if x2 is not None else None), dy_h.data_ptr(), dw.data_ptr(),
                                                  _p(dbias), N, H, W, C1, C2, OH, OW, Cout, KH, KW, stride[0], stride[1], pad[0], pad[1], idil,
                                                  1 if ups else 0, 1 if accumulate else 0, ws.data_ptr(), wsb, item, ctypes.byref(nb),
                                                  ctypes.byref(fm), _stream()), "conv2d_wgrad_h_deferred")
            collector.add(item.raw, nb.value, fm.value)
            return dw
        ws = workspace(wsb, x.device) if wsb else None
        check(lib.v2a_conv2d_wgrad_h(x_h.data_ptr(), _p(x2_h if x2 is not None else None), dy_h.data_ptr(), dw.data_ptr(), _p(dbias), N, H, W,
                                     C1, C2, OH, OW, Cout, KH, KW, stride[0], stride[1], pad[0], pad[1], idil, 1 if ups else 0,
                                     1 if accumulate else 0, _p(ws), wsb, _stream()),
              "conv2d_wgrad_h")
        last_kernel[0] = "conv_wgrad_bf16h<128,128>" if Cout > 64 else "conv_wgrad_bf16h<64,128>"
        return dw
    wsb = lib.v2a_conv2d_wgrad_workspace_bytes(M, Cout, K)
    if collector is not None and wsb:
        import ctypes
        ws = collector.slab(slab_key if slab_key is not None else dw.data_ptr(), wsb)
        item = ctypes.create_string_buffer(collector.item_bytes)
        nb, fm = ctypes.c_int(0), ctypes.c_int(0)
        check(lib.v2a_conv2d_wgrad_deferred(x.data_ptr(), _p(x2), dy.data_ptr(), dw.data_ptr(), _p(dbias), N, H, W, C1, C2, OH, OW, Cout, KH, KW,
                                            stride[0], stride[1], pad[0], pad[1], idil, 1 if ups else 0, 1 if accumulate else 0,
                                            ws.data_ptr(), wsb, item, ctypes.byref(nb), ctypes.byref(fm), _stream()), "conv2d_wgrad_deferred")
        collector.add(item.raw, nb.value, fm.value)
    else:
        ws = workspace(wsb, x.device) if wsb else None
        check(lib.v2a_conv2d_wgrad(x.data_ptr(), _p(x2), dy.data_ptr(), dw.data_ptr(), _p(dbias), N, H, W, C1, C2, OH, OW, Cout, KH, KW,
                                   stride[0], stride[1], pad[0], pad[1], idil, 1 if ups else 0, 1 if accumulate else 0,
                                   _p(ws), wsb, _stream()), "conv2d_wgrad")
    if lib.v2a_get_precision() == 1:
        kn = "conv_wgrad_bf16"
    else:       # mirrors the dispatch in v2a_conv2d_wgrad: whole 16-B pieces -> the LDS-DMA kernel
        dma = Cout % 4 == 0 and K % 4 == 0 and C1 % 4 == 0 and (C1 + C2) % 4 == 0
        kn = "conv_wgrad_dma_f32" if dma else "conv_wgrad_f32"
    last_kernel[0] = _plan_name(lib.v2a_conv2d_wgrad_plan, kn, M, Cout, K)
    return dw


# ---- bf16-storage family (csrc/igemm_h.hip): activations / packed weights are torch.bfloat16 tensors, accumulation fp32
_zeros_h = {}
_FUSED_STATS = True      # GroupNorm statistics from the producing conv's epilogue


def _zero_line(device):
    idx = torch.device(device).index
    z = _zeros_h.get(idx)
    if z is None:
        z = torch.zeros(256, dtype=torch.bfloat16, device=device)
        _zeros_h[idx] = z
    return z


POLICY_HALF = [torch.bfloat16]                     # dtype of the policy's 16-bit twins (v2a_hip.set_policy_half: bf16 | fp16)
HALF_DTYPES = (torch.bfloat16, torch.float16)      # the two 16-bit storage formats (csrc/common.h: template flag F16 of the `_h` kernels)


def _chk_h(t, name="tensor"):
    """16-bit storage tensor (bf16 or IEEE fp16).  The C side keeps one format flag PER HOST THREAD for its `_h` entry points: it is set
    here from the dtype of the tensor every wrapper checks first, so mixed use (bf16 policy twins, fp16 video storage, several threads)
    stays correct."""
    assert t.is_cuda and t.dtype in HALF_DTYPES and t.is_contiguous(), f"{name}: need contiguous bf16 / fp16 CUDA tensor"
    _set_fmt(t.dtype)
    return t


_fmt_now = _tls                    # (the C flag is thread_local too: csrc/igemm_h.hip g_v2a_half_f16)


def _set_fmt(dtype):
    """Select the 16-bit format of this thread's `_h` launches.  The cache only short-cuts repeated calls from this module;
    set_half_format() below is the one public way to flip the flag, so the cache cannot go stale."""
    if getattr(_fmt_now, "v", None) is not dtype:
        lib.v2a_set_half_format(1 if dtype == torch.float16 else 0)
        _fmt_now.v = dtype


def set_half_format(dtype):
    """Public form of _set_fmt (torch.bfloat16 | torch.float16)."""
    assert dtype in HALF_DTYPES
    _fmt_now.v = None
    _set_fmt(dtype)


def pack_weight_h(w: torch.Tensor, out: torch.Tensor = None, dtype=torch.bfloat16) -> torch.Tensor:
    """torch fp32 weight [Cout,Cin,*taps] -> 16-bit [Cout][taps][Cin] (the K order of conv2d_h); dtype bf16 (default) or fp16."""
    _chk(w, "weight")
    co, ci = w.shape[0], w.shape[1]
    taps = w.numel() // (co * ci)
    if out is None:
        out = torch.empty(w.numel(), dtype=dtype, device=w.device)
    _set_fmt(out.dtype)
    check(lib.v2a_pack_weight_h(w.data_ptr(), out.data_ptr(), co, ci, taps, _stream()), "pack_weight_h")
    return out


def cast_h(x: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """fp32 -> bf16 (default) or fp16, round to nearest even, same shape."""
    _chk(x, "x")
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib.v2a_cast_f32_h(x.data_ptr(), y.data_ptr(), x.numel(), 1 if dtype == torch.float16 else 0, _stream()), "cast_f32_h")
    return y


def pad_cast_h(x: torch.Tensor, cpad: int, dtype=torch.bfloat16) -> torch.Tensor:
    """fp32 [..., C] -> 16-bit [..., cpad] with zero channels behind C (C <= cpad, cpad % 8 == 0)."""
    _chk(x, "x")
    C = x.shape[-1]
    y = torch.empty(x.shape[:-1] + (cpad,), dtype=dtype, device=x.device)
    _set_fmt(dtype)
    check(lib.v2a_pad_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel() // C, C, cpad, _stream()), "pad_cast_f32_bf16")
    return y


def cast_f(x: torch.Tensor) -> torch.Tensor:
    """bf16 -> fp32, same shape."""
    assert x.is_cuda and x.dtype in HALF_DTYPES and x.is_contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib.v2a_cast_h_f32(x.data_ptr(), y.data_ptr(), x.numel(), 1 if x.dtype == torch.float16 else 0, _stream()), "cast_h_f32")
    return y


# test hooks (flipped by tests/test_ops_gpu.py / test_video_gpu.py to run the unfused / tap-by-tap forms of the same layers beside the
# default ones; not configuration): GroupNorm folded into the halo conv's loader, the halo / frame-stack conv kernels themselves
GN_FUSE = [True]
CONV_H3 = [True]
CONV_H2 = [True]       # the multi-stage 256-row kernel (csrc/igemm_h2.hip)
_GN_FUSE_MAX_REPEAT = 2


def gn_fuse_pays(Cout):
    """Is folding the GroupNorm into the halo conv's loader faster than apply pass + plain conv?  Every output-channel tile of the
    kernel normalises the input halo again: with three or more 128-wide tiles (Cout = 384, 640 -> the 512 x 128 instance) it is not
    (32 x 32 x 384: 289 vs 308 us; 768 -> 384: 548 vs 575 us; sampler +0.7 %)."""
    return Cout % 256 == 0 or Cout // 128 <= _GN_FUSE_MAX_REPEAT


def gn_fusable(N, H, W, C, Cout, KH, KW, stride, pad, ups):
    """True when a GroupNorm + activation in front of this conv can run inside the halo kernel (v2a_conv2d_fwd_h3_gn)."""
    return bool(GN_FUSE[0] and CONV_H3[0] and not ups and C <= 1024
                and lib.v2a_conv2d_h3_eligible(N, H, W, C, Cout, KH, KW, stride[0], stride[1], pad[0], pad[1], 0, 0))


class PendingGN:
    """A GroupNorm + activation whose statistics are done (scale / shift table `ab` [N][2][C]) and whose apply pass has not run:
    either the consuming conv applies it to its input tile in LDS, or `apply()` materialises the normalised tensor."""

    def __init__(self, x, x2, ab, act):
        self.x, self.x2, self.ab, self.act = x, x2, ab, act
        self.N, self.S, self.C1 = x.shape
        self.C = self.C1 + (x2.shape[-1] if x2 is not None else 0)

    def apply(self):
        y = torch.empty((self.N, self.S, self.C), dtype=self.x.dtype, device=self.x.device)
        _set_fmt(self.x.dtype)
        check(lib.v2a_groupnorm_apply_h(self.x.data_ptr(), _p(self.x2), self.C1, self.ab.data_ptr(), y.data_ptr(), self.N, self.S, self.C,
                                        ACT[self.act], _stream()), "groupnorm_apply_h")
        return y


def groupnorm_prep_h(x, gamma, beta, G, act="none", eps=1e-5, x2=None, stats=None, stats2=None):
    """Statistics half of groupnorm_fwd_h: returns a PendingGN (same arguments)."""
    _chk_h(x, "x")
    N, S, C1 = x.shape
    C = C1 + (x2.shape[-1] if x2 is not None else 0)
    if x2 is not None:
        _chk_h(x2, "x2")
    wsb = lib.v2a_groupnorm_h_workspace_bytes(N, S, C)
    ws = workspace(wsb, x.device)
    if stats is None or S % 64 or (x2 is not None and stats2 is None):
        stats = stats2 = None
    ab = torch.empty((N, 2, C), dtype=torch.float32, device=x.device)
    check(lib.v2a_groupnorm_prep_h(x.data_ptr(), _p(x2), C1, gamma.data_ptr(), beta.data_ptr(), None, None, _p(stats), _p(stats2),
                                   N, S, C, G, eps, ab.data_ptr(), ws.data_ptr(), wsb, _stream()), "groupnorm_prep_h")
    return PendingGN(x, x2, ab, act)


def conv2d_h(x, w_packed, bias, Cout, KH, KW, stride=(1, 1), pad=(0, 0), x2=None, rowvec=None, rows_per_batch=1, residual=None,
             ups=False, out_f32=False, idil=1, out_hw=None, y=None, want_stats=False, defer=False, pre_gn=None):
    """bf16-storage conv: x [N,H,W,C1] (+x2) bf16, w_packed bf16 [Cout][KH][KW][C1+C2], bias / rowvec fp32, residual bf16.
    Returns bf16 [N,OH,OW,Cout] (fp32 when out_f32).  Needs C1 % 64 == 0 and C2 % 64 == 0.
    pre_gn (a PendingGN over [x | x2]): the conv input is act(GroupNorm(.)) of the given tensors, applied inside the halo kernel --
    the caller checked gn_fusable() first."""
    _chk_h(x, "x")
    N, H, W, C1 = x.shape
    C2 = 0
    if x2 is not None:
        _chk_h(x2, "x2")
        C2 = x2.shape[-1]
    res_h = residual if (residual is not None and residual.dtype in HALF_DTYPES) else None
    res_f = residual if (residual is not None and residual.dtype == torch.float32) else None
    sh, sw = stride
    ph, pw = pad
    if out_hw is None:
        HL, WL = (2 * H, 2 * W) if ups else (((H - 1) * idil + 1, (W - 1) * idil + 1) if idil > 1 else (H, W))
        OH = (HL + 2 * ph - KH) // sh + 1
        OW = (WL + 2 * pw - KW) // sw + 1
    else:
        OH, OW = out_hw
    M, K = N * OH * OW, KH * KW * (C1 + C2)
    if y is None:
        y = torch.empty((N, OH, OW, Cout), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    else:
        out_f32 = y.dtype == torch.float32
    if pre_gn is not None:
        assert gn_fusable(N, H, W, C1 + C2, Cout, KH, KW, stride, pad, ups) and idil == 1 and not out_f32 and res_f is None and not defer
        stats = torch.empty(((M + 63) // 64, 2, Cout), dtype=torch.float32, device=x.device) if (want_stats and _FUSED_STATS) else None
        check(lib.v2a_conv2d_fwd_h3_gn(x.data_ptr(), _p(x2), C1, pre_gn.ab.data_ptr(), N // pre_gn.N, ACT[pre_gn.act], w_packed.data_ptr(),
                                       _p(bias), _p(rowvec), _p(res_h), y.data_ptr(), _zero_line(x.device).data_ptr(), N, H, W, C1 + C2, Cout,
                                       rows_per_batch, _p(stats), _stream()), "conv2d_fwd_h3_gn")
        last_kernel[0] = f"conv_halo_h3_gn<{'256x256' if Cout % 256 == 0 else ('512x128' if OH % 32 == 0 else '256x128')}>"
        return (y, stats) if want_stats else y
    if (idil == 1 and not out_f32 and res_f is None and y.dtype in HALF_DTYPES and not defer and x2 is None
            and CONV_H3[0]
            and lib.v2a_conv2d_h3_eligible(N, H, W, C1, Cout, KH, KW, sh, sw, ph, pw, 1 if ups else 0, C2)):
        # 3x3 / stride 1: the halo-tile kernel (csrc/igemm_h3.hip) -- the nine taps share one DMA of the input patch
        stats = torch.empty(((M + 63) // 64, 2, Cout), dtype=torch.float32, device=x.device) if (want_stats and _FUSED_STATS) else None
        check(lib.v2a_conv2d_fwd_h3(x.data_ptr(), w_packed.data_ptr(), _p(bias), _p(rowvec), _p(res_h), y.data_ptr(),
                                    _zero_line(x.device).data_ptr(), N, H, W, C1, Cout, 1 if ups else 0, rows_per_batch, _p(stats), _stream()),
              "conv2d_fwd_h3")
        last_kernel[0] = f"conv_halo_h3<{'256x256' if Cout % 256 == 0 else ('512x128' if OH % 32 == 0 else '256x128')}>"
        return (y, stats) if want_stats else y
    if (idil == 1 and not out_f32 and res_f is None and y.dtype in HALF_DTYPES and not defer and x2 is None and not ups
            and CONV_H3[0]
            and lib.v2a_conv2d_t3_eligible(N, H, W, C1, Cout, KH, KW, sh, sw, ph, pw, 0, C2)):
        # temporal 3x1 over [B, F, HW, C]: the frame-stack kernel (csrc/igemm_h3.hip) -- the three taps share one DMA of the frames
        stats = torch.empty(((M + 63) // 64, 2, Cout), dtype=torch.float32, device=x.device) if (want_stats and _FUSED_STATS) else None
        check(lib.v2a_conv2d_fwd_t3(x.data_ptr(), w_packed.data_ptr(), _p(bias), _p(rowvec), _p(res_h), y.data_ptr(),
                                    _zero_line(x.device).data_ptr(), N, H, W, C1, Cout, rows_per_batch, _p(stats), _stream()), "conv2d_fwd_t3")
        last_kernel[0] = "conv_frames_h3<448x128>"
        return (y, stats) if want_stats else y
    if (CONV_H2[0] and idil == 1 and not out_f32 and res_f is None and y.dtype in HALF_DTYPES
            and lib.v2a_conv2d_h2_eligible(M, Cout, K, C1, C2)):
        # large layer: the multi-stage 256-row kernel (csrc/igemm_h2.hip)
        stats = torch.empty(((M + 63) // 64, 2, Cout), dtype=torch.float32, device=x.device) if (want_stats and _FUSED_STATS) else None
        check(lib.v2a_conv2d_fwd_h2(x.data_ptr(), _p(x2), w_packed.data_ptr(), _p(bias), _p(rowvec), _p(res_h), y.data_ptr(),
                                    _zero_line(x.device).data_ptr(), N, H, W, C1, C2, Cout, KH, KW, sh, sw, ph, pw, 1 if ups else 0,
                                    OH, OW, rows_per_batch, _p(stats), _stream()), "conv2d_fwd_h2")
        last_kernel[0] = f"conv_igemm_h2<{'256x256' if Cout % 256 == 0 else '256x128'}>"
        return (y, stats) if want_stats else y
    wsb = lib.v2a_conv2d_h_workspace_bytes(M, Cout, K)
    ws = workspace(wsb, x.device) if wsb else None
    if defer:        # fp32 output, split-K plan: leave the reduce (and bias / residual) to the consuming GroupNorm launch
        assert out_f32 and res_h is None and not want_stats
        if wsb and rowvec is None and lib.v2a_conv2d_h_splits(M, Cout, K) > 1:
            import ctypes
            ns = ctypes.c_int(0)
            check(lib.v2a_conv2d_fwd_h_d(x.data_ptr(), _p(x2), w_packed.data_ptr(), _p(bias), None, _p(res_f), y.data_ptr(),
                                         _zero_line(x.device).data_ptr(), N, H, W, C1, C2, Cout, KH, KW, sh, sw, ph, pw, 1 if ups else 0,
                                         idil, OH, OW, rows_per_batch, ctypes.byref(ns), _p(ws), wsb, _stream()), "conv2d_fwd_h_d")
            last_kernel[0] = _plan_name_h(M, Cout, K, 64, "bf16")
            return y, Slabs(ws, ns.value, M * Cout, bias, res_f)
        return conv2d_h(x, w_packed, bias, Cout, KH, KW, stride, pad, x2=x2, rowvec=rowvec, rows_per_batch=rows_per_batch, residual=residual,
                        ups=ups, out_f32=out_f32, idil=idil, out_hw=out_hw, y=y), None
    stats = None
    if want_stats and _FUSED_STATS and not out_f32 and lib.v2a_conv2d_h_can_emit_stats(M, Cout, K):      # GroupNorm statistics ride along
        stats = torch.empty(((M + 63) // 64, 2, Cout), dtype=torch.float32, device=x.device)
    check(lib.v2a_conv2d_fwd_h(x.data_ptr(), _p(x2), w_packed.data_ptr(), _p(bias), _p(rowvec), _p(res_h), _p(res_f),
                               None if out_f32 else y.data_ptr(), y.data_ptr() if out_f32 else None, _zero_line(x.device).data_ptr(),
                               N, H, W, C1, C2, Cout, KH, KW, sh, sw, ph, pw, 1 if ups else 0, idil, OH, OW, rows_per_batch, _p(stats),
                               _p(ws), wsb, _stream()), "conv2d_fwd_h")
    last_kernel[0] = _plan_name_h(M, Cout, K, 64, "bf16")
    if want_stats:
        return y, stats
    return y


def linear(x2d, w, bias=None, residual=None, want_stats=False):
    """y = x @ w.T + b for x [M,K], torch weight [N,K] (already K-contiguous: no pack needed).  bf16 x takes a bf16 weight."""
    M, K = x2d.shape
    if x2d.dtype in HALF_DTYPES:
        r = conv2d_h(x2d.view(1, 1, M, K), w, bias, w.shape[0], 1, 1, residual=None if residual is None else residual.view(1, 1, M, -1),
                     want_stats=want_stats)
        return (r[0].view(M, w.shape[0]), r[1]) if want_stats else r.view(M, w.shape[0])
    y = conv2d(x2d.view(1, 1, M, K), w, bias, w.shape[0], 1, 1, residual=residual)
    return y.view(M, w.shape[0])


def colsum(x2d, out=None, accumulate=False):
    rows, cols = x2d.shape
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    check(lib.v2a_colsum(x2d.data_ptr(), out.data_ptr(), rows, cols, 1 if accumulate else 0, _stream()), "colsum")
    return out


# ------------------------------------------------------------------------------------------------ group norm
def groupnorm_fwd(x, gamma, beta, G, act="none", residual=None, film=None, eps=1e-5, y=None, x2=None, twin_out=None, slabs=None,
                  stats=None, stats2=None, post=None, post_slabs=None, planes_out=None):
    """x [N,S,C] (any leading/spatial shape flattened by the caller); x2 [N,S,C2]: virtual channel concat [x | x2].
    Returns (y [N,S,C(+C2)], mean, rstd).  twin_out (a list): also emit the bf16 twin of y and append it (bf16-MFMA mode: the conv
    that consumes y takes it as x_h and skips its cast launch)."""
    _chk(x, "x")
    N, S, C1 = x.shape
    C = C1 + (x2.shape[-1] if x2 is not None else 0)
    if y is None:
        y = torch.empty((N, S, C), dtype=torch.float32, device=x.device)
    mean = torch.empty(N * G, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N * G, dtype=torch.float32, device=x.device)
    wsb = lib.v2a_groupnorm_workspace_bytes(N, S, C, G)
    ws = workspace(wsb, x.device) if wsb else None
    # film: [N, 2*C] rows (scale | shift); may be a column slice of a wider [N, NF] matrix (batched FiLM projections): row stride
    film_ld = 0 if film is None else (film.stride(0) if film.dim() >= 2 else 2 * C)
    yh = None
    yh3 = 0
    if twin_out is not None and C % 4 == 0:
        yh = torch.empty((N, S, C), dtype=POLICY_HALF[0], device=x.device)
        twin_out.append(yh)
    elif planes_out is not None and x2 is None and gn_takes_post(S, C, G):      # three bf16 planes of y (conv_p3 operand), float4 wave path only
        yh = torch.empty((3, N, S, C), dtype=torch.bfloat16, device=x.device)
        yh3 = N * S * C
        planes_out.append(yh)
    po = (None, None, 0, 0, None)      # explicit operands of the launch: (dense post, post slabs, their number, their stride, their bias)
    if post is not None or post_slabs is not None:
        # added to the OUTPUT (after activation / FiLM): dense tensor, or the split-K slabs (+ bias) of the conv that produces it
        # (conv2d(defer=True) on a scratch lane of its own); float4 wave kernels only -- ask gn_takes_post first
        assert gn_takes_post(S, C, G) and x2 is None and not (post is not None and post_slabs is not None)
        if post_slabs is not None:
            assert post_slabs.residual is None and post_slabs.stride == N * S * C
            po = (None, post_slabs.ws.data_ptr(), post_slabs.n, post_slabs.stride, _p(post_slabs.bias))
        else:
            _chk(post, "post")
            assert post.numel() == N * S * C
            po = (post.data_ptr(), None, 0, 0, None)
    if slabs is not None or po[0] is not None or po[1] is not None or yh3:
        # slabs: x is the (still unwritten) conv output -- the kernel sums the conv's split-K slabs and stores x too
        assert slabs is None or slabs.residual is None
        sl = (slabs.ws.data_ptr(), slabs.n, slabs.stride, _p(slabs.bias)) if slabs is not None else (None, 0, 0, None)
        check(lib.v2a_groupnorm_fwd_s(x.data_ptr(), _p(x2), C1, gamma.data_ptr(), beta.data_ptr(), _p(residual), _p(film), film_ld, y.data_ptr(),
                                      _p(yh), yh3, mean.data_ptr(), rstd.data_ptr(), N, S, C, G, eps, ACT[act], sl[0], sl[1], sl[2], sl[3],
                                      po[0], po[1], po[2], po[3], po[4], _p(ws), wsb, _stream()), "groupnorm_fwd_s")
        return y, mean, rstd
    if (stats is not None and (x2 is None or stats2 is not None) and S % 64 == 0 and residual is None and film is None and yh is None):
        # statistics from the producing convs' epilogues (conv2d(want_stats=True)): the tensor is read once, by the apply pass
        check(lib.v2a_groupnorm_fwd_st(x.data_ptr(), _p(x2), C1, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), N, S, C, G, eps, ACT[act], stats.data_ptr(), _p(stats2), _p(ws), wsb, _stream()),
              "groupnorm_fwd_st")
        return y, mean, rstd
    check(lib.v2a_groupnorm_fwd_t(x.data_ptr(), _p(x2), C1, gamma.data_ptr(), beta.data_ptr(), _p(residual), _p(film), film_ld, y.data_ptr(),
                                  _p(yh), mean.data_ptr(), rstd.data_ptr(), N, S, C, G, eps, ACT[act], _p(ws), wsb, _stream()),
          "groupnorm_fwd")
    return y, mean, rstd


class PendingGN32:
    """fp32 GroupNorm (+ activation) whose apply pass is postponed: only mean / rstd exist (from the producing conv's statistics blocks).
    conv2d_x3p_gn() normalises inside the conv kernel; apply() materialises the tensor for every other consumer."""

    def __init__(self, x, gamma, beta, G, act, eps, stats, mean, rstd):
        self.x, self.gamma, self.beta, self.G, self.act, self.eps, self.stats, self.mean, self.rstd = x, gamma, beta, G, act, eps, stats, mean, rstd

    def apply(self):
        y, _, _ = groupnorm_fwd(self.x, self.gamma, self.beta, self.G, self.act, eps=self.eps, stats=self.stats)
        return y


def groupnorm_prep_f32(x, gamma, beta, G, act="none", eps=1e-5, stats=None):
    """Statistics only (v2a_groupnorm_stats_f32: mean / rstd from the conv epilogue's per-64-row blocks) -> PendingGN32, or None when
    the shape is outside the fused form (no statistics blocks, S % 64, channel groups that are not whole float4s, exact-f32 mode)."""
    N, S, C = x.shape
    if stats is None or S % 64 or C % G or (C // G) % 4 or lib.v2a_get_f32_conv_mode() != 1 or act not in ("none", "silu"):
        return None
    mean = torch.empty(N * G, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N * G, dtype=torch.float32, device=x.device)
    wsb = N * 64 * 2 * C * 8
    ws = workspace(wsb, x.device)
    check(lib.v2a_groupnorm_stats_f32(stats.data_ptr(), mean.data_ptr(), rstd.data_ptr(), N, S, C, G, eps, ws.data_ptr(), wsb, _stream()),
          "groupnorm_stats_f32")
    return PendingGN32(x, gamma, beta, G, act, eps, stats, mean, rstd)


def conv2d_x3p_gn_ok(N, H, W, C, Cout):
    return bool(lib.v2a_conv2d_x3p_eligible(N, H, W, C, Cout)) and lib.v2a_get_f32_conv_mode() == 1 and lib.v2a_get_precision() == 0


def conv2d_x3p_gn(pg, x4, w_packed, bias, Cout, frames_per_sample):
    """3x3 / stride 1 / pad 1 conv over act(GroupNorm(x)) with the normalisation applied inside the kernel (csrc/igemm_x3p.hip
    conv_patch_x3<GN>).  x4 = pg.x viewed as [N images, H, W, C]; frames_per_sample images form one GroupNorm sample."""
    N, H, W, C = x4.shape
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x4.device)
    last_kernel[0] = "conv_patch_x3_gn<256x128>"
    check(lib.v2a_conv2d_fwd_x3p_gn(x4.data_ptr(), pg.mean.data_ptr(), pg.rstd.data_ptr(), pg.gamma.data_ptr(), pg.beta.data_ptr(), pg.G,
                                    frames_per_sample, ACT[pg.act], w_packed.data_ptr(), _p(bias), y.data_ptr(), _zero_line(x4.device).data_ptr(),
                                    N, H, W, C, Cout, _stream()), "conv2d_fwd_x3p_gn")
    return y


def pack_weight_ups4(w_packed, Cout, C, out=None):
    """Forward pack [Cout][3][3][C] of an Upsample + 3x3 conv -> the four class filters [4][Cout][2][2][C] (v2a_pack_weight_ups4)."""
    if out is None:
        out = torch.empty((4, Cout, 2, 2, C), dtype=torch.float32, device=w_packed.device)
    check(lib.v2a_pack_weight_ups4(w_packed.data_ptr(), out.data_ptr(), Cout, C, _stream()), "pack_weight_ups4")
    return out


def conv2d_x3p_ups4_ok(N, OH, OW, C, Cout):
    return (bool(lib.v2a_conv2d_x3p_ups4_eligible(N, OH, OW, C, Cout)) and lib.v2a_get_f32_conv_mode() == 1
            and lib.v2a_get_precision() == 0)


def conv2d_x3p_ups4(x, w_ups4, bias, Cout):
    """Upsample (nearest x2) + 3x3 / pad 1 conv of x [N, H, W, C] -> [N, 2H, 2W, Cout] as four 2x2 class convs over x (4 / 9 of the MACs)."""
    N, H, W, C = x.shape
    y = torch.empty((N, 2 * H, 2 * W, Cout), dtype=torch.float32, device=x.device)
    last_kernel[0] = "conv_patch_x3_ups4<256x128>"
    check(lib.v2a_conv2d_fwd_x3p_ups4(x.data_ptr(), w_ups4.data_ptr(), _p(bias), y.data_ptr(), _zero_line(x.device).data_ptr(), N, 2 * H, 2 * W,
                                      C, Cout, _stream()), "conv2d_fwd_x3p_ups4")
    return y


def conv2d_hp_ups4_ok(N, OH, OW, C, Cout):
    return bool(lib.v2a_conv2d_hp_ups4_eligible(N, OH, OW, C, Cout))


def conv2d_hp_ups4(x, w_ups4, bias, Cout):
    """16-bit Upsample (nearest x2) + 3x3 / pad 1 conv of x [N, H, W, C] -> [N, 2H, 2W, Cout] as four 2x2 class convs over x
    (csrc/igemm_hp.hip); w_ups4 = the pre-summed class filters [4, Cout, 2, 2, C] in x's dtype."""
    _chk_h(x, "x")
    _set_fmt(x.dtype)
    N, H, W, C = x.shape
    assert w_ups4.dtype == x.dtype
    y = torch.empty((N, 2 * H, 2 * W, Cout), dtype=x.dtype, device=x.device)
    last_kernel[0] = "conv_patch_h_ups4<512x128>"
    check(lib.v2a_conv2d_fwd_hp_ups4(x.data_ptr(), w_ups4.data_ptr(), _p(bias), y.data_ptr(), _zero_line(x.device).data_ptr(), N, 2 * H, 2 * W,
                                     C, Cout, _stream()), "conv2d_fwd_hp_ups4")
    return y


def groupnorm_fwd_h(x, gamma, beta, G, act="none", eps=1e-5, x2=None, stats=None, stats2=None):
    """bf16-storage GroupNorm + activation: x [N,S,C1] (+ x2 [N,S,C2] virtual concat) bf16 -> y [N,S,C] bf16.
    stats / stats2: the per-64-row statistic slabs conv2d_h(want_stats=True) returned with x / x2 (skips the statistics pass)."""
    _chk_h(x, "x")
    N, S, C1 = x.shape
    C = C1 + (x2.shape[-1] if x2 is not None else 0)
    if x2 is not None:
        _chk_h(x2, "x2")
    y = torch.empty((N, S, C), dtype=x.dtype, device=x.device)
    wsb = lib.v2a_groupnorm_h_workspace_bytes(N, S, C)
    ws = workspace(wsb, x.device)
    if stats is None or S % 64 or (x2 is not None and stats2 is None):
        stats = stats2 = None
    check(lib.v2a_groupnorm_fwd_h(x.data_ptr(), _p(x2), C1, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), None, None, _p(stats),
                                  _p(stats2), N, S, C, G, eps, ACT[act], ws.data_ptr(), wsb, _stream()), "groupnorm_fwd_h")
    return y


def groupnorm_bwd(x, gamma, beta, G, dout, mean, rstd, act="none", residual=None, film=None, want_dres=False, want_dfilm=False,
                  dgamma=None, dbeta=None, accumulate_params=False, dfilm_out=None, twin_out=None, colsum=None, defer_params=False,
                  dout_slabs=None, dout_sum=None, planes_out=None):
    """Returns dx, dgamma, dbeta, dres (or None), dfilm [N,2,C] (or None).  dfilm_out: [N, 2*C] destination with the SAME row stride
    as `film` (a column slice of the batched [N, NF] gradient matrix).  twin_out (a list): also emit the bf16 twin of dx.
    defer_params: only fill `colsum` [N,2,C] (per-sample sums); the caller reduces it over n for many layers at once
    (gn_param_grads_multi) -- dgamma / dbeta are returned as None."""
    N, S, C = x.shape
    dx = torch.empty_like(x)
    dxh = None
    yh3 = 0
    if twin_out is not None and C % 4 == 0:
        dxh = torch.empty(x.shape, dtype=POLICY_HALF[0], device=x.device)
        twin_out.append(dxh)
    elif planes_out is not None and gn_takes_post(S, C, G):      # three bf16 planes of dx (conv_p3 operand of the data gradient that follows)
        dxh = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
        yh3 = x.numel()
        planes_out.append(dxh)
    dres = torch.empty_like(x) if want_dres else None
    dfilm = dfilm_out if dfilm_out is not None else (torch.empty((N, 2, C), dtype=torch.float32, device=x.device) if want_dfilm else None)
    film_ld = 0 if film is None else (film.stride(0) if film.dim() >= 2 else 2 * C)
    if dfilm_out is not None:
        assert film is not None and dfilm_out.stride(0) == film_ld
    colsum_ = torch.empty((N, 2, C), dtype=torch.float32, device=x.device) if colsum is None else colsum
    assert colsum_.numel() == N * 2 * C and colsum_.is_contiguous()
    if defer_params:
        dgamma = dbeta = None
    else:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if dgamma is None else dgamma
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if dbeta is None else dbeta
    wsb = lib.v2a_groupnorm_workspace_bytes(N, S, C, G)
    ws = workspace(wsb, x.device) if wsb else None
    if dout_slabs is not None:      # dout = sum of the producing data-gradient conv's split-K slabs (+ its epilogue residual)
        sl = dout_slabs
        assert sl.bias is None
        check(lib.v2a_groupnorm_bwd_s(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(residual), _p(film), film_ld, None,
                                      mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dxh), yh3, _p(dres), _p(dfilm), colsum_.data_ptr(),
                                      _p(dgamma), _p(dbeta), 1 if accumulate_params else 0, N, S, C, G, ACT[act], sl.ws.data_ptr(), sl.n,
                                      sl.stride, _p(sl.residual), _p(dout_sum), None, 0, _stream()), "groupnorm_bwd_s")
        return dx, dgamma, dbeta, dres, dfilm
    if yh3:
        check(lib.v2a_groupnorm_bwd_s(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(residual), _p(film), film_ld, dout.data_ptr(),
                                      mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dxh), yh3, _p(dres), _p(dfilm), colsum_.data_ptr(),
                                      _p(dgamma), _p(dbeta), 1 if accumulate_params else 0, N, S, C, G, ACT[act], None, 0, 0, None, None,
                                      _p(ws), wsb, _stream()), "groupnorm_bwd_s")
        return dx, dgamma, dbeta, dres, dfilm
    check(lib.v2a_groupnorm_bwd_t(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(residual), _p(film), film_ld, dout.data_ptr(),
                                  mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dxh), _p(dres), _p(dfilm), colsum_.data_ptr(),
                                  _p(dgamma), _p(dbeta), 1 if accumulate_params else 0, N, S, C, G, ACT[act], _p(ws), wsb,
                                  _stream()),
          "groupnorm_bwd")
    return dx, dgamma, dbeta, dres, dfilm


def gn_param_grads_multi(table, work, nwork):
    """dgamma / dbeta of many GroupNorm layers from their colsum buffers in one launch (see v2a_gn_param_grads_multi)."""
    check(lib.v2a_gn_param_grads_multi(table.data_ptr(), work.data_ptr(), nwork, _stream()), "gn_param_grads_multi")


# ------------------------------------------------------------------------------------------------ elementwise
def act_fwd(x, act, y=None):
    y = torch.empty_like(x) if y is None else y
    check(lib.v2a_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), ACT[act], _stream()), "act_fwd")
    return y


def act_bwd(x, dy, act):
    dx = torch.empty_like(x)
    check(lib.v2a_act_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ACT[act], _stream()), "act_bwd")
    return dx


def axpy(a, b, alpha=1.0, out=None):
    out = torch.empty_like(a) if out is None else out
    check(lib.v2a_axpy(a.data_ptr(), b.data_ptr(), out.data_ptr(), float(alpha), a.numel(), _stream()), "axpy")
    return out


def copy2d(src, dst, rows, cols, ld_src, ld_dst, src_off=0, dst_off=0, accumulate=False):
    check(lib.v2a_copy2d(src.data_ptr() + 4 * src_off, dst.data_ptr() + 4 * dst_off, rows, cols, ld_src, ld_dst,
                         1 if accumulate else 0, _stream()), "copy2d")
    return dst


def sincos_embed(t_long, dim, kind):
    B = t_long.numel()
    out = torch.empty((B, dim), dtype=torch.float32, device=t_long.device)
    check(lib.v2a_sincos_embed(t_long.data_ptr(), out.data_ptr(), B, dim, kind, _stream()), "sincos_embed")
    return out


def add_noise(act, noise, t_long, alphas_cumprod, limits=None):
    """limits: (min [Da], max [Da]) device tensors of the action normaliser, or None for -1 / +1."""
    out = torch.empty_like(act)
    B = act.shape[0]
    lo, hi = limits if limits is not None else (None, None)
    check(lib.v2a_add_noise(act.data_ptr(), noise.data_ptr(), t_long.data_ptr(), alphas_cumprod.data_ptr(), out.data_ptr(), B,
                            act.numel() // B, _p(lo), _p(hi), act.shape[-1], _stream()), "add_noise")
    return out


def mse_loss(pred, target, want_grad=True, grad_scale_ptr=0):
    """grad_scale_ptr: device address of a float the loss GRADIENT is multiplied by (the fp16 mode's dynamic loss scale), or 0."""
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    check(lib.v2a_mse_loss_scaled(pred.data_ptr(), target.data_ptr(), loss.data_ptr(), _p(dpred), pred.numel(), grad_scale_ptr or None,
                                  _stream()), "mse_loss")
    return loss, dpred


def nchw_to_nhwc(src, normalize=False):
    N, C, H, W = src.shape
    dst = torch.empty((N, H, W, C), dtype=torch.float32, device=src.device)
    assert src.is_contiguous()
    if src.dtype == torch.uint8:
        check(lib.v2a_nchw_to_nhwc_u8(src.data_ptr(), dst.data_ptr(), N, C, H * W, 1 if normalize else 0, _stream()), "nchw_to_nhwc")
    else:
        _chk(src)
        check(lib.v2a_nchw_to_nhwc_f32(src.data_ptr(), dst.data_ptr(), N, C, H * W, 1 if normalize else 0, _stream()), "nchw_to_nhwc")
    return dst


def nchw_to_nhwc4p(src, dst, pad, normalize=False):
    """NCHW RGB batch -> the interior of the zero-bordered [N, H + 2 pad, W + 2 pad, 4] image `dst` (the caller zeroed it once)."""
    N, C, H, W = src.shape
    assert C == 3 and src.is_contiguous() and tuple(dst.shape) == (N, H + 2 * pad, W + 2 * pad, 4) and dst.dtype == torch.float32
    if src.dtype != torch.uint8:
        _chk(src)
    check(lib.v2a_nchw_to_nhwc4p(src.data_ptr(), 1 if src.dtype == torch.uint8 else 0, dst.data_ptr(), N, H, W, pad, 1 if normalize else 0,
                                 _stream()), "nchw_to_nhwc4p")
    return dst


def conv2d_window(xp, w_win, Cout, KH, KW, stride, out_hw, xpitch, C):
    """Channel-window conv (lib.v2a_conv2d_fwd_window_f32): xp [N, Hp, Wp, c] zero-bordered few-channel image viewed as windows of C
    floats every `xpitch` floats; w_win [Cout, KH, KW, C].  fp32 three-plane conv mode only."""
    N, Hp, Wp, c = xp.shape
    assert (Wp * c) % xpitch == 0
    OH, OW = out_hw
    y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=xp.device)
    M, K = N * OH * OW, KH * KW * C
    wsb = lib.v2a_conv2d_dma_f32_workspace_bytes(M, Cout, K)
    ws = workspace(wsb, xp.device) if wsb else None
    last_kernel[0] = _plan_name_h(M, Cout, K, 32, "float")
    check(lib.v2a_conv2d_fwd_window_f32(xp.data_ptr(), w_win.data_ptr(), None, y.data_ptr(), _zero_line(xp.device).data_ptr(), N, Hp,
                                        (Wp * c) // xpitch, xpitch, C, Cout, KH, KW, stride[0], stride[1], OH, OW, _p(ws), wsb, _stream()),
          "conv2d_fwd_window_f32")
    return y


def nhwc_to_nchw(src):
    N, H, W, C = src.shape
    dst = torch.empty((N, C, H, W), dtype=torch.float32, device=src.device)
    check(lib.v2a_nhwc_to_nchw_f32(src.data_ptr(), dst.data_ptr(), N, C, H * W, _stream()), "nhwc_to_nchw")
    return dst


# ------------------------------------------------------------------------------------------------ attention etc.
def attention(qkv, n_frames, L, heads, head_ch):
    if qkv.dtype in HALF_DTYPES:
        _set_fmt(qkv.dtype)
        out = torch.empty((n_frames * L, heads * head_ch), dtype=qkv.dtype, device=qkv.device)
        check(lib.v2a_attention_fwd_h(qkv.data_ptr(), out.data_ptr(), n_frames, L, heads, head_ch, _stream()), "attention_fwd_h")
        return out
    out = torch.empty((n_frames * L, heads * head_ch), dtype=torch.float32, device=qkv.device)
    check(lib.v2a_attention_fwd(qkv.data_ptr(), out.data_ptr(), n_frames, L, heads, head_ch, _stream()), "attention_fwd")
    return out


def attention_bwd(qkv, out, dout, n_frames, L, heads, head_ch):
    """d(qkv) [n_frames*L, 3*C] of attention(qkv) given its output and d(output) (fp32)."""
    _chk(qkv, "qkv"); _chk(out, "out"); _chk(dout, "dout")
    dqkv = torch.empty_like(qkv)
    check(lib.v2a_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), n_frames, L, heads, head_ch, _stream()),
          "attention_bwd")
    return dqkv


def sumpool2x2(du):
    """[N, 2H, 2W, C] -> [N, H, W, C]: gradient of the nearest x2 upsample folded into a conv."""
    _chk(du, "du")
    N, H2, W2, C = du.shape
    dx = torch.empty((N, H2 // 2, W2 // 2, C), dtype=torch.float32, device=du.device)
    check(lib.v2a_sumpool2x2(du.data_ptr(), dx.data_ptr(), N, H2 // 2, W2 // 2, C, _stream()), "sumpool2x2")
    return dx


def colsum_batched(x3d, out=None, accumulate=False):
    """x [B, rows, C] -> [B, C] column sums per sample."""
    _chk(x3d, "x")
    B, rows, C = x3d.shape
    if out is None:
        out = torch.empty((B, C), dtype=torch.float32, device=x3d.device)
    nb = lib.v2a_colsum_batched_workspace_bytes(B, rows, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=x3d.device)
    check(lib.v2a_colsum_batched(x3d.data_ptr(), out.data_ptr(), B, rows, C, 1 if accumulate else 0, ws.data_ptr(), nb, _stream()),
          "colsum_batched")
    return out


def perceiver_attention(q, kv, q_scale, k_scale, B, Lq, Lk, H, D, sim_scale=8.0):
    out = torch.empty((B, Lq, H * D), dtype=torch.float32, device=q.device)
    check(lib.v2a_perceiver_attention(q.data_ptr(), kv.data_ptr(), q_scale.data_ptr(), k_scale.data_ptr(), out.data_ptr(), B, Lq,
                                      Lk, H, D, sim_scale, _stream()), "perceiver_attention")
    return out


def layernorm(x2d, g, b=None, eps=1e-5):
    rows, Dm = x2d.shape
    y = torch.empty_like(x2d)
    check(lib.v2a_layernorm(x2d.data_ptr(), g.data_ptr(), _p(b), y.data_ptr(), rows, Dm, eps, _stream()), "layernorm")
    return y


def perceiver_attention_bwd(q, kv, q_scale, k_scale, out, dout, B, Lq, Lk, H, D, sim_scale=8.0):
    """-> (dq [B,Lq,H*D], dkv [B,Lk,2*H*D], dq_scale [D], dk_scale [D])."""
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    part = torch.empty((B * H, 2, D), dtype=torch.float32, device=q.device)
    check(lib.v2a_perceiver_attention_bwd(q.data_ptr(), kv.data_ptr(), q_scale.data_ptr(), k_scale.data_ptr(), out.data_ptr(),
                                          dout.data_ptr(), dq.data_ptr(), dkv.data_ptr(), part.data_ptr(), B, Lq, Lk, H, D, sim_scale,
                                          _stream()), "perceiver_attention_bwd")
    sums = colsum(part.view(B * H, 2 * D))
    return dq, dkv, sums[:D], sums[D:]


def layernorm_bwd(x2d, g, dy2d, eps=1e-5):
    """-> (dx [rows,D], dg [D], db [D]) of y = layernorm(x) * g (+ b)."""
    rows, Dm = x2d.shape
    dx = torch.empty_like(x2d)
    contrib = torch.empty((rows, 2 * Dm), dtype=torch.float32, device=x2d.device)
    check(lib.v2a_layernorm_bwd(x2d.data_ptr(), g.data_ptr(), dy2d.data_ptr(), dx.data_ptr(), contrib.data_ptr(), rows, Dm, eps, _stream()),
          "layernorm_bwd")
    sums = colsum(contrib)
    return dx, sums[:Dm], sums[Dm:]


def bcast_rows(dout2d, R, scale=1.0):
    """[B,D] -> [B,R,D]: every row of sample b receives scale * dout[b] (gradient of a mean over R rows with scale = 1/R)."""
    B, Dm = dout2d.shape
    dx = torch.empty((B, R, Dm), dtype=torch.float32, device=dout2d.device)
    check(lib.v2a_bcast_rows(dout2d.data_ptr(), dx.data_ptr(), B, R, Dm, float(scale), _stream()), "bcast_rows")
    return dx


def mean_rows(x3d):
    B, R, Dm = x3d.shape
    out = torch.empty((B, Dm), dtype=torch.float32, device=x3d.device)
    check(lib.v2a_mean_rows(x3d.data_ptr(), out.data_ptr(), B, R, Dm, _stream()), "mean_rows")
    return out


def maxpool_fwd(x):
    N, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, OH, OW, C), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, OH, OW, C), dtype=torch.int8, device=x.device)
    check(lib.v2a_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, H, W, C, _stream()), "maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, in_shape):
    N, H, W, C = in_shape
    dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    check(lib.v2a_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), N, H, W, C, _stream()), "maxpool_bwd")
    return dx


def spatial_softmax_fwd(feat):
    B, H, W, K = feat.shape
    kp = torch.empty((B, K * 2), dtype=torch.float32, device=feat.device)
    att = torch.empty_like(feat)
    check(lib.v2a_spatial_softmax_fwd(feat.data_ptr(), kp.data_ptr(), att.data_ptr(), B, H, W, K, _stream()), "spatial_softmax_fwd")
    return kp, att


def spatial_softmax_bwd(att, kp, dkp):
    B, H, W, K = att.shape
    dfeat = torch.empty_like(att)
    check(lib.v2a_spatial_softmax_bwd(att.data_ptr(), kp.data_ptr(), dkp.data_ptr(), dfeat.data_ptr(), B, H, W, K, _stream()),
          "spatial_softmax_bwd")
    return dfeat


_TSTAMP = os.environ.get("V2A_TSTAMP", "0")      # measurement aid (tools/phase_clock.py): "1" phase stamps, "2" also per-block stamps
_TS = {"on": _TSTAMP in ("1", "2"), "fine": _TSTAMP == "2", "buf": None, "names": []}


def tstamp(name):
    """Measurement aid (V2A_TSTAMP=1): record the wall clock at this point of the CURRENT stream into the next slot (capturable: the
    slot order is the call order of the first pass; replays overwrite the same slots).  Read with tstamp_table()."""
    if not _TS["on"]:
        return
    if _TS["buf"] is None:
        _TS["buf"] = torch.zeros(4096, dtype=torch.int64, device="cuda")
    names = _TS["names"]
    i = _TS.get("cursor", 0)
    if i < len(names):
        assert names[i] == name, (names[i], name)
    else:
        names.append(name)
    _TS["cursor"] = i + 1
    check(lib.v2a_debug_timestamp(_TS["buf"].data_ptr() + 8 * i, _stream()), "debug_timestamp")


def tstamp_fine(name):
    """Block-level probes (V2A_TSTAMP=2 only): one per ResNet block / residual block, forward and backward."""
    if _TS["fine"]:
        tstamp(name)


def tstamp_reset():
    _TS["cursor"] = 0


def tstamp_table():
    """[(name, microseconds since the first slot)] of the last pass."""
    if _TS["buf"] is None:
        return []
    torch.cuda.synchronize()
    v = _TS["buf"][:len(_TS["names"])].cpu().tolist()
    t0 = min(v) if v else 0
    return [(n, (x - t0) / 100.0) for n, x in zip(_TS["names"], v)]


def philox_normal(out, seed, offset_dev=None, offset_imm=0):
    check(lib.v2a_philox_normal(out.data_ptr(), out.numel(), seed, _p(offset_dev), offset_imm, _stream()), "philox_normal")
    return out


def philox_normal_rows(out, seeds, offset_imm=0):
    """out [rows, ...] ~ N(0, 1) with one Philox seed per row (seeds: int64 device tensor [rows]): row b = philox_normal of a one-row call
    with seed seeds[b]."""
    rows = out.shape[0]
    check(lib.v2a_philox_normal_rows(out.data_ptr(), rows, out.numel() // rows, seeds.data_ptr(), offset_imm, _stream()), "philox_normal_rows")
    return out


def philox_randint(out, high, seed, offset_dev=None, offset_imm=0):
    check(lib.v2a_philox_randint(out.data_ptr(), out.numel(), high, seed, _p(offset_dev), offset_imm, _stream()), "philox_randint")
    return out


def policy_sched_step(eps, sample, noise, coef, mode):
    out = torch.empty_like(sample)
    c_sb, c_sa, c0, c1, sigma = coef
    check(lib.v2a_policy_sched_step(eps.data_ptr(), sample.data_ptr(), _p(noise), out.data_ptr(), sample.numel(), c_sb, c_sa, c0, c1,
                                    sigma, mode, _stream()), "policy_sched_step")
    return out


def unnormalize_action(x, limits=None):
    out = torch.empty_like(x)
    lo, hi = limits if limits is not None else (None, None)
    check(lib.v2a_unnormalize_action(x.data_ptr(), out.data_ptr(), x.numel(), _p(lo), _p(hi), x.shape[-1], _stream()), "unnormalize_action")
    return out


def scale_by_device_scalar(x, scalar):
    """x *= scalar (a 0-dim / 1-element device tensor), in place, one kernel."""
    check(lib.v2a_scale_by_device_scalar(x.data_ptr(), x.numel(), scalar.data_ptr(), _stream()), "scale_by_device_scalar")
    return x


def video_pack(x_full, f, H, W, ci=3):
    """x_full [B, f*ci + 3, H, W] (generated frames 'b (f c) h w' then the RGB conditioning image) -> [B,f,H,W,ci+3] channels-last."""
    _chk(x_full, "x")
    B = x_full.shape[0]
    HW = H * W
    out = torch.empty((B, f, H, W, ci + 3), dtype=torch.float32, device=x_full.device)
    bs = (f * ci + 3) * HW
    check(lib.v2a_video_pack(x_full.data_ptr(), x_full.data_ptr() + 4 * f * ci * HW, out.data_ptr(), B, f, HW, bs, bs, ci, _stream()), "video_pack")
    return out


def video_pack2(img, cond, f, H, W, ci=3):
    """img [B,ci*f,H,W], cond [B,3,H,W] (separate tensors) -> [B,f,H,W,ci+3]."""
    B = img.shape[0]
    HW = H * W
    out = torch.empty((B, f, H, W, ci + 3), dtype=torch.float32, device=img.device)
    check(lib.v2a_video_pack(img.data_ptr(), cond.data_ptr(), out.data_ptr(), B, f, HW, ci * f * HW, 3 * HW, ci, _stream()), "video_pack")
    return out


def video_denoise_step(v, v_uncond, img, noise, coef, mode, final, f, HW, ci=3):
    """coef = (sa, s1, ra, rm, c1, c2, sigma, gw); see csrc/elementwise.hip video_denoise_kernel."""
    out = torch.empty_like(img)
    B = img.shape[0]
    check(lib.v2a_video_denoise_step(v.data_ptr(), _p(v_uncond), img.data_ptr(), _p(noise), out.data_ptr(), B, f, HW, *[float(c) for c in coef],
                                     mode, 1 if final else 0, ci, _stream()), "video_denoise_step")
    return out


def video_denoise_table(rows, device, out=None):
    """rows: [(sa, s1, ra, rm, c1, c2, sigma, gw, mode, final, t)] per sampler step -> device table for video_denoise_step2."""
    import numpy as np
    n = len(rows)
    host = np.zeros((n, 12), dtype=np.float32)
    iv = host.view(np.int32)
    for i, r in enumerate(rows):
        host[i, :8] = [float(c) for c in r[:8]]
        iv[i, 8:11] = [int(r[8]), 1 if r[9] else 0, int(r[10])]
    t = torch.from_numpy(host)
    if out is None:
        return t.to(device)
    assert out.shape[0] >= n
    out[:n].copy_(t)
    return out


def video_denoise_step2(v, v_uncond, img, noise, table, objective, f, HW, ci=3, state=None, step=0, use_philox=False, out=None, guided=None,
                        row_seeds=None):
    """One table-driven sampler step (csrc/elementwise.hip video_denoise_kernel2).  state: uint64[3] device tensor {row, seed, counter}
    (row index and Philox state read on the device) or None (row `step`).  out=img updates the sampler state in place.
    guided: the table's rows carry a guidance weight > 0 (default: whether v_uncond was given) -- then v_uncond is mandatory and the C
    side refuses a null pointer instead of dereferencing it."""
    if out is None:
        out = torch.empty_like(img)
    B = img.shape[0]
    if guided is None:
        guided = v_uncond is not None
    flags = (1 if use_philox else 0) | (2 if guided else 0)
    check(lib.v2a_video_denoise_step2(v.data_ptr(), _p(v_uncond), img.data_ptr(), _p(noise), out.data_ptr(), B, f, HW, ci,
                                      _OBJECTIVES[objective], table.data_ptr(), _p(state), int(step), flags, _p(row_seeds), _stream()),
          "video_denoise_step2")
    return out


def video_sampler_advance(state, table, tt, nrows):
    check(lib.v2a_video_sampler_advance(state.data_ptr(), table.data_ptr(), tt.data_ptr(), tt.numel(), nrows, _stream()), "video_sampler_advance")


def emb_linear_multi(x, ws, bs):
    """[x @ w.T + b for w, b in zip(ws, bs)] for weight matrices sharing x [B <= 16, K] in one launch (<= 32 per launch)."""
    import ctypes
    _chk(x, "x")
    B, K = x.shape
    outs = [torch.empty((B, w.shape[0]), dtype=torch.float32, device=x.device) for w in ws]
    mx = lib.v2a_emb_linear_multi_max()
    for i in range(0, len(ws), mx):
        n = len(ws[i:i + mx])
        arr = lambda vals: (ctypes.c_void_p * n)(*vals)
        check(lib.v2a_emb_linear_multi(x.data_ptr(), B, K, arr([w.data_ptr() for w in ws[i:i + n]]),
                                       arr([_p(b) for b in bs[i:i + n]]), arr([o.data_ptr() for o in outs[i:i + n]]),
                                       (ctypes.c_int * n)(*[w.shape[0] for w in ws[i:i + n]]), n, _stream()), "emb_linear_multi")
    return outs


# ---------------------------------------------------------------------------------------------- video-model training loss
_OBJECTIVES = {"pred_noise": 0, "pred_x0": 1, "pred_v": 2}


def video_qsample(img, noise, t, sqrt_acp, sqrt_1m_acp, normalize):
    """goal_diffusion.py:674-680 on 'b (f c) h w' tensors (+ forward's 2x-1, :722, when `normalize`)."""
    _chk(img, "img"); _chk(noise, "noise")
    out = torch.empty_like(img)
    B = img.shape[0]
    check(lib.v2a_video_qsample(img.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_acp.data_ptr(), sqrt_1m_acp.data_ptr(), out.data_ptr(), B,
                                img.numel() // B, 1 if normalize else 0, _stream()), "video_qsample")
    return out


def video_loss_fwd(out_cl, img, noise, t, sqrt_acp, sqrt_1m_acp, loss_weight, objective, loss_type, normalize):
    """out_cl [B,f,H,W,ci] (model output, channels-last), img / noise [B,f*ci,H,W] -> scalar loss tensor (goal_diffusion.py:699-713)."""
    B, f, H, W, ci = out_cl.shape
    loss = torch.empty((), dtype=torch.float32, device=out_cl.device)
    nb = lib.v2a_video_loss_workspace_bytes(B)
    ws = torch.empty(nb, dtype=torch.uint8, device=out_cl.device)
    check(lib.v2a_video_loss_fwd(out_cl.data_ptr(), img.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_acp.data_ptr(), sqrt_1m_acp.data_ptr(),
                                 loss_weight.data_ptr(), loss.data_ptr(), B, f, H * W, ci, _OBJECTIVES[objective], 1 if loss_type == "l1" else 0,
                                 1 if normalize else 0, ws.data_ptr(), nb, _stream()), "video_loss_fwd")
    return loss


def video_loss_bwd(out_cl, img, noise, t, sqrt_acp, sqrt_1m_acp, loss_weight, objective, loss_type, normalize, gscale=None):
    """d loss / d out_cl, times the device scalar `gscale` when given."""
    B, f, H, W, ci = out_cl.shape
    dout = torch.empty_like(out_cl)
    check(lib.v2a_video_loss_bwd(out_cl.data_ptr(), img.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_acp.data_ptr(), sqrt_1m_acp.data_ptr(),
                                 loss_weight.data_ptr(), _p(gscale), dout.data_ptr(), B, f, H * W, ci, _OBJECTIVES[objective],
                                 1 if loss_type == "l1" else 0, 1 if normalize else 0, _stream()), "video_loss_bwd")
    return dout


# ---------------------------------------------------------------------------------------------- Transformer policy backbone
def mha_fwd(q, k, v, mask, B, Tq, Tk, H, D, qoff=0, koff=0, voff=0, p_drop=0.0, seed=0, stream_id=0):
    """q / k / v: 2-D packed projection outputs (rows = B*T); *off = first column of the block.  -> out [B*Tq, H*D].
    p_drop > 0: dropout on the attention probabilities with the stateless mask (seed, stream_id) -- pass the same triple to mha_bwd."""
    out = torch.empty((B * Tq, H * D), dtype=torch.float32, device=q.device)
    check(lib.v2a_mha_fwd(q.data_ptr() + 4 * qoff, k.data_ptr() + 4 * koff, v.data_ptr() + 4 * voff, _p(mask), out.data_ptr(), B, Tq, Tk, H, D,
                          q.shape[1], k.shape[1], v.shape[1], float(p_drop), int(seed), int(stream_id), _stream()), "mha_fwd")
    return out


def mha_bwd(q, k, v, mask, dout, dq, dk, dv, B, Tq, Tk, H, D, qoff=0, koff=0, voff=0, p_drop=0.0, seed=0, stream_id=0):
    """Gradients are written into dq / dk / dv at the same column offsets (same packed layouts as q / k / v)."""
    check(lib.v2a_mha_bwd(q.data_ptr() + 4 * qoff, k.data_ptr() + 4 * koff, v.data_ptr() + 4 * voff, _p(mask), dout.data_ptr(),
                          dq.data_ptr() + 4 * qoff, dk.data_ptr() + 4 * koff, dv.data_ptr() + 4 * voff, B, Tq, Tk, H, D, q.shape[1], k.shape[1],
                          v.shape[1], float(p_drop), int(seed), int(stream_id), _stream()), "mha_bwd")


def dropout(x, p, seed, stream_id):
    """y = x * keep / (1 - p), keep decided per element by hash(seed, stream_id, index): the same call on dy is the backward."""
    _chk(x, "x")
    y = torch.empty_like(x)
    check(lib.v2a_dropout(x.data_ptr(), y.data_ptr(), x.numel(), float(p), int(seed), int(stream_id), _stream()), "dropout")
    return y
