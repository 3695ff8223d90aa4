"""Flip-aware gradient comparison for the policy's image encoders.

The ResNet18-GN encoders take ~0.6 M discrete decisions per image (ReLU masks, 3x3 max-pool winners).  Two correct fp32 implementations
agree on all of them except where a pre-activation is within rounding of zero / two window elements are within rounding of each other, and
ONE decision taken the other way moves a small weight gradient by 1 / (B H W) of its terms -- 1e-3 ... 4e-2 at B <= 8.  Instead of a loose
bound, the comparison is made exact in two independent parts:
  (1) ROUTING: the HIP forward's decisions (PolicyEngine.debug_decisions) against the CPU oracle's own (oracle.policy dec={"record": ..}):
      at most FRAC of them differ, and every differing one is a genuine tie on the oracle's pre-activations (|z| or the gap between the two
      winners <= TIE x max|z| of that tensor);
  (2) ARITHMETIC: the oracle's backward is routed through the HIP forward's decisions (dec={"use": ..}); every gradient tensor must then
      meet the strict bound -- no tensor is excused.
"""
import torch

FRAC = 1e-5        # share of all decisions of a batch that may differ between the two forwards
TIE = 3e-5         # a differing decision must sit on a pre-activation (gap) this close to zero, relative to the tensor's largest |z|


def hip_decisions(eng_dec):
    """PolicyEngine.debug_decisions (NHWC, device) -> the oracle's layout (NCHW, CPU; masks bool, pool winners int64 taps 0..8)."""
    out = {}
    for k, v in eng_dec.items():
        t = v.permute(0, 3, 1, 2).contiguous().cpu()
        out[k] = t.long() if k.endswith(".pool") else t.bool()
    return out


def check_routing(rec, use, tag=""):
    """rec: dec["record"] of a free oracle forward; use: hip_decisions(...).  Returns (differing, total, largest tie distance seen)."""
    n_diff = n_tot = 0
    worst = 0.0
    for name, mine in use.items():
        ref = rec[name]
        z = rec[name + ":z"].double()
        assert ref.shape == mine.shape, (name, ref.shape, mine.shape)
        n_tot += ref.numel()
        d = ref != mine
        k = int(d.sum())
        if k == 0:
            continue
        n_diff += k
        zmax = float(z.abs().max())
        if name.endswith(".pool"):
            N, C, H, W = z.shape
            OH, OW = ref.shape[2], ref.shape[3]
            oh = torch.arange(OH).view(1, 1, OH, 1)
            ow = torch.arange(OW).view(1, 1, 1, OW)

            def val(tap):
                ih, iw = 2 * oh - 1 + tap // 3, 2 * ow - 1 + tap % 3
                ok = (ih >= 0) & (ih < H) & (iw >= 0) & (iw < W)
                v = z.flatten(2).gather(2, (ih.clamp(0, H - 1) * W + iw.clamp(0, W - 1)).flatten(2)).view(N, C, OH, OW)
                return torch.where(ok, v, torch.full_like(v, -1e30))
            gap = (val(ref) - val(mine)).abs()[d]
        else:
            gap = z.abs()[d]
        g = float(gap.max()) / max(zmax, 1e-30)
        worst = max(worst, g)
        assert g <= TIE, f"{tag}{name}: {k} decisions differ and the furthest is no tie ({g:.2e} of max |z|)"
    assert n_diff <= max(1, int(FRAC * n_tot)), f"{tag}{n_diff} of {n_tot} decisions differ"
    try:
        from conftest import parity_record
        parity_record(f"{tag}routing: share of decisions differing", n_diff / max(n_tot, 1), FRAC, largest_tie_distance=worst, tie_bound=TIE)
    except ImportError:
        pass
    return n_diff, n_tot, worst
