"""Per-shape time of one op family inside a full-size video training step (GPU box).  python tools/video_train_shapes.py [wgrad|conv|colsum] [--batch 2]"""
import argparse
import collections
import copy
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="wgrad")
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    from v2a_hip.video_train import VideoTrainStep
    from v2a_hip import ops
    import v2a_hip
    v2a_hip.set_precision(os.environ.get("V2A_PRECISION", "fp32"))
    torch.manual_seed(0)
    m = Unet_Libero().to("cuda:0")
    d = GoalGaussianDiffusion(m, image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=100, loss_type="l2", objective="pred_v",
                              beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0).to("cuda:0")
    ts = VideoTrainStep(d, None)
    B = a.batch
    img, cond, te = torch.rand(B, 21, 128, 128, device="cuda:0"), torch.rand(B, 3, 128, 128, device="cuda:0"), torch.randn(B, 8, 512, device="cuda:0")
    ts.step(img, cond, te)
    agg = collections.OrderedDict()
    target = {"wgrad": "conv2d_wgrad", "conv": "conv2d", "colsum": "colsum_batched"}[a.what]
    orig = getattr(ops, target)

    def timed(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*args, **kw)
        e1.record()
        e1.synchronize()
        if a.what == "wgrad":
            x, dy, wshape = args[0], args[1], args[2]
            M = dy.shape[0] * dy.shape[1] * dy.shape[2]
            key = (M, wshape[1] * args[3] * args[4], wshape[0], tuple(kw.get("stride", (1, 1))), bool(kw.get("ups", False)))
            fl = 2.0 * M * key[1] * key[2]
        elif a.what == "conv":
            x = args[0]
            y = out
            M = y.shape[0] * y.shape[1] * y.shape[2]
            K = x.shape[-1] * args[4] * args[5]
            key = (M, K, args[3], kw.get("idil", 1), bool(kw.get("ups", False)))
            fl = 2.0 * M * K * args[3]
        else:
            key = tuple(args[0].shape)
            fl = args[0].numel() * 4.0
        ent = agg.setdefault(key, [0, 0.0, fl, ops.last_kernel[0] if a.what != "colsum" else ""])
        ent[0] += 1
        ent[1] += e0.elapsed_time(e1)
        return out

    setattr(ops, target, timed)
    ts.step(img, cond, te)
    setattr(ops, target, orig)
    tot = sum(v[1] for v in agg.values())
    print(f"{a.what}: total {tot:.2f} ms over {sum(v[0] for v in agg.values())} calls")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        unit = "GB/s" if a.what == "colsum" else "TF"
        rate = v[2] / (v[1] / v[0] * 1e-3) / (1e9 if a.what == "colsum" else 1e12)
        print(f"{str(k):48s} n={v[0]:3d} total {v[1]:8.3f} ms  avg {v[1]/v[0]*1e3:8.1f} us  {rate:7.1f} {unit}  {v[3]}")


if __name__ == "__main__":
    main()
