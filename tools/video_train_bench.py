"""Time one optimisation step of the full-size AVDC video model (Unet_Libero, 201 M parameters, 7 frames of 128x128) on one MI355X:
q_sample -> UNet forward with tape -> loss -> hand-written backward -> clip + Adam + EMA.   python tools/video_train_bench.py [--batch 2]"""
import argparse
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tokens", type=int, default=8)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32", help="MFMA input precision (fp32 storage either way)")
    a = ap.parse_args()
    import copy
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    from v2a_hip.video_train import VideoTrainStep
    from v2a_hip import ops
    import v2a_hip
    v2a_hip.set_precision(a.precision)
    torch.manual_seed(0)
    m = Unet_Libero().to("cuda:0")
    d = GoalGaussianDiffusion(m, image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=100, loss_type="l2", objective="pred_v",
                              beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0).to("cuda:0")
    ema = copy.deepcopy(d).requires_grad_(False)
    ts = VideoTrainStep(d, ema)
    B = a.batch
    img = torch.rand(B, 21, 128, 128, device="cuda:0")
    cond = torch.rand(B, 3, 128, 128, device="cuda:0")
    te = torch.randn(B, a.tokens, 512, device="cuda:0")
    for _ in range(a.warmup):
        loss = ts.step(img, cond, te)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = ts.step(img, cond, te)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # phase split
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = ts.loss_and_grads(img, cond, te)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ts.apply()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print({"precision": a.precision, "batch": B, "ms_per_step": dt * 1e3, "samples_per_s": B / dt, "loss": float(loss), "fwd_bwd_ms": (t1 - t0) * 1e3,
           "opt_ms": (t2 - t1) * 1e3, "peak_GB": torch.cuda.max_memory_allocated() / 2 ** 30})


if __name__ == "__main__":
    main()
