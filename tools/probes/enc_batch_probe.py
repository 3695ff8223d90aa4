"""What would batching the two camera encoders into single launches buy?  One encoder at B = 128 runs exactly the launches a batched
pair at B = 64 would (same tiles; only the weight pointer differs per half) -- time its forward and backward (hipGraph replay) against
the two-stream B = 64 pair of the real step (tools/phase_clock.py: enc_fwd 2.07 ms, enc_bwd 4.7 ms in fp32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
import v2a_hip
from v2a_hip import ops
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
v2a_hip.set_precision(prec)
dev = "cuda:0"
torch.manual_seed(0)
pol = build_policy(DEFAULT_CONF).to(dev)
eng = pol.engine
names = pol.trainable_names()
for B in (64, 128):
    img = torch.rand(B, 3, 128, 128, device=dev)
    arena = torch.zeros(eng.grad_layout(names)[1], device=dev)
    grads = eng.grad_views(arena, names)
    eng._dw_names = {v.data_ptr(): n for n, v in grads.items()}
    key = eng.cfg.rgb_keys[0]
    eng._cur_batch = B
    eng.refresh_packs()

    def fwd():
        save = {}
        f = eng.encode_fwd(key, img, save)
        return f, save

    def bwd(f, save):
        df = torch.ones_like(f)
        eng.encode_bwd(key, df, save[key], grads)

    for _ in range(2):
        f, save = fwd(); bwd(f, save)
    torch.cuda.synchronize()
    gf = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf):
        f, save = fwd()
    gb = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gb, pool=gf.pool()):
        bwd(f, save)
    for g, tag in ((gf, "fwd"), (gb, "bwd")):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{prec} one encoder B={B} {tag}: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
