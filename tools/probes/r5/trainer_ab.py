"""A/B of a PolicyTrainer / PolicyEngine attribute inside ONE process (boxes differ by up to 0.3 ms per step): alternating builds of the
B = 64 trainer with the attribute at each given value, best of three 30-step timings each.
Usage: python tools/probes/r5/trainer_ab.py trainer|engine ATTR V0 V1 [reps]"""
import os, sys, time, random, ast
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import numpy as np, torch, bench
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from v2a_hip.trainer import PolicyTrainer
where, attr = sys.argv[1], sys.argv[2]
vals = [ast.literal_eval(v) for v in sys.argv[3:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
for v in vals * reps:
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    pol = build_policy(DEFAULT_CONF).to("cuda:0")
    store = bench.build_store(torch, "cuda:0", 64, seed=100)
    if where == "engine":
        setattr(pol.engine, attr, v)
    if where == "ops":
        from v2a_hip import ops
        getattr(ops, attr)[0] = v
    tr = PolicyTrainer(pol, store, batch_size=64, seed=0, use_graph=True)
    if where == "trainer":
        setattr(tr, attr, v)
    for _ in range(6): tr.step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30): tr.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"{where}.{attr} = {v}: policy step {min(ts):.3f} ms {['%.3f' % t for t in ts]} loss {float(tr.loss.item()):.6f}", flush=True)
    del tr, pol, store
    torch.cuda.empty_cache()
