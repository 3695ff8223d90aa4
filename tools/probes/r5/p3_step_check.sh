set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_policy_gpu.py -q -x -k "presplit or graph_replay or operand_packs or presummed or golden_and_oracle or three_train" > gpurun_out/r5g_tests.txt 2>&1
tail -15 gpurun_out/r5g_tests.txt
python - > gpurun_out/r5g_step.txt 2>&1 <<'PY'
import os, sys, time, random
sys.path.insert(0, "."); sys.path.insert(0, "video-to-action-release_amd")
import numpy as np, torch, bench
from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
from v2a_hip.trainer import PolicyTrainer
for use in (True, False, True, False):
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    pol = build_policy(DEFAULT_CONF).to("cuda:0")
    pol.engine.use_p3 = use
    store = bench.build_store(torch, "cuda:0", 64, seed=100)
    tr = PolicyTrainer(pol, store, batch_size=64, seed=0, use_graph=True)
    for _ in range(6): tr.step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30): tr.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"use_p3={use}: policy step {min(ts):.3f} ms {['%.3f' % t for t in ts]} loss {float(tr.loss.item()):.6f}", flush=True)
    del tr, pol, store
    torch.cuda.empty_cache()
PY
cat gpurun_out/r5g_step.txt
