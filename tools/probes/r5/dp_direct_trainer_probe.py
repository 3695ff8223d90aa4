"""Two ranks on one GPU, PolicyTrainer(dp_algo="direct"), rank 1 deliberately late by `skew_ms` per step: does rank 0's waiting exchange
kernel keep rank 1's backward (whole-CU conv kernels) from running?  argv: sync_each graph blocks skew_ms.
Measured (round 5): 256 workgroups (one per CU) -> rank 1's encoder backward does not finish until rank 0's kernel gives up: both ranks
time out; 64 workgroups -> every step completes."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))


def worker(rank, world, port, sync_each, graph, blocks, skew_ms):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import random
    from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
    from v2a_hip.replay import ReplayStore
    from v2a_hip.trainer import PolicyTrainer
    torch.manual_seed(1)
    pol = build_policy(DEFAULT_CONF).to("cuda:0")
    store = ReplayStore(64, 200, 30, capacity_frames=40 * 12)
    gen = torch.Generator().manual_seed(3 + rank)
    for e in range(12):
        n = 30 + e
        store.add_one_episode("t", "agentview", e, torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=gen),
                              torch.rand(n - 1, 7, generator=gen) * 2 - 1)
    np.random.seed(5 + rank); random.seed(5 + rank)
    tr = PolicyTrainer(pol, store, batch_size=4, seed=11, use_graph=graph, process_group=dist.group.WORLD, world_size=world, rank=rank,
                       dp_algo="direct")
    tr.reducer.timeout_ms = 3000
    tr.reducer.blocks = blocks
    for i in range(6):
        if rank == 1 and skew_ms:
            time.sleep(skew_ms * 1e-3)
        t0 = time.time()
        try:
            tr.step().item()
            if sync_each:
                torch.cuda.synchronize()
                tr.reducer.check()
            print(f"rank {rank} step {i} ok host {time.time() - t0:.2f}s", flush=True)
        except Exception as e:
            print(f"rank {rank} step {i} FAILED after {time.time() - t0:.2f}s: {e}", flush=True)
            import ctypes
            ctypes.c_int.from_address(tr.reducer._direct["err"]).value = 0
    torch.cuda.synchronize()
    try:
        tr.reducer.check()
        print(f"rank {rank} end ok", flush=True)
    except Exception as e:
        print(f"rank {rank} end FAILED: {e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    sync_each, graph, blocks, skew_ms = (int(a) for a in sys.argv[1:5])
    mp.spawn(worker, args=(2, 29655 + sync_each * 2 + graph, bool(sync_each), bool(graph), blocks, skew_ms), nprocs=2, join=True)
