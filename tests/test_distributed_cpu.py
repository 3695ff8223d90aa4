"""world_size-2 gloo test of the data-parallel path: the gradient arena all-reduce + 1/world scaling + identical replicas.
The HIP kernels cannot run here; what is exercised is exactly the host logic PolicyTrainer uses around them: a flat fp32 arena,
ONE sum all-reduce, averaging folded into the optimiser, per-rank RNG / replay shards, replicas staying bit-identical."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import optim as O
    from v2a_hip.replay import sample_indices
    import random
    torch.manual_seed(0)                                     # identical replica init
    params = [torch.randn(300), torch.randn(17, 5)]
    total = sum(p.numel() for p in params)
    arena = torch.zeros(total)
    views, off = [], 0
    for p in params:
        views.append(arena[off:off + p.numel()].view(p.shape)); off += p.numel()
    ms, vs = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    em = [p.clone() for p in params]
    st = O.EmaState(power=0.75)
    np.random.seed(100 + rank); random.seed(100 + rank)      # per-rank replay stream (trainer: seed + rank)
    lens = np.full(12, 121, dtype=np.int32)
    for step in range(1, 4):
        ep, start = sample_indices(lens, 8, 16)              # native sampler on this rank's own generator states
        g = torch.Generator().manual_seed(1000 * rank + step)
        for v in views:
            v.copy_(torch.randn(v.shape, generator=g) + float(ep.sum() % 7))
        dist.all_reduce(arena, op=dist.ReduceOp.SUM)         # the ONE collective of the path (RCCL on the GPU box)
        arena.mul_(1.0 / world)                              # v2a_opt_scale_grads
        O.train_tail(params, [v.clone() for v in views], ms, vs, em, step, st)
    gathered = [torch.zeros(total) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([p.flatten() for p in params]))
    if rank == 0:
        out.put((gathered[0].numpy(), gathered[1].numpy(), [ep.tolist(), start.tolist()]))
    else:
        out.put(("idx", [ep.tolist(), start.tolist()]))
    dist.destroy_process_group()


def test_dp2_replicas_stay_identical_and_shards_differ():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = [r for r in res if not isinstance(r[0], str)][0]
    other = [r for r in res if isinstance(r[0], str)][0]
    assert np.array_equal(full[0], full[1])                  # parameters bit-identical across ranks after 3 steps
    assert full[2] != other[1]                               # ranks drew different replay windows


def _video_worker(rank, world, port, out):
    """Host logic of v2a_hip.video_train.VideoTrainStep.apply for world > 1: the gradient arena the hand-written backward fills is ONE flat
    tensor (views per parameter), summed by one all-reduce and averaged before the optimiser."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2a_hip.video_train import _GradArena
    torch.manual_seed(0)
    params = {"unet.a.weight": torch.randn(8, 4, 3, 3), "unet.a.bias": torch.randn(8), "unet.b.weight": torch.randn(5, 8)}
    arena = _GradArena(params)
    assert arena.flat.numel() == sum(p.numel() for p in params.values()) and list(arena.views) == list(params)
    for n, v in arena.views.items():                                  # views alias the flat buffer in named_parameters() order
        assert v.shape == params[n].shape and v.data_ptr() >= arena.flat.data_ptr()
        v.copy_(torch.full(v.shape, float(rank + 1)) * (1 + len(n)))
    dist.all_reduce(arena.flat, op=dist.ReduceOp.SUM)
    arena.flat.mul_(1.0 / world)                                      # FusedAdamWEMA.scale_grads(1 / world) on the GPU
    out.put((rank, {n: float(v.mean()) for n, v in arena.views.items()}))
    dist.destroy_process_group()


def test_video_trainer_arena_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + os.getpid() % 500
    procs = [ctx.Process(target=_video_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    for n, v in res[0].items():
        assert abs(v - 1.5 * (1 + len(n))) < 1e-6                     # mean of rank values (1, 2) x the per-parameter factor
