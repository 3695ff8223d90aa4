"""world_size-2 gloo tests of the data-parallel path.  The HIP kernels cannot run here; what runs is the product's own exchange
code -- v2a_hip.dp.GradReducer, the object PolicyTrainer drives on the GPU (two asynchronous slice all-reduces of one flat fp32
arena in gradient-ready order, averaging folded into the consumer) -- plus the native replay sampler on per-rank generator states:
gradient equality (N ranks x B/N rows == 1 rank x B rows, the reference's DDP mean: lb_online_trainer_v7.py:604-608), replicas
staying bit-identical, and loud failure when ranks disagree on the collective layout."""
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
    from v2a_hip.dp import GradReducer
    red = GradReducer(arena, [(300, total), (0, 300)], dist.group.WORLD, world)      # second tensor's slice is ready first
    for step in range(1, 4):
        ep, start = sample_indices(lens, 8, 16)              # native sampler on this rank's own generator states
        g = torch.Generator().manual_seed(1000 * rank + step)
        views[1].copy_(torch.randn(views[1].shape, generator=g) + float(ep.sum() % 7))
        red.launch(0)                                        # travels while the rest of the "backward" runs
        views[0].copy_(torch.randn(views[0].shape, generator=g) + float(ep.sum() % 7))
        red.launch(1)
        red.finish()                                         # wait + 1/world (the fused optimiser's gradient scale on the GPU)
        O.train_tail(params, [v.clone() for v in views], ms, vs, em, step, st)
    assert red.launches == 6 and red.bytes_per_step() == 4 * total
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


def _equality_worker(rank, world, port, out):
    """Gradient equality: rank r holds the gradient of the mean loss over ITS half of the rows; sum / world must equal the
    gradient of the mean over all rows, which is what clip + AdamW then consume."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2a_hip.dp import GradReducer
    torch.manual_seed(3)
    w = torch.randn(40, 9, dtype=torch.float64)
    x = torch.randn(8, 9, dtype=torch.float64)
    y = torch.randn(8, 40, dtype=torch.float64)

    def grad(rows):                                           # d/dw mean((x w^T - y)^2) over the given rows
        r = x[rows] @ w.T - y[rows]
        return (2.0 / r.numel()) * r.T @ x[rows]

    full = grad(slice(0, 8)).float().flatten()
    arena = torch.zeros(360)
    arena.copy_(grad(slice(4 * rank, 4 * rank + 4)).float().flatten())
    red = GradReducer(arena, [(0, 200), (200, 360)], dist.group.WORLD, world)
    red.launch(0); red.launch(1)
    try:
        red.launch(1)
        raised = False
    except RuntimeError:
        raised = True
    red.finish()
    out.put((rank, float((arena - full).abs().max()), float(full.abs().max()), raised))
    dist.destroy_process_group()


def test_dp_gradient_equality_two_ranks_vs_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30700 + os.getpid() % 500
    procs = [ctx.Process(target=_equality_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale, raised in res:
        assert err <= 1e-6 * scale, (rank, err, scale)
        assert raised                                         # launching a slice twice in one step is refused


def _mismatch_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2a_hip.dp import GradReducer
    arena = torch.zeros(100)
    res = []
    for slices in ([(0, 60 + rank), (60 + rank, 100)], [(0, 50)] if rank == 0 else [(0, 25), (25, 50)]):
        try:
            GradReducer(arena, slices, dist.group.WORLD, world)
            res.append("accepted")
        except RuntimeError as e:
            res.append(str(e)[:40])
    out.put((rank, res))
    dist.destroy_process_group()


def test_dp_layout_disagreement_fails_loudly_on_every_rank():
    """a per-rank fallback decision would issue mismatched collectives (a hang under RCCL).  There is no fallback any
    more, and ranks whose slice tables differ -- in boundaries or in count -- refuse to start, all of them."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31300 + os.getpid() % 500
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert all(r.startswith("data-parallel ranks disagree") for r in res[rank]), res


def _w4_worker(rank, world, port, out, wire):
    """Four ranks, three slices launched in gradient-ready order at DIFFERENT host times per rank (uneven readiness: a rank that is
    late with slice k must not let anybody's slice k+1 overtake it), fp32 or bf16 wire."""
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2a_hip.dp import GradReducer
    from v2a_hip.video_train import gradient_ready_slices
    names = ["unet.time_embed.0.weight", "unet.input_blocks.0.0.weight", "unet.middle_block.0.w", "unet.output_blocks.0.0.w", "unet.out.2.bias"]
    numels = [1000, 5000, 3000, 7001, 13]
    sl = gradient_ready_slices(names, numels)
    assert sl == [(9000, 16014), (0, 9000)]
    total = sum(numels)
    arena = torch.zeros(total)
    red = GradReducer(arena, [sl[0], (1000, 9000), (0, 1000)], dist.group.WORLD, world, wire=wire)
    res = []
    for step in range(3):
        g = torch.Generator().manual_seed(17 * rank + step)
        vals = torch.randn(total, generator=g)
        for k, (lo, hi) in enumerate(red.slices):
            time.sleep(0.02 * ((rank + k + step) % 4))          # uneven readiness
            arena[lo:hi] = vals[lo:hi]
            red.launch(k)
        assert red.pending() == {0, 1, 2}
        red.finish()
        res.append(arena.clone())
    assert red.bytes_per_step() == (2 if wire == "bf16" else 4) * total
    out.put((rank, [r.numpy() for r in res]))
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_dp4_uneven_slice_readiness_and_wire_format(wire):
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30700 + os.getpid() % 500 + (7 if wire == "bf16" else 0)
    procs = [ctx.Process(target=_w4_worker, args=(r, world, port, q, wire)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = 16014
    for step in range(3):
        want = sum(torch.randn(total, generator=torch.Generator().manual_seed(17 * r + step)) for r in range(world)) / world
        for r in range(world):
            got = torch.from_numpy(res[r][step])
            assert np.array_equal(res[r][step], res[0][step])                 # every replica holds the same averaged gradient
            if wire == "fp32":
                assert torch.allclose(got, want, rtol=0, atol=1e-6)
            else:                                                              # bf16 wire: inputs rounded to bf16, sums rounded per hop
                tol = 4 * 2.0 ** -8 * float(want.abs().max() + 1)
                assert float((got - want).abs().max()) <= tol and float((got - want).abs().max()) > 0


def _shard_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2a_hip.dp import shard_rows, shard_tasks, joint_steps_per_sec
    lo, hi = shard_rows(16, world, rank)
    tasks = shard_tasks(8, world, rank)
    # every rank "samples" its rows and its tasks' rollouts: a row / task is represented by a value only its owner can produce
    rows = torch.zeros(16)
    rows[lo:hi] = torch.arange(lo, hi, dtype=torch.float32) + 1000.0
    tk = torch.zeros(8)
    for t in tasks:
        tk[t] = 100.0 + t
    owners_r, owners_t = torch.zeros(16), torch.zeros(8)
    owners_r[lo:hi] = 1
    for t in tasks:
        owners_t[t] = 1
    for v in (rows, tk, owners_r, owners_t):
        dist.all_reduce(v)                                  # (test bookkeeping only: the product path has no collective here)
    # the slowest rank decides the round (MAX over ranks, as bench.py does with its timings)
    mine = torch.tensor([float(len(tasks)) * 0.63])         # 0.63 s per bs-1 rollout (round-3 measurement)
    dist.all_reduce(mine, op=dist.ReduceOp.MAX)
    rate = joint_steps_per_sec(world, 10.0, 0.63)
    if rank == 0:
        out.put((rows.tolist(), tk.tolist(), owners_r.tolist(), owners_t.tolist(), float(mine), rate))
    dist.destroy_process_group()


def test_world8_sampler_rows_and_exploration_tasks_partition_without_a_collective():
    """BASELINE configs[3] / [2] at 8 ranks: the B = 16 rows of a sample() call and the 8 per-task exploration rollouts are dealt by
    v2a_hip.dp.shard_rows / shard_tasks (what bench.py --gpus 8 and the joint loop use): every row and every task has exactly one owner,
    nothing is dropped, and the joint-loop rate is the arithmetic of the slowest rank's share."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 8
    port = 29900 + os.getpid() % 50
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rows, tk, own_r, own_t, t_round, rate = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert own_r == [1.0] * 16 and own_t == [1.0] * 8
    assert rows == [1000.0 + i for i in range(16)] and tk == [100.0 + t for t in range(8)]
    assert abs(t_round - 0.63) < 1e-6                       # one rollout per rank at world 8
    assert abs(rate - 8 * 200 / (200 * 10.0e-3 + 0.63)) < 1e-9
    from v2a_hip.dp import shard_rows, shard_tasks
    for w in (1, 2, 3, 4, 5, 8, 16, 32):                     # ragged worlds: still a partition
        cover = sorted(i for r in range(w) for i in range(*shard_rows(16, w, r)))
        assert cover == list(range(16)), (w, cover)
        assert sorted(t for r in range(w) for t in shard_tasks(8, w, r)) == list(range(8))


def test_direct_exchange_is_for_hbm_arenas_only():
    """algo="direct" maps HBM between ranks: a CPU arena is refused at construction (no silent change of algorithm); alloc_arena on the
    CPU is an ordinary zeroed tensor."""
    from v2a_hip.dp import GradReducer, alloc_arena
    a = alloc_arena(100, "cpu")
    assert a.dtype == torch.float32 and a.numel() == 100 and float(a.abs().sum()) == 0.0
    with pytest.raises(ValueError, match="HBM"):
        GradReducer(a, [(0, 100)], None, 1, algo="direct")
    with pytest.raises(ValueError):
        GradReducer(a, [(0, 100)], None, 1, algo="ring")
