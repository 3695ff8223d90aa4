"""Data-parallel gradient exchange of the policy train step: the ONE collective of the path.

The reference wraps the policy in torch DDP through `accelerator.prepare` (lb_online_trainer_v7.py:72-76,153-154); its hooks
all-reduce(mean) the gradients inside `accelerator.backward` (:604) before `clip_grad_norm_` (:608).  Here every gradient already
lives in one flat fp32 arena, so the exchange is a sum all-reduce of arena slices (torch.distributed 'nccl' = RCCL over xGMI on a
node; 'gloo' in the CPU tests and when ranks share one GPU) issued asynchronously in gradient-ready order, and the 1/world
averaging is folded into the consumer (the fused optimiser's gradient scale).

The class is device-agnostic on purpose: tests/test_distributed_cpu.py drives exactly this code with CPU tensors under gloo, and
PolicyTrainer drives it with the HBM arena.  There is no silent fallback: an unsupported backend raises on the first launch, and
a rank whose slice layout differs from rank 0's raises at construction (mismatched collectives would hang RCCL).

Two algorithms sit behind the one interface (`algo=`):
  "rccl"    torch.distributed all_reduce of the slice (RCCL on a node; gloo in the CPU tests) -- the default;
  "direct"  csrc/dp.hip: reduce-scatter + all-gather over hipIpc peer pointers, one stream-ordered launch per slice, sums in rank order
            (bit-identical on every rank).  For a node where RCCL runs the 349 MB message over a ring: xGMI is a full mesh, a ring keeps one
            link per direction busy, the direct exchange all seven.  The process group is used ONCE, at construction,
            to hand the 64-byte memory handles around; after that no step touches it.  bench.py --gpus N measures both and says which
            one the timed steps ran on (`comm.algo`).
"""
import ctypes
import os
import torch
import torch.distributed as dist

# hipIpc mappings of this process: handle bytes -> [mapped base, users].  One open per handle and process (a second GradReducer whose
# peers' arenas sit in the same allocator block shares the mapping).
_IPC_OPEN = {}


def _ipc_open(handle: bytes) -> int:
    from ._lib import lib, check
    ent = _IPC_OPEN.get(handle)
    if ent is None:
        base = ctypes.c_void_p()
        check(lib.v2a_dp_ipc_open(handle, ctypes.byref(base)), "dp_ipc_open (peer arena not mappable from this process)")
        ent = _IPC_OPEN[handle] = [base.value, 0]
    ent[1] += 1
    return ent[0]


def _ipc_close(handle: bytes):
    from ._lib import lib
    ent = _IPC_OPEN.get(handle)
    if ent is None:
        return
    ent[1] -= 1
    if ent[1] <= 0:
        lib.v2a_dp_ipc_close(ent[0])
        del _IPC_OPEN[handle]


def _ipc_export(ptr: int):
    from ._lib import lib, check
    h = ctypes.create_string_buffer(64)
    off, size = ctypes.c_uint64(), ctypes.c_uint64()
    check(lib.v2a_dp_ipc_export(ptr, h, ctypes.byref(off), ctypes.byref(size)), "dp_ipc_export")
    if (1 << 31) <= size.value < (1 << 32):
        # measured on this runtime (ROCm 7.2, tools/probes/r5/ipc_open_size_probe.py, profiles/r05_dp_direct.txt): a peer that maps an
        # allocation of 2 GiB <= size < 4 GiB never returns from hipIpcOpenMemHandle; 0.5 / 1 / 1.5 / 1.99 and 4 / 8 / 16 GiB map at once
        raise RuntimeError(f"the memory to share is a window of an allocation of {size.value} bytes: peers cannot map allocations of 2 to 4 GiB "
                           f"on this runtime (hipIpcOpenMemHandle does not return).  Build the arena with v2a_hip.dp.alloc_arena()")
    return h.raw, int(off.value)


# ---- how the path's independent units are dealt to the ranks (no data-path collective: rows of a sample() call, the per-task exploration
# rollouts and the replay minibatches are independent; bench.py --gpus N and the trainer use exactly these functions)
def shard_rows(batch: int, world: int, rank: int):
    """[lo, hi) rows of one B-row sample() call that rank `rank` computes: ceil(B / world) rows per rank, the last ranks may get fewer
    (or none when world > B) -- the reference samples whole batches, rows are independent (goal_diffusion.py:582-641)."""
    per = -(-int(batch) // int(world))
    lo = min(rank * per, batch)
    return lo, min(lo + per, batch)


def shard_tasks(n_tasks: int, world: int, rank: int):
    """Task ids of one video-guided exploration round (8 Libero tasks, one bs-1 rollout each: lb_online_trainer_v7.py:871,888-891) that
    rank `rank` samples: r, r + world, ..."""
    return list(range(rank, int(n_tasks), int(world)))


def joint_steps_per_sec(world: int, ms_per_step: float, seconds_per_rollout: float, n_tasks: int = 8, every: int = 200):
    """BASELINE configs[3] arithmetic: `every` data-parallel train steps (all ranks step together: world batch-64 steps per step time)
    plus one exploration round whose rollouts are dealt by shard_tasks (the slowest rank holds ceil(n_tasks / world) of them)."""
    t_round = -(-n_tasks // world) * seconds_per_rollout
    return world * every / (every * ms_per_step * 1e-3 + t_round)


class _OwnAllocation:
    """numel fp32 elements of HBM from the library's allocator (csrc/dp.hip v2a_dp_arena_alloc), seen by torch through
    __cuda_array_interface__; released (outside any graph capture) after the last tensor over it is gone."""

    def __init__(self, numel, device):
        from ._lib import lib, check
        p = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(lib.v2a_dp_arena_alloc(ctypes.byref(p), 4 * int(numel)), "dp_arena_alloc")
        self.ptr, self.device = p.value, device
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f4", "data": (self.ptr, False), "version": 2, "strides": None}

    def __del__(self):
        # Garbage collection can run at any moment -- also while some stream of the process is capturing a hipGraph, where hipFree is not
        # allowed (it invalidates the capture; torch's allocator never frees during one either).  The pointer is parked and released by the
        # next alloc_arena() / drain_arenas() call outside a capture.
        if getattr(self, "ptr", None):
            _PENDING_FREE.append((self.ptr, self.device))
            self.ptr = None


_PENDING_FREE = []


def drain_arenas():
    """Release the arenas whose tensors are gone (see _OwnAllocation.__del__).  A no-op while the current stream is capturing."""
    if not _PENDING_FREE or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return
    from ._lib import lib
    while _PENDING_FREE:
        ptr, device = _PENDING_FREE.pop()
        with torch.cuda.device(device):
            lib.v2a_dp_arena_free(ptr)


def alloc_arena(numel, device):
    """The flat fp32 gradient arena of a data-parallel trainer: zeroed, an allocation of its own (so that algo="direct" can hand exactly it
    to the peers), otherwise an ordinary torch tensor."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.zeros(int(numel), dtype=torch.float32, device=dev)
    drain_arenas()
    own = _OwnAllocation(numel, dev)
    t = torch.as_tensor(own, device=dev)
    if t.data_ptr() != own.ptr:
        raise RuntimeError("torch copied the arena instead of adopting it")
    return t


def _to_wire(src, dst):
    """fp32 slice -> bf16 wire buffer (round to nearest even): the HIP cast kernel on the device, torch on CPU tensors (gloo tests)."""
    if src.is_cuda:
        from ._lib import lib, check
        from . import ops
        n = src.numel()
        check(lib.v2a_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), n - n % 4, ops._stream()), "cast_f32_bf16")
        if n % 4:
            dst[n - n % 4:].copy_(src[n - n % 4:])
    else:
        dst.copy_(src)


def _from_wire(src, dst):
    if src.is_cuda:
        from . import ops
        dst.copy_(ops.cast_f(src))
    else:
        dst.copy_(src)


class GradReducer:
    def __init__(self, arena: torch.Tensor, slices, process_group=None, world_size=None, wire="fp32", algo="rccl", timeout_ms=30000):
        """arena: flat fp32 gradient buffer; slices: [(lo, hi), ...] element ranges in launch order (they must tile a prefix-free,
        non-overlapping part of the arena; empty ranges are skipped on every rank alike).
        wire: "fp32" (default: the reference's DDP exchanges fp32 gradients) or "bf16" -- every slice is rounded to a bf16 staging
        buffer, summed on the wire in bf16 and widened back into the arena: half the bytes per link (175 instead of 349 MB for the
        policy), at bf16 resolution of the SUM (2^-9 relative per element); an opt-in performance mode, never the parity path.
        algo: "rccl" (torch.distributed all_reduce) or "direct" (peer-pointer exchange, csrc/dp.hip; HBM arena, fp32 wire, <= 8 ranks on
        one node).  timeout_ms: how long a direct launch waits for a peer before it raises the error word (check())."""
        assert arena.dim() == 1 and arena.is_contiguous()
        if wire not in ("fp32", "bf16"):
            raise ValueError(wire)
        if algo not in ("rccl", "direct"):
            raise ValueError(algo)
        self.wire = wire
        self.algo = algo
        self.timeout_ms = int(timeout_ms)
        self.blocks = 0                    # workgroups of a direct launch (0: the library's default, 64; the same on every rank)
        self._direct = None
        self._stage = {}
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if world_size is None else int(world_size)
        self.slices = [(int(lo), int(hi)) for lo, hi in slices]
        for lo, hi in self.slices:
            if not (0 <= lo <= hi <= arena.numel()):
                raise ValueError(f"slice ({lo}, {hi}) outside the arena of {arena.numel()} elements")
        self._works = []
        self._launched = set()
        self.launches = 0
        self._last_evt = None                  # recorded by finish() behind the step's exchange launches (algo="direct"): check(sync=True)
        # a single rank with a live communicator still issues its collectives (V2A_FORCE_DP: the RCCL path on a one-GPU box)
        self.active = dist.is_available() and dist.is_initialized()
        self._agree_on_layout()
        if algo == "direct":
            self._connect_peers()

    # ---- algo="direct": the peer table
    def _connect_peers(self):
        """Exchange the hipIpc handles of every rank's arena and signal block (once), map the peers' and agree that everybody could.
        Raises on EVERY rank when any rank failed, so that no rank is left launching into a barrier nobody joins."""
        from ._lib import lib, check
        if not self.arena.is_cuda:
            raise ValueError('algo="direct" exchanges HBM arenas; CPU tensors go through algo="rccl" (gloo)')
        if self.wire != "fp32":
            raise ValueError('algo="direct" reads the peers\' fp32 arenas in place: there is no wire format to narrow')
        if not self.active:
            raise RuntimeError('algo="direct" needs an initialised process group to hand the memory handles around')
        if self.world > lib.v2a_dp_max_world():
            raise ValueError(f'algo="direct" connects at most {lib.v2a_dp_max_world()} ranks (one xGMI node), got {self.world}')
        if len(self.slices) > lib.v2a_dp_slots():
            raise ValueError(f'algo="direct" keeps at most {lib.v2a_dp_slots()} slices in flight, got {len(self.slices)}')
        rank = dist.get_rank(self.pg)
        d = self._direct = dict(rank=rank, handles=[], sig=None, err=None, epoch=[0] * len(self.slices))
        err, mine = None, None
        try:
            with torch.cuda.device(self.arena.device):
                sig = ctypes.c_void_p()
                check(lib.v2a_dp_signal_alloc(ctypes.byref(sig)), "dp_signal_alloc")
                d["sig"] = sig.value
                word = ctypes.c_void_p()
                check(lib.v2a_dp_errword_alloc(ctypes.byref(word)), "dp_errword_alloc")
                d["err"] = word.value
                ah, aoff = _ipc_export(self.arena.data_ptr())
                sh, soff = _ipc_export(sig.value)
            mine = dict(rank=rank, pid=os.getpid(), arena=(ah, aoff), signal=(sh, soff), numel=self.arena.numel())
        except Exception as e:                                     # noqa: BLE001 -- reported collectively below
            err = f"rank {rank}: {e}"
            mine = dict(rank=rank, error=err)
        table = [None] * self.world
        dist.all_gather_object(table, mine, group=self.pg)
        table = sorted(table, key=lambda t: t["rank"])
        arenas, signals = (ctypes.c_void_p * self.world)(), (ctypes.c_void_p * self.world)()
        # One rank maps at a time, the others sit in the barrier: mapping a peer's allocation is served by a runtime thread of the EXPORTING
        # process, and two processes that map each other's memory at the same moment were measured to block each other for good
        # (bench.py --gpus 2 on one GPU, round 5: both ranks inside hipIpcOpenMemHandle).
        healthy = err is None and not any("error" in t for t in table)
        for turn in range(self.world):
            if healthy and turn == rank:
                try:
                    for p, t in enumerate(table):
                        if t["numel"] != self.arena.numel():
                            raise RuntimeError(f"rank {p} holds an arena of {t['numel']} elements, this rank {self.arena.numel()}")
                        if p == rank:
                            arenas[p], signals[p] = self.arena.data_ptr(), d["sig"]
                            continue
                        with torch.cuda.device(self.arena.device):
                            for key, out in (("arena", arenas), ("signal", signals)):
                                h, off = t[key]
                                out[p] = _ipc_open(h) + off
                                d["handles"].append(h)
                except Exception as e:                             # noqa: BLE001
                    err = f"rank {rank}: {e}"
            dist.barrier(group=self.pg)
        # collective verdict (object gather again: works on every backend and carries the reason)
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, err or next((t["error"] for t in table if "error" in t), None), group=self.pg)
        bad = [v for v in verdicts if v]
        if bad:
            self.close(collective=False)
            raise RuntimeError('algo="direct": the peer table could not be built: ' + "; ".join(sorted(set(bad))))
        d["arenas"], d["signals"] = arenas, signals

    def check(self, sync=False):
        """Raise if a direct launch gave up waiting for a peer (the kernel raises a pinned host word and terminates; results of that step
        are garbage).  Read without synchronising by launch() and finish() -- a give-up in step N is then seen at step N + 1's launch, AFTER
        the optimiser consumed the garbage.  sync=True first waits for the event finish() recorded behind the step's last exchange launch:
        call it before anything that must not see such a step (a checkpoint write, a logged loss): `PolicyTrainer.verify_exchange()`."""
        if sync and self._last_evt is not None:
            self._last_evt.synchronize()
        d = self._direct
        if d is not None and d.get("err"):
            w = (ctypes.c_int * 5).from_address(d["err"])
            if w[0]:
                raise RuntimeError(f'algo="direct": rank {d["rank"]} waited {self.timeout_ms} ms for rank {w[0] - 1} in the gradient exchange '
                                   f"and gave up (a rank died, or the ranks issued different launches): slice {w[1]}, launch {w[2] // 3}, "
                                   f"barrier {w[2] % 3 or 3} of 3, workgroup {w[3]}, flag at {w[4]} of {w[2]}; this rank has issued "
                                   f'{d["epoch"]} launches per slice')

    def close(self, collective=True):
        """Unmap the peers and free the signal block.  collective=True: every rank calls it; they first drain their GPUs and meet, so
        nobody unmaps memory a peer's kernel still reads."""
        d, self._direct = self._direct, None
        if d is None:
            return
        from ._lib import lib
        if self.arena.is_cuda:
            torch.cuda.synchronize(self.arena.device)
        if collective and self.active and self.world > 1:
            dist.barrier(group=self.pg)
        for h in d["handles"]:
            _ipc_close(h)
        if d.get("sig"):
            lib.v2a_dp_signal_free(d["sig"])
        drain_arenas()                     # arenas parked by dead trainers (349 MB each) go back now, not at the next alloc_arena()
        if d.get("err"):
            lib.v2a_dp_errword_free(d["err"])

    def _agree_on_layout(self):
        """Every rank must issue the same collectives in the same order: compare the slice table with rank 0's (one tiny
        all-reduce pair, at construction only)."""
        if self.world <= 1:
            return
        desc = [self.arena.numel(), len(self.slices)] + [v for s in self.slices for v in s]
        dev = self.arena.device if self.arena.is_cuda and dist.get_backend(self.pg) != "gloo" else "cpu"
        mine = torch.tensor(desc, dtype=torch.int64, device=dev)
        n = torch.tensor([mine.numel()], dtype=torch.int64, device=dev)
        nmax, nmin = n.clone(), n.clone()
        dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=self.pg)
        dist.all_reduce(nmin, op=dist.ReduceOp.MIN, group=self.pg)
        if int(nmax) != int(nmin):
            raise RuntimeError(f"data-parallel ranks disagree on the number of gradient slices ({int(nmin)} vs {int(nmax)} table entries)")
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        if not torch.equal(hi, lo):
            raise RuntimeError(f"data-parallel ranks disagree on the gradient arena layout: this rank {desc}, "
                               f"element-wise min {lo.tolist()}, max {hi.tolist()}")

    def launch(self, which: int):
        """Start the sum all-reduce of slice `which` (asynchronous; the caller's stream keeps running)."""
        if which in self._launched:
            raise RuntimeError(f"slice {which} was already launched in this step")
        self._launched.add(which)
        lo, hi = self.slices[which]
        if hi <= lo or not self.active:
            return
        if self.algo == "direct":
            from ._lib import lib, check
            from . import ops
            d = self._direct
            if d is None:
                raise RuntimeError("this GradReducer was closed")
            self.check()
            d["epoch"][which] += 1
            check(lib.v2a_dp_allreduce_direct(d["arenas"], d["signals"], self.world, d["rank"], lo, hi, which, d["epoch"][which], d["err"],
                                              self.timeout_ms, self.blocks, ops._stream()), "dp_allreduce_direct")
            self.launches += 1
            return
        if self.wire == "bf16":
            st = self._stage.get(which)
            if st is None:
                st = self._stage[which] = torch.empty(hi - lo, dtype=torch.bfloat16, device=self.arena.device)
            _to_wire(self.arena[lo:hi], st)
            self._works.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), which))
        else:
            self._works.append((dist.all_reduce(self.arena[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), None))
        self.launches += 1

    def finish(self, scale_fn=None):
        """Wait for every launched slice (on a GPU: makes the current stream wait on the communicator's stream), then average:
        `scale_fn(1/world)` (the fused optimiser's gradient scale) or an in-place multiply of the reduced slices."""
        if len(self._launched) != len(self.slices):
            missing = sorted(set(range(len(self.slices))) - self._launched)
            raise RuntimeError(f"finish() before slices {missing} were launched: ranks would issue different collectives")
        self.check()                       # direct: the launches are stream-ordered, there is nothing to wait for on the host
        if self.algo == "direct" and self.arena.is_cuda and not torch.cuda.is_current_stream_capturing():
            self._last_evt = torch.cuda.Event()
            self._last_evt.record()
        for w, staged in self._works:
            w.wait()
            if staged is not None:
                lo, hi = self.slices[staged]
                _from_wire(self._stage[staged], self.arena[lo:hi])
        self._works = []
        self._launched = set()
        if self.world > 1:
            if scale_fn is not None:
                scale_fn(1.0 / self.world)
            else:
                for lo, hi in self.slices:
                    self.arena[lo:hi].mul_(1.0 / self.world)

    def abort(self):
        """Abandon the step in flight (a non-finite loss, an exception between launch and finish): wait for the collectives that were
        already issued -- every rank issued the same ones up to here, so nothing hangs -- and forget them WITHOUT averaging or widening
        into the arena, so that the next step can launch its slices again.  The decision to skip a step must be collective (all ranks
        call abort() for the same step); a rank that finishes while another aborts would be one collective ahead."""
        for w, _ in self._works:
            w.wait()
        self._works = []
        self._launched = set()

    def pending(self):
        """Slices launched since the last finish()."""
        return set(self._launched)

    def bytes_per_step(self):
        return (2 if self.wire == "bf16" else 4) * sum(hi - lo for lo, hi in self.slices)
