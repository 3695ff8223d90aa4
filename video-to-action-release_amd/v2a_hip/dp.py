"""Data-parallel gradient exchange of the policy train step: the ONE collective of the path.

The reference wraps the policy in torch DDP through `accelerator.prepare` (lb_online_trainer_v7.py:72-76,153-154); its hooks
all-reduce(mean) the gradients inside `accelerator.backward` (:604) before `clip_grad_norm_` (:608).  Here every gradient already
lives in one flat fp32 arena, so the exchange is a sum all-reduce of arena slices (torch.distributed 'nccl' = RCCL over xGMI on a
node; 'gloo' in the CPU tests and when ranks share one GPU) issued asynchronously in gradient-ready order, and the 1/world
averaging is folded into the consumer (the fused optimiser's gradient scale).

The class is device-agnostic on purpose: tests/test_distributed_cpu.py drives exactly this code with CPU tensors under gloo, and
PolicyTrainer drives it with the HBM arena.  There is no fallback path: an unsupported backend raises on the first launch, and
a rank whose slice layout differs from rank 0's raises at construction (mismatched collectives would hang RCCL).
"""
import torch
import torch.distributed as dist


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
    def __init__(self, arena: torch.Tensor, slices, process_group=None, world_size=None, wire="fp32"):
        """arena: flat fp32 gradient buffer; slices: [(lo, hi), ...] element ranges in launch order (they must tile a prefix-free,
        non-overlapping part of the arena; empty ranges are skipped on every rank alike).
        wire: "fp32" (default: the reference's DDP exchanges fp32 gradients) or "bf16" -- every slice is rounded to a bf16 staging
        buffer, summed on the wire in bf16 and widened back into the arena: half the bytes per link (175 instead of 349 MB for the
        policy), at bf16 resolution of the SUM (2^-9 relative per element); an opt-in performance mode, never the parity path."""
        assert arena.dim() == 1 and arena.is_contiguous()
        if wire not in ("fp32", "bf16"):
            raise ValueError(wire)
        self.wire = wire
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
        # a single rank with a live communicator still issues its collectives (V2A_FORCE_DP: the RCCL path on a one-GPU box)
        self.active = dist.is_available() and dist.is_initialized()
        self._agree_on_layout()

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
