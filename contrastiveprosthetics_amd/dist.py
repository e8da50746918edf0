"""One-process-per-GPU data parallelism over torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" for the CPU plumbing tests).  The reference has no distributed code at all; this is the
multi-GPU design of SURVEY.md 8e:

* groups (the batch axis) are sharded over ranks -- the 41x41 contrastive blocks are independent;
* the 8 MB flat gradient buffer is summed with ONE all-reduce per step and averaged inside the fused
  Adam kernel (grad_scale = 1/world); the L2 regulariser is applied after the reduction, once;
* the z embeddings (N_local x 16 f32) can be all-gathered into the global-batch matrix; under the
  reference's per-group loss every rank then scores its own row slice (parity-neutral);
* BatchNorm statistics stay local to a rank (standard DDP semantics == the reference at B_local).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as td


def world_size() -> int:
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def rank() -> int:
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


_default_tiles = None        # None: $CPNATIVE_TILE_SCHEDULE or static


def default_tile_schedule() -> str:
    """The tile schedule an Engine built from now on starts with ("static" | "dynamic"; cp_config.tile_schedule,
    include/cpnative.h): what share_gpu_with_other_kernels() chose, else $CPNATIVE_TILE_SCHEDULE, else static.  Read by
    Engine.__init__ -- the library itself reads no environment variable."""
    if _default_tiles is not None:
        return _default_tiles
    return "dynamic" if os.environ.get("CPNATIVE_TILE_SCHEDULE") == "dynamic" else "static"


def share_gpu_with_other_kernels():
    """The engines this process builds will run their GEMM launches beside other processes' kernels on the same GPU (a
    packed sweep): they draw their tiles dynamically (Engine.tile_schedule -> cp_config.tile_schedule) unless
    $CPNATIVE_TILE_SCHEDULE says otherwise.  Data-parallel runs keep the static schedule: the collectives are placed so
    that at most one persistent GEMM launch per step (the projection's data gradient) can meet RCCL's kernels -- the z
    all-gather runs beside the head and the projection's backward, the large gradient bucket beside the conv backward --
    and a late-starting launch costs less there than the dynamic schedule's 2 % on all fifteen."""
    global _default_tiles
    if os.environ.get("CPNATIVE_TILE_SCHEDULE"):
        return
    _default_tiles = "dynamic"


def init_from_env(backend: str = None) -> Tuple[int, int]:
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  No-op for 1 process."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    if w <= 1 or td.is_initialized():
        return rank(), world_size()
    if backend is None:
        # CP_DIST_BACKEND=gloo: rehearsal of the multi-rank training flow on a box with fewer GPUs than ranks
        # (ranks share devices, collectives go through gloo); production is nccl == RCCL, one GPU per rank
        backend = os.environ.get("CP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        td.init_process_group(backend)
    return rank(), world_size()


def shutdown():
    if td.is_available() and td.is_initialized():
        td.destroy_process_group()


def shard_range(n_items: int, r: int, w: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, end) of n_items for rank r of w (first n % w ranks get one more)."""
    base, extra = divmod(n_items, w)
    start = r * base + min(r, extra)
    return start, start + base + (1 if r < extra else 0)


def all_reduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        td.all_reduce(flat, op=td.ReduceOp.SUM)
    return flat


class GradAllReduce:
    """Sum of the flat gradient buffer over the ranks in two buckets, the big one beside the backward pass.

    The backward call records an event once every gradient but the conv stack's is final (cp_encoder_backward_ev: the
    flat buffer from the first fc weight on, 7.9 of 8.1 MB); that part is summed on a side stream behind the event,
    while conv2's and conv1's backward kernels (~0.5 ms at 167,936 windows) still run.  The 0.15 MB in front of it
    follows when the backward is complete.  Element-wise the result is that of one all-reduce of the whole buffer.
    Whatever else writes into the large bucket (the glove-angle class encoder's backward) has to be enqueued BEFORE
    engine.encoder_backward.
    Collectives are issued in the same order on every rank."""

    def __init__(self, engine, force: bool = False):
        first_fc = "emg_net.linear.0.weight"
        self.split = engine.grads.offsets[first_fc][0]
        assert all(k.startswith("emg_net.conv_emg.") for k, (o, _) in engine.grads.offsets.items() if o < self.split)
        self.flat = engine.grads.flat
        self.active = (world_size() > 1 or force) and self.flat.is_cuda
        if self.active:
            self.side = torch.cuda.Stream(device=self.flat.device)
            self.event = torch.cuda.Event()
            self.event.record()                      # creates the HIP event whose handle the C call records
            engine.fc_grads_ready = self.event

    def __call__(self):
        """Call after engine.encoder_backward (everything is enqueued by then); returns when both sums are ordered
        before whatever the current stream runs next."""
        if not self.active:                          # CPU tensors (gloo tests): the same two buckets, no streams
            if world_size() > 1:
                td.all_reduce(self.flat[self.split:], op=td.ReduceOp.SUM)
                td.all_reduce(self.flat[:self.split], op=td.ReduceOp.SUM)
            return self.flat
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.event)
            big = td.all_reduce(self.flat[self.split:], op=td.ReduceOp.SUM, async_op=True)
        small = td.all_reduce(self.flat[:self.split], op=td.ReduceOp.SUM, async_op=True)
        big.wait()
        small.wait()
        return self.flat


def barrier():
    """All ranks meet here (no-op for one process): e.g. between rank 0 writing a checkpoint and everybody reading it."""
    if world_size() > 1:
        td.barrier()


def broadcast_buffers_(tensors, src: int = 0):
    """Rank src's BatchNorm running statistics to every rank before an evaluation, as DDP's broadcast_buffers does
    (training keeps them rank-local: each rank updates them from its own shard)."""
    if world_size() > 1:
        for t in tensors:
            td.broadcast(t, src)


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    if world_size() > 1:
        td.broadcast(flat, src)
    return flat


def all_gather_rows(local: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """(n_local, d) per rank -> (world*n_local, d), rank-major.  Equal n_local on every rank."""
    w = world_size()
    if w == 1:
        return local
    if out is None:
        out = torch.empty((w * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    td.all_gather_into_tensor(out, local.contiguous())
    return out


def local_rows(gathered: torch.Tensor, n_local: int) -> torch.Tensor:
    r = rank()
    return gathered[r * n_local:(r + 1) * n_local]


# ---- hyper-parameter sweep packing ----------------------------------------------
# The random search of code/train.py:140-166,175-194 trains `--crossval_size` independent models one after the
# other.  Packed, rank r of w trains configurations r, r+w, r+2w, ... on its own (no gradient exchange), with any
# number of ranks per GPU: at the reference's batch sizes a training step is launch-latency-bound, so several
# processes share one MI355X without slowing each other much.  Results travel as Python objects over gloo.
def init_packed_from_env():
    """Join RANK / WORLD_SIZE over gloo and return (rank, world, device index = LOCAL_RANK mod visible GPUs)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    dev = local % n_dev if n_dev else 0
    if w > 1 and not td.is_initialized():
        td.init_process_group("gloo")
    if w > 1:
        share_gpu_with_other_kernels()
    return rank(), world_size(), dev


def packed_indices(n_items: int, r: int, w: int):
    """Round-robin share of rank r: r, r+w, ... (keeps every rank busy when later configurations are the slow ones)."""
    return list(range(r, n_items, w))


def gather_packed(local: dict, n_items: int):
    """local: {item index: result} of this rank -> list of n_items results on every rank, in item order."""
    if world_size() == 1:
        parts = [local]
    else:
        parts = [None] * world_size()
        td.all_gather_object(parts, local)
    merged = {}
    for part in parts:
        for k, v in part.items():
            if k in merged:
                raise RuntimeError(f"configuration {k} was trained by two ranks")
            merged[k] = v
    missing = [i for i in range(n_items) if i not in merged]
    if missing:
        raise RuntimeError(f"configurations {missing} were trained by no rank")
    return [merged[i] for i in range(n_items)]
