"""Data-parallel training: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) sum all-reduce of
the flat G / D gradient buffers over xGMI, chunked so the fused Adam of chunk k overlaps the all-reduce of
chunk k+1 (north_star).

Replaces torch.nn.DataParallel + DataParallelWithCallback (sync_batchnorm/replicate.py:50-94,
base_manager.py:15-23): no per-iteration parameter broadcast, no scatter/gather, BatchNorm statistics are
shard-local (sync-free, identical to the reference's single-device branch on each shard), and the gradient
exchange is one collective per optimizer step instead of a reduce-to-GPU0.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


def chunk_bounds(total, chunk_elems):
    """[(lo, hi)] covering [0, total) in pieces of about chunk_elems (16-byte aligned)."""
    chunk_elems = max(4, (int(chunk_elems) + 3) // 4 * 4)
    return [(lo, min(total, lo + chunk_elems)) for lo in range(0, total, chunk_elems)]


class GradAllReduce:
    """Callable installed as FlatAdam.reduce_hook: launches the chunked async all-reduces on RCCL's stream and makes
    the compute stream wait for them in order; returns the 1/world scale the Adam kernel folds into the gradient
    (== reference DP's mean of per-replica mean losses, trainer_manager.py:36)."""

    def __init__(self, world, chunk_mb=24.0, group=None):
        self.world, self.group = world, group
        self.chunk_elems = int(chunk_mb * (1 << 20) / 4)

    def __call__(self, flat_grad):
        if self.world <= 1:
            return 1.0
        works = [dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for lo, hi in chunk_bounds(flat_grad.numel(), self.chunk_elems)]
        for w in works:
            w.wait()   # stream-level wait on NCCL/RCCL; host-level on gloo
        return 1.0 / self.world


def attach(trainer, world, chunk_mb=24.0):
    """Hook a TrainerManager's optimizers up for data-parallel training and check rank consistency of the init."""
    hook = GradAllReduce(world, chunk_mb)
    trainer.optimizer_G.reduce_hook = hook
    trainer.optimizer_D.reduce_hook = hook
    if world > 1:
        for opt in (trainer.optimizer_G, trainer.optimizer_D):
            dist.broadcast(opt.flat, src=0)   # identical start on every rank (also true by seed)
        for net in (trainer.sr_model.netSR, trainer.sr_model.netD, trainer.sr_model.netE):
            if net is not None:
                for b in net.buffers():
                    if b.is_floating_point():
                        dist.broadcast(b, src=0)
    return trainer
