"""Data-parallel training: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) sum all-reduce of
the flat G / D gradient buffers over xGMI, issued as chunks on RCCL's stream; the fused Adam launch of chunk k runs
on the compute stream while RCCL is still reducing chunk k+1 (deepsee_amd.optim.FlatAdam.step; north_star,
SURVEY 2.3, 8e).

Replaces torch.nn.DataParallel + DataParallelWithCallback (sync_batchnorm/replicate.py:50-94,
base_manager.py:15-23): no per-iteration parameter broadcast, no scatter/gather, BatchNorm statistics are
shard-local by default (sync-free, identical to the reference's single-device branch on each shard; `opt.sync_bn`
switches to statistics over the global batch, the reference's DP branch, sync_batchnorm/batchnorm.py:70-145), and the
gradient exchange is one chunked collective per optimizer step instead of a reduce-to-GPU0.

Rank consistency: parameters and spectral-norm buffers start identical (broadcast from rank 0) and stay
identical because every rank applies the same all-reduced gradient; the encoder-branch coins come from
networks.DeviceNoise's own rank-independent coin function; the per-tensor "has a gradient" flags ride in the header of the
first all-reduced gradient chunk (sum > 0), so no rank can update a different parameter subset and no collective of their
own blocks the step; the device noise seed is offset per rank; every rank pins itself to its own host cores.
"""
import os

import torch
import torch.distributed as dist

NOISE_SEED_STRIDE = 1000003   # per-rank offset of the Philox noise seed (different noise per shard)


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # bring-up on a box with fewer GPUs than ranks: DSEE_DIST_BACKEND=gloo DSEE_ONE_DEVICE=1 runs every rank on cuda:0
    # with gloo collectives (RCCL refuses two ranks on one device) -- exercises launcher, hooks and bench plumbing only
    backend = backend or os.environ.get("DSEE_DIST_BACKEND")
    if os.environ.get("DSEE_ONE_DEVICE") == "1":
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)          # (whatever the backend: a gloo bring-up run must not pile onto cuda:0)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    pin_rank_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    return rank, local, world


def rank_core_set(local_rank, local_world, cores=None):
    """The host cores of one rank: the process's allowed cores split into `local_world` contiguous sets."""
    cores = sorted(os.sched_getaffinity(0)) if cores is None else sorted(cores)
    per = max(1, len(cores) // max(1, local_world))
    mine = cores[local_rank * per:(local_rank + 1) * per]
    return mine or cores


def pin_rank_cores(local_rank, local_world):
    """8 ranks x ~1 500 Python-issued launches per half step on one host: without pinning the launch threads migrate and
    contend for the same cores.  Every rank takes its own contiguous core set (no-op for a single rank)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    mine = rank_core_set(local_rank, local_world)
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(8, len(mine))))
    except OSError:
        return None
    return mine


def chunk_bounds(total, chunk_elems):
    """[(lo, hi)] covering [0, total) in pieces of about chunk_elems (16-byte aligned)."""
    chunk_elems = max(4, (int(chunk_elems) + 3) // 4 * 4)
    return [(lo, min(total, lo + chunk_elems)) for lo in range(0, total, chunk_elems)]


class CapiComm:
    """RCCL through the C ABI (include/deepsee_hip.h: dsee_comm_*) instead of torch.distributed's ProcessGroupNCCL -- what a
    host that binds the library directly would use (opt.dp_comm = "capi").  Collectives are enqueued on the CURRENT torch
    stream; `GradAllReduce` gives them a side stream of their own."""

    def __init__(self, world, rank, id_bytes):
        import ctypes as C
        from . import lib as L
        assert len(id_bytes) == self.id_bytes()
        self.world, self.rank = int(world), int(rank)
        handle = C.c_void_p()
        rc = L.lib().dsee_comm_init(C.byref(handle), bytes(id_bytes), self.world, self.rank)
        if rc != 0:
            raise L.DseeError("dsee_comm_init failed (%d): %s" % (rc, L.lib().dsee_last_error().decode()))
        self.handle = handle
        self.stream = torch.cuda.Stream()

    @staticmethod
    def id_bytes():
        return 128      # DSEE_COMM_ID_BYTES

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import lib as L
        buf = C.create_string_buffer(CapiComm.id_bytes())
        rc = L.lib().dsee_comm_unique_id(buf)
        if rc != 0:
            raise L.DseeError("dsee_comm_unique_id failed (%d): %s" % (rc, L.lib().dsee_last_error().decode()))
        return buf.raw

    @classmethod
    def bootstrap(cls, world, rank):
        """Rank 0 draws the id; the 128 bytes travel over whatever the launcher already set up (torch.distributed's store
        here; a host without torch ships them over its own channel)."""
        ident = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        return cls(world, rank, ident[0])

    def all_reduce_sum_(self, t):
        from . import lib as L
        assert t.dtype == torch.float32 and t.is_contiguous()
        L.call("comm_allreduce_sum", self.handle, t, t.numel())
        return t

    def broadcast_(self, t, root=0):
        from . import lib as L
        assert t.is_contiguous()
        L.call("comm_broadcast", self.handle, t, t.numel() * t.element_size(), root)
        return t

    def all_gather(self, local):
        from . import lib as L
        local = local.contiguous()
        out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        L.call("comm_allgather", self.handle, local, out, local.numel() * local.element_size())
        return out

    def close(self):
        from . import lib as L
        if self.handle is not None:
            torch.cuda.synchronize()
            handle, self.handle = self.handle, None
            rc = L.lib().dsee_comm_destroy(handle)
            if rc != 0:
                raise L.DseeError("dsee_comm_destroy failed (%d): %s" % (rc, L.lib().dsee_last_error().decode()))


class _StreamWork:
    """The handle GradAllReduce returns for a chunk reduced on CapiComm's side stream: wait() orders the current stream after
    it (the stream-level wait torch's NCCL work objects do)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class GradAllReduce:
    """Installed as FlatAdam.reduce_hook.  `start` launches the asynchronous sum all-reduce of every chunk on RCCL's
    stream and returns the work handles; FlatAdam waits for chunk k (stream-level on NCCL/RCCL, host-level on gloo),
    launches the Adam blocks of chunk k and moves on, so the update overlaps the remaining all-reduces.  `scale` is
    the 1/world the Adam kernel folds into the gradient (== reference DP's mean of per-replica mean losses,
    trainer_manager.py:36)."""

    def __init__(self, world, chunk_mb=24.0, group=None, force=False, comm=None):
        self.world, self.group, self.comm = int(world), group, comm
        self.chunk_elems = max(4, int(chunk_mb * (1 << 20) / 4))
        self.scale = 1.0 / max(1, self.world)
        # `force`: run the collectives even in a 1-rank process group (exercises the RCCL path on a single GPU)
        self.active = self.world > 1 or (force and (comm is not None or (dist.is_available() and dist.is_initialized())))

    def start(self, flat_grad, bounds):
        if not self.active:
            return [None] * len(bounds)
        if self.comm is not None:
            # dsee_comm_allreduce_sum per chunk on the communicator's own stream, behind everything the compute stream has
            # enqueued so far (the gather launch); one event per chunk for the Adam launch that consumes it
            side, works = self.comm.stream, []
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for lo, hi in bounds:
                    self.comm.all_reduce_sum_(flat_grad[lo:hi])
                    ev = torch.cuda.Event()
                    ev.record(side)
                    works.append(_StreamWork(ev))
            flat_grad.record_stream(side)
            return works
        return [dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                for lo, hi in bounds]

    def __call__(self, flat_grad):
        """Blocking form (whole buffer in chunks): returns the gradient scale."""
        for w in self.start(flat_grad, chunk_bounds(flat_grad.numel(), self.chunk_elems)):
            if w is not None:
                w.wait()
        return self.scale


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def gather_stats(local, world, group=None):
    """SyncBN forward exchange: every rank contributes its shard's (mean, M2) rows [2, C]; returns [world, 2, C] in rank
    order (the merge kernel folds them with Chan's update in that fixed order, so every rank computes bit-identical
    statistics).  Replaces the master's ReduceAddCoalesced + Broadcast of (sum, ssum) through Python queue pipes
    (sync_batchnorm/batchnorm.py:105-126, comm.py:46-133): 8*C bytes per rank per BN layer."""
    if isinstance(group, CapiComm):
        return group.all_gather(local)
    if world <= 1 and not _dist_on():
        return local.unsqueeze(0)
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1), group=group)
    return out


def allreduce_sums(sums, world, group=None):
    """SyncBN backward exchange: the per-channel sums (sum d, sum d*xhat) of the BN backward over the GLOBAL batch
    (the reference gets them through autograd of its ReduceAddCoalesced / Broadcast nodes)."""
    if isinstance(group, CapiComm):
        if not sums.is_contiguous():
            tmp = group.all_reduce_sum_(sums.contiguous())
            sums.copy_(tmp)
        else:
            group.all_reduce_sum_(sums)
    elif world > 1 or _dist_on():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def attach(trainer, world, chunk_mb=24.0, rank=None, force=False):
    """Hook a TrainerManager's optimizers up for data-parallel training, make the start state identical on every
    rank, give every rank its own noise stream, switch SyncBN on if opt.sync_bn.  `force` keeps the collectives in the
    path even with one rank (a 1-rank NCCL group: the -m gpu test that runs this code on the MI355X)."""
    from . import ops
    if rank is None:
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    # opt.dp_comm = "capi": RCCL through include/deepsee_hip.h's dsee_comm_* (torch.distributed only carries the 128-byte id)
    comm = None
    if getattr(trainer.opt, "dp_comm", "torch") == "capi" and (world > 1 or force):
        comm = CapiComm.bootstrap(world, rank)
    elif getattr(trainer.opt, "dp_comm", "torch") not in ("torch", "capi"):
        raise ValueError("opt.dp_comm must be 'torch' or 'capi'")
    trainer.dp_comm = comm
    hook = GradAllReduce(world, chunk_mb, force=force, comm=comm)
    trainer.optimizer_G.reduce_hook = hook
    if trainer.optimizer_D is not None:
        trainer.optimizer_D.reduce_hook = hook
    model = trainer.sr_model_on_one_gpu
    model.noise.seed += NOISE_SEED_STRIDE * rank
    model.dp_world, model.dp_rank = int(world), int(rank)
    opt = trainer.opt
    cfg = (ops.SyncBNConfig(world, comm, getattr(opt, "sync_bn_clamp", True))
           if getattr(opt, "sync_bn", False) and (world > 1 or force) else None)
    model.plan = model.plan.replace(sync_bn=cfg)      # (per model: deepsee_amd/plan.py)
    if world > 1 or comm is not None:
        bcast = (lambda t: comm.broadcast_(t, 0)) if comm is not None else (lambda t: dist.broadcast(t, src=0))
        for opt in (trainer.optimizer_G, trainer.optimizer_D):
            if opt is not None:
                bcast(opt.flat)   # identical start on every rank (also true by seed)
        for net in (model.netSR, model.netD, model.netE):
            if net is not None:
                for b in net.buffers():
                    if b.is_floating_point():
                        bcast(b)
    return trainer
