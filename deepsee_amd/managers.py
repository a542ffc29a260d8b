"""BaseManager / TrainerManager — same method surface as the reference's managers/base_manager.py and
managers/trainer_manager.py, driving the HIP-backed SRModel."""
import torch

from . import ops
from .sr_model import SRModel


class BaseManager:
    def __init__(self, opt, create_model=True):
        self.opt = opt
        if create_model:
            self.create_model(opt)

    def create_model(self, opt):
        """base_manager.py:15-23.  The reference wraps SRModel in DataParallelWithCallback (one process, one thread
        per GPU); here parallelism is one process per GPU + RCCL gradient all-reduce (deepsee_amd.parallel), so the
        wrapped and the unwrapped model are the same object."""
        self.sr_model = SRModel(opt)
        self.sr_model_on_one_gpu = self.sr_model

    def use_gpu(self):
        return True

    def preprocess(self, data, from_dataloader=False):
        """base_manager.py:28-66 + data/preprocessor.py: label -> integer map (kept as uint8 instead of a one-hot
        tensor), HR -> LR bicubic + clamp, everything NHWC on the device."""
        if isinstance(data.get("input_semantics"), ops.Labels):
            return data        # already native: a batch of deepsee_amd.data.DeviceLoader
        if from_dataloader and data["image"].dtype == torch.uint8:
            from .data import device_preprocess      # uint8 wire format: [N,H,W] labels, [N,H,W,3] images
            return device_preprocess(self.opt, data)
        out = dict(data)
        for k, v in data.items():
            if isinstance(v, torch.Tensor) and not v.is_cuda:
                out[k] = v.cuda(non_blocking=True)
        if not from_dataloader:
            return out
        opt = self.opt
        image = ops.to_nhwc(out["image"])
        res = {
            "input_semantics": ops.Labels(ops.label_to_u8(out["label"]), opt.label_nc),
            "image_lr": ops.bicubic_down(image, opt.start_size),
            "image_hr": image,
        }
        if opt.guiding_style_image:
            res["guiding_image"] = ops.to_nhwc(out["guiding_image"])
            res["guiding_label"] = ops.Labels(ops.label_to_u8(out["guiding_label"]), opt.label_nc)
        return res


class TrainerManager(BaseManager):
    """trainer_manager.py:6-96.

    opt.hip_graphs (default ON for a training manager; bench.py --no-graphs / opt.hip_graphs = False enqueue every kernel from
    Python): a G step and a D step are each captured ONCE per encoder-branch variant as a hipGraph
    (zero_grad, preprocessing, forward, backward, clipping, fused Adam: ~1 500 kernel launches each) and replayed
    afterwards -- one host call instead of ~50 ms of Python launch enqueue per step.  What makes that legal: every
    shape is static; inputs are copied into static buffers; kernel arguments that change per step do not exist (the
    Philox streams are offset by a DEVICE-side epoch that a captured add advances, learning rates and Adam step counts
    live in device memory); the two branch coins of the independent variant are a pure function of the forward index, so
    the variant to replay is known before it runs.  The first occurrence of a variant runs eagerly (it also sizes every
    workspace), the second is captured, later ones replay.  All graphs share one memory pool: a replay invalidates the
    activations of the previous one, which is why losses / the generated image are copied out after every replay.
    Data-parallel runs replay everything up to the backward pass + gradient gather and issue the chunked all-reduce + per-chunk
    Adam eagerly behind it (~12 launches per half step).  opt.dp_graph_collectives captures those as well (the graph is then
    the whole step, bit-identical results) but is OFF by default: capturing RCCL operations aborts intermittently inside the
    HIP runtime on this stack (tests/test_gpu_model.py::test_dp_collectives_captured_inside_the_graph_world1)."""

    _live = None      # weak set of the managers of this process (close_all(): a host that builds many of them, a test suite)

    @classmethod
    def close_all(cls):
        """close() every live manager of this process: captured graphs and their memory pools, static input buffers and the
        C-ABI communicators are released now instead of whenever the garbage collector gets to them."""
        errors = []
        for tm in list(cls._live):
            try:
                tm.close()
            except Exception as e:           # (one manager's failing communicator must not keep the others' graphs alive)
                errors.append(e)
        if errors:
            raise errors[0]

    def __init__(self, opt):
        super().__init__(opt, create_model=True)
        assert opt.isTrain
        self.optimizer_G, self.optimizer_D = self.sr_model_on_one_gpu.create_optimizers(opt)
        self.old_lr = opt.lr
        self.generated = None
        self.logs = {}
        self.g_losses, self.d_losses = {}, {}
        self.use_graphs = bool(getattr(opt, "hip_graphs", True))
        self.dp_in_graph = bool(getattr(opt, "dp_graph_collectives", False))
        self._graphs, self._seen, self._static, self._pool = {}, {}, {}, None
        # batch-shape signatures that own graphs / static buffers, least recently used first; at most `max_graph_shapes` are
        # kept (a loader with many distinct shapes would otherwise grow device memory without bound)
        self._sig_lru = []
        self.max_graph_shapes = int(getattr(opt, "max_graph_shapes", 4))
        self.graph_stats = {"eager": 0, "captured": 0, "replayed": 0, "evicted_shapes": 0}
        self.dp_comm = None
        TrainerManager._live.add(self)       # (closed by the ONE module-level atexit hook below: close_all)

    def get_logs(self):
        return {**self.logs, **self.sr_model_on_one_gpu.get_logs()}

    def preprocess_input(self, data):
        return super().preprocess(data, from_dataloader=True)

    # ---- the two steps, eager form (trainer_manager.py:32-61)
    def _g_step(self, data, pinned=None, opt_step=True):
        self.optimizer_G.zero_grad()
        d = self.preprocess_input(data)
        g_losses, generated = self.sr_model(d, mode="generator")
        g_loss = sum(g_losses.values()).mean()
        g_loss.backward()
        if opt_step:
            self.optimizer_G.step(clip=self.opt.gradient_clip, pinned=pinned)
        # (detached: a kept loss would keep this iteration's autograd graph -- and its AccumulateGrad nodes, which are
        # bound to the stream they were created on -- alive into a later hipGraph capture on another stream)
        return {k: v.detach() for k, v in g_losses.items()}, generated.detach()

    def _d_step(self, data, pinned=None, opt_step=True):
        self.optimizer_D.zero_grad()
        d = self.preprocess_input(data)
        d_losses = self.sr_model(d, mode="discriminator")
        d_loss = sum(d_losses.values()).mean()
        d_loss.backward()
        if opt_step:
            self.optimizer_D.step(clip=self.opt.gradient_clip, pinned=pinned)
        return {k: v.detach() for k, v in d_losses.items()}, None

    def run_generator_one_step(self, data):
        if self.use_graphs:
            self.g_losses, self.generated = self._graphed("G", data)
        else:
            self.g_losses, self.generated = self._g_step(data)

    def run_discriminator_one_step(self, data):
        if self.use_graphs:
            self.d_losses, _ = self._graphed("D", data)
        else:
            self.d_losses, _ = self._d_step(data)

    # ---- hipGraph capture / replay
    @staticmethod
    def _shape_signature(data):
        """Shapes / dtypes of a batch: graphs, their static input buffers and the eager-first / capture-second counters are
        keyed by it, so a batch of another shape (a loader without drop_last: the last partial batch of an epoch) gets its
        own eager pass -- which sizes the workspaces for THAT shape outside any capture -- and its own graph, and never
        replays kernels that baked in the addresses or sizes of another shape."""
        sig = []
        for k in sorted(data):
            v = data[k]
            if isinstance(v, torch.Tensor):
                sig.append((k, tuple(v.shape), str(v.dtype)))
            elif isinstance(v, ops.Labels):
                sig.append((k, tuple(v.t.shape), "labels%d" % v.nc))
        return tuple(sig)

    def _static_inputs(self, skey, data):
        """Copy the batch into buffers whose addresses the captured kernels know (one set per step and batch shape)."""
        st = self._static.setdefault(skey, {})
        out = {}
        for k, v in data.items():
            if isinstance(v, torch.Tensor):
                v = v.cuda(non_blocking=True) if not v.is_cuda else v
                if k not in st:
                    st[k] = torch.empty_like(v)
                st[k].copy_(v)
                if hasattr(v, "dsee_layout"):
                    st[k].dsee_layout = v.dsee_layout
                out[k] = st[k]
            elif isinstance(v, ops.Labels):
                if k not in st:
                    st[k] = ops.Labels(torch.empty_like(v.t), v.nc)
                st[k].t.copy_(v.t)
                out[k] = st[k]
            else:
                out[k] = v
        return out

    def _graphed(self, which, data):
        model = self.sr_model_on_one_gpu
        optim = self.optimizer_G if which == "G" else self.optimizer_D
        step_fn = self._g_step if which == "G" else self._d_step
        hook = optim.reduce_hook
        multi = hook is not None and hook.active
        noise = model.noise
        if not hasattr(noise, "step"):            # (a replayed oracle tape: no graphs)
            return step_fn(data)
        if multi and model.plan.sync_bn is not None and not self.dp_in_graph:
            # opt.sync_bn in a multi-rank run: the SyncBN all-gather / all-reduce sit INSIDE the step function, i.e. they would be
            # captured into the hipGraph -- and capturing RCCL operations is what dp_graph_collectives is off for (intermittent
            # abort inside the HIP runtime on this stack).  Such a run stays eager.
            return step_fn(data)
        sig = self._shape_signature(data)
        self._touch_signature(sig)
        # (the plan is part of the key: a graph replays the kernels chosen under the plan it was captured with)
        key = (which,) + tuple(model.encoder_branch(False, step=noise.step + 1)) + (sig, model.plan)
        seen = self._seen.get(key, 0)
        self._seen[key] = seen + 1
        if seen == 0 or ops.PROFILE is not None:
            self.graph_stats["eager"] += 1
            return step_fn(data)                  # first occurrence: eager (sizes every workspace, sets kernel attributes)
        sd = self._static_inputs((which, sig), data)
        optim.sync_lr()
        rec = self._graphs.get(key)
        if rec is None:
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            pinned = optim.staging()
            state = (noise.step, noise.offset)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            ops.begin_capture()
            # thread-local capture mode: RCCL's watchdog thread (event queries, in a data-parallel run) must not
            # invalidate a capture in progress on this thread
            eager_opt = multi and not self.dp_in_graph       # all-reduce + Adam outside the graph
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                losses, generated = step_fn(sd, pinned=pinned, opt_step=not eager_opt)
            ops.begin_capture()                   # (pools created on the capture stream belong to the graph)
            noise.step, noise.offset = state      # the capture ran the Python side once; the replay below is the real step
            # tensors the forward leaves in model.logs live in the shared graph pool as well: copied out after every replay
            logged = {k: v for k, v in model.logs.items() if isinstance(v, torch.Tensor)} if which == "G" else {}
            rec = {"graph": g, "losses": losses, "generated": generated, "pinned": pinned, "eager_opt": eager_opt,
                   "grads": [p.grad for p in optim.params], "full": model.last_encoded_style_is_full,
                   "noisy": model.last_encoded_style_is_noisy,
                   "out_generated": None if generated is None else torch.empty_like(generated),
                   "logs": logged, "out_logs": {k: torch.empty_like(v) for k, v in logged.items()}}
            self._graphs[key] = rec
            self.graph_stats["captured"] += 1
        else:
            self.graph_stats["replayed"] += 1
        # host-side state a forward leaves behind
        noise.step += 1
        model.last_encoded_style_is_full, model.last_encoded_style_is_noisy = rec["full"], rec["noisy"]
        rec["graph"].replay()
        if rec["eager_opt"]:
            for p, g in zip(optim.params, rec["grads"]):   # the gradient tensors this graph writes (static addresses)
                p.grad = g
            optim.step(clip=self.opt.gradient_clip)
        # The loss scalars are handed out as FRESH tensors every step, like the reference's (a caller that accumulates or keeps
        # them across iterations must not see a later replay's values): one stacked copy + views, not one per term.
        # The generated image (and the logged tensors) stay in the per-graph buffers out_generated / out_logs, overwritten by the
        # next replay of the same graph key -- the aliasing contract INTEGRATION.md states; clone() what must outlive a step.
        keys = list(rec["losses"])
        fresh = torch.stack([rec["losses"][k].reshape(()) for k in keys]) if keys else None
        out_losses = {k: fresh[i].reshape(rec["losses"][k].shape) for i, k in enumerate(keys)}
        if rec["generated"] is not None:
            rec["out_generated"].copy_(rec["generated"])
        for k, v in rec["logs"].items():
            rec["out_logs"][k].copy_(v)
            if hasattr(v, "dsee_layout"):
                rec["out_logs"][k].dsee_layout = v.dsee_layout
            model.logs[k] = rec["out_logs"][k]
        return out_losses, rec["out_generated"]

    def _touch_signature(self, sig):
        """LRU bookkeeping of the batch shapes that own graphs: a shape beyond `max_graph_shapes` evicts the least recently
        used one with everything keyed by it (graphs, static input buffers, eager-first counters)."""
        if self._sig_lru and self._sig_lru[-1] == sig:
            return
        if sig in self._sig_lru:
            self._sig_lru.remove(sig)
        self._sig_lru.append(sig)
        while len(self._sig_lru) > max(1, self.max_graph_shapes):
            old = self._sig_lru.pop(0)
            dead = [k for k in self._graphs if old in k]
            if dead:
                torch.cuda.synchronize()       # (a replay of one of them may still be running)
            for k in dead:
                del self._graphs[k]
            for k in [k for k in self._seen if old in k]:
                del self._seen[k]
            for k in [k for k in self._static if old in k]:
                del self._static[k]
            self.graph_stats["evicted_shapes"] += 1

    def close(self):
        """Drop the captured graphs, then the data-parallel communicator of opt.dp_comm = "capi" (parallel.attach) -- in that
        order, and before the HIP runtime goes away at interpreter exit (registered with atexit; idempotent)."""
        try:
            self.release_graphs()
        finally:
            comm, self.dp_comm = getattr(self, "dp_comm", None), None
            if comm is not None:
                comm.close()

    def release_graphs(self):
        """Drop every captured graph (and its static buffers).  Call it BEFORE torch.distributed.destroy_process_group() in a
        data-parallel run: with opt.dp_graph_collectives the graphs hold captured RCCL operations, and destroying them after
        their communicator is gone can abort the process at exit."""
        if self._graphs:
            torch.cuda.synchronize()
        self._graphs.clear()
        self._seen.clear()
        self._static.clear()
        del self._sig_lru[:]
        import gc
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def get_latest_losses(self):
        return {**self.g_losses, **self.d_losses}

    def get_latest_generated(self):
        return self.generated

    def save(self, epoch):
        self.sr_model_on_one_gpu.save(epoch)

    def update_learning_rate(self, epoch):
        """trainer_manager.py:76-96: linear decay after opt.niter, TTUR split kept."""
        if epoch > self.opt.niter:
            new_lr = self.old_lr - self.opt.lr / self.opt.niter_decay
        else:
            new_lr = self.old_lr
        if new_lr != self.old_lr:
            if self.opt.no_TTUR:
                new_lr_g, new_lr_d = new_lr, new_lr
            else:
                new_lr_g, new_lr_d = new_lr / 2, new_lr * 2
            for g in self.optimizer_D.param_groups:
                g["lr"] = new_lr_d
            for g in self.optimizer_G.param_groups:
                g["lr"] = new_lr_g
            print("update learning rate: %f -> %f" % (self.old_lr, new_lr))
            self.old_lr = new_lr


import weakref as _weakref

TrainerManager._live = _weakref.WeakSet()


def _close_all_at_exit():
    try:
        TrainerManager.close_all()
    except Exception as e:      # (interpreter shutdown: report, do not raise into atexit's own traceback printer)
        import sys
        print("deepsee_amd: closing the managers at exit failed: %r" % (e,), file=sys.stderr)


import atexit as _atexit
_atexit.register(_close_all_at_exit)
