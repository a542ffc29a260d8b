"""Fused multi-tensor Adam over flat fp32 buffers (C ABI: dsee_adam_step_range) with torch.optim.Adam semantics
(sr_model.py:488-493: betas (beta1, beta2), eps 1e-8, no weight decay) including "a parameter whose .grad is
None is skipped" per tensor, and the data-parallel gradient exchange overlapped with the update: the flat gradient
is all-reduced in chunks (deepsee_amd.parallel.GradAllReduce) and the Adam launch of chunk k runs while RCCL is
still reducing chunk k+1 (north_star).

The per-tensor descriptors (offset, numel, first block, step count, lr, active flag: dsee_adam_tensor in
include/deepsee_hip.h) live on the DEVICE for good: a step uploads only the `touched` flags (pinned, asynchronous),
optionally MAX-reduces them over the ranks, and bumps the step counters with a device-side add -- no blocking
host-to-device copy per optimizer step."""
import numpy as np
import torch

from . import lib as L

ADAM_DT = np.dtype([("offset", "<i8"), ("numel", "<i8"), ("first_block", "<i4"), ("step", "<i4"), ("lr", "<f4"),
                    ("active", "<i4")])
_W = ADAM_DT.itemsize // 4          # 32-bit words per descriptor
_STEP, _LR, _ACTIVE = 5, 6, 7       # word index of the dynamic fields
BLOCK = 1024                        # elements per Adam block (a block never straddles two tensors)


class FlatAdam:
    """Owns flat param / grad / exp_avg / exp_avg_sq buffers; every nn.Parameter handed in is re-pointed to a view of
    the flat param buffer.  Gradients: zero_grad() sets every .grad to None, so autograd's AccumulateGrad simply keeps
    the tensor a backward pass produced (no `grad += new` kernel per parameter); step() hands the addresses of those
    tensors to ONE gather kernel (dsee_grad_gather) that fills the flat gradient buffer -- zeros for parameters without a
    gradient -- and the per-tensor active flags, so the RCCL all-reduce and the Adam kernel see one contiguous tensor."""

    def __init__(self, groups, betas=(0.0, 0.9), eps=1e-8):
        # groups: list of dict(params=[(name, Parameter)], lr=float)
        self.betas, self.eps = betas, eps
        self.param_groups = []
        self.names, self.params, self.group_of = [], [], []
        for gi, g in enumerate(groups):
            self.param_groups.append({"lr": float(g["lr"]), "params": [p for _, p in g["params"]]})
            for name, p in g["params"]:
                self.names.append(name)
                self.params.append(p)
                self.group_of.append(gi)
        dev = self.params[0].device
        # header: the first `hdr` floats of the flat buffers carry the per-tensor "has a gradient" flags (as 0 / 1 floats,
        # written by the gather kernel), so that under data-parallel training they travel INSIDE the first all-reduced
        # gradient chunk (sum > 0 <=> some rank has a gradient) instead of in a blocking collective of their own
        self.hdr = (len(self.params) + BLOCK - 1) // BLOCK * BLOCK
        offs, total = [], self.hdr
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every tensor 16 B aligned
        self.total = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        desc = np.zeros(len(self.params), dtype=ADAM_DT)
        blocks, block_start, fb = [], [], 0
        for i, (p, o) in enumerate(zip(self.params, offs)):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = None
            nb = (n + BLOCK - 1) // BLOCK
            desc[i] = (o, n, fb, 0, self.param_groups[self.group_of[i]]["lr"], 1)
            blocks += [i] * nb
            block_start += [o + j * BLOCK for j in range(nb)]
            fb += nb
        self.nblocks = fb
        self.block_start = block_start                      # first element of every block (host copy, for the chunking)
        self.block_tensor = torch.tensor(blocks, dtype=torch.int32, device=dev)
        self.desc_dev = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        self._d32 = self.desc_dev.view(torch.int32).view(len(self.params), _W)
        self._dlr = self.desc_dev.view(torch.float32).view(len(self.params), _W)[:, _LR]
        self._lr_sent = [self.param_groups[g]["lr"] for g in self.group_of]
        self.offsets = offs
        self.reduce_hook = None   # parallel.GradAllReduce, installed by parallel.attach
        # "p.grad is None -> skipped" (torch.optim.Adam after zero_grad(set_to_none=True)): a tensor is active in
        # a step iff autograd delivered a gradient for it since the last zero_grad().  `touched`: host copy of that.
        self.touched = np.zeros(len(self.params), dtype=np.int32)
        # ring of pinned staging buffers for the asynchronous upload of the gradient addresses (the host runs ahead of
        # the GPU: a buffer is reused only after the copy that read it has executed)
        self._ring, self._ring_pos = [], 0
        for _ in range(4 if dev.type == "cuda" else 1):
            t = torch.zeros(len(self.params), dtype=torch.int64)
            self._ring.append([t.pin_memory() if dev.type == "cuda" else t, None])
        self._ptr_dev = torch.zeros(len(self.params), dtype=torch.int64, device=dev)
        self._active_dev = torch.zeros(len(self.params), dtype=torch.int32, device=dev)
        self._ranges = {}
        self._held = None

    def zero_grad(self, set_to_none=True):
        """set_to_none=True (default, torch's default): every .grad becomes None and the next backward pass simply keeps the
        tensors it produces.  set_to_none=False (torch's other mode): existing .grad tensors are zero-filled in place and
        kept -- autograd then accumulates into them (one add kernel per parameter) and, as in torch, Adam treats every such
        tensor as "has a gradient"."""
        for p in self.params:
            if set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.detach_()
                p.grad.zero_()

    def grad_view(self, name):
        """The gradient of parameter `name` as step() consumed it: a view of the flat gradient buffer -- after the step of a
        data-parallel run the SUM over the ranks (the Adam kernel applies 1/world and the value clip on the fly; neither is
        written back).  `p.grad` itself is NOT this view: it stays the local tensor the backward pass produced (un-reduced,
        un-clipped; the reference's clip_grad_value_ rewrites p.grad in place, trainer_manager.py:39-40), so code that
        inspects gradients after a step should read them here."""
        i = self.names.index(name)
        o, p = self.offsets[i], self.params[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def staging(self):
        """A pinned staging tensor for step(pinned=...) (hipGraph capture: allocate BEFORE the capture starts)."""
        t = torch.zeros(len(self.params), dtype=torch.int64)
        return t.pin_memory() if self.flat.is_cuda else t

    def _gather(self):
        L.call("grad_gather", self._ptr_dev, self.desc_dev, self.block_tensor, self.nblocks, self.grad, self._active_dev)

    def chunk_ranges(self, chunk_elems):
        """[(first_block, end_block, lo, hi)]: block ranges of about chunk_elems elements whose element spans [lo, hi)
        tile the flat buffer exactly (tensor padding included), so 'all-reduce [lo,hi) then Adam on its blocks' covers
        every gradient element once."""
        if chunk_elems not in self._ranges:
            out, b0 = [], 0
            while b0 < self.nblocks:
                b1 = b0 + 1
                while b1 < self.nblocks and self.block_start[b1] - self.block_start[b0] < chunk_elems:
                    b1 += 1
                lo = 0 if b0 == 0 else self.block_start[b0]
                hi = self.total if b1 == self.nblocks else self.block_start[b1]
                out.append((b0, b1, lo, hi))
                b0 = b1
            self._ranges[chunk_elems] = out
        return self._ranges[chunk_elems]

    def _launch(self, b0, b1, grad_scale, clip):
        L.call("adam_step_range", self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.desc_dev, self.block_tensor,
               b0, b1 - b0, float(self.betas[0]), float(self.betas[1]), float(self.eps), float(grad_scale), float(clip))

    def sync_lr(self):
        """Upload the learning rates if a param_group's lr changed (TrainerManager.update_learning_rate).  A blocking
        host-to-device copy: never inside a hipGraph capture (managers call it before capturing / replaying)."""
        lrs = [self.param_groups[g]["lr"] for g in self.group_of]
        if lrs != self._lr_sent:
            self._dlr.copy_(torch.tensor(lrs, dtype=torch.float32))
            self._lr_sent = lrs

    def step(self, clip=-1.0, pinned=None):
        """`pinned`: a pinned int64 staging tensor (FlatAdam.staging()) owned by the caller -- required while a hipGraph is being captured (the
        captured host-to-device copy reads it on every replay, so it must hold this step's `touched` flags for good)."""
        hook = self.reduce_hook
        multi = hook is not None and hook.active
        capturing = self.flat.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.sync_lr()
        # active flags: this rank's `touched`, MAX over the ranks (a rank whose encoder coin differed would otherwise
        # update a different parameter subset with a different step count -- parameters would silently diverge)
        grads = []
        for p in self.params:
            g = p.grad
            if g is not None and not (g.is_contiguous() and g.dtype == torch.float32):
                g = g.contiguous().float()
            grads.append(g)
        ptrs = np.array([0 if g is None else g.data_ptr() for g in grads], dtype=np.int64)
        self.touched = (ptrs != 0).astype(np.int32)
        if capturing:
            # (data parallel: the chunked all-reduce below is captured too -- RCCL collectives are capturable; the graph then
            # holds forward + backward + gather + every chunk's all-reduce / Adam pair and a rank's host does ONE launch)
            assert pinned is not None, "graph capture: caller-owned staging buffer"
            pinned.copy_(torch.from_numpy(ptrs))
            self._ptr_dev.copy_(pinned, non_blocking=True)
        else:
            slot = self._ring[self._ring_pos]
            self._ring_pos = (self._ring_pos + 1) % len(self._ring)
            if slot[1] is not None:
                slot[1].synchronize()
            slot[0].copy_(torch.from_numpy(ptrs))
            self._ptr_dev.copy_(slot[0], non_blocking=True)
            if self._ptr_dev.is_cuda:
                slot[1] = torch.cuda.Event()
                slot[1].record()
        self._gather()           # flat gradient + active flags in one launch
        self._held = grads       # (the gradient tensors stay referenced until the next step has been enqueued)
        if multi:
            ranges = self.chunk_ranges(hook.chunk_elems)
            works = hook.start(self.grad, [(lo, hi) for _, _, lo, hi in ranges])
            for i, ((b0, b1, _, _), w) in enumerate(zip(ranges, works)):
                w.wait()      # the compute stream waits for THIS chunk only; RCCL keeps reducing the later ones
                if i == 0:
                    # the flags rode in the header of chunk 0: a tensor is active iff ANY rank has a gradient for it (a
                    # rank whose encoder coin differed would otherwise update a different parameter subset with a
                    # different step count -- parameters would silently diverge)
                    self._d32[:, _ACTIVE].copy_(self.grad[:len(self.params)] > 0.5)
                self._launch(b0, b1, hook.scale, clip)
        else:
            self._d32[:, _ACTIVE].copy_(self._active_dev)
            self._launch(0, self.nblocks, 1.0, clip)
        self._d32[:, _STEP] += self._d32[:, _ACTIVE]

    def steps(self):
        """Per-tensor update counts (torch's state['step']), read back from the device."""
        return self._d32[:, _STEP].cpu().numpy().copy()

    def state_dict(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.steps()}
