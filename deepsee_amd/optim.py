"""Fused multi-tensor Adam over flat fp32 buffers (C ABI: dsee_adam_step) with torch.optim.Adam semantics
(sr_model.py:488-493: betas (beta1, beta2), eps 1e-8, no weight decay) including "a parameter whose .grad is
None is skipped" per tensor, and the hook for the data-parallel gradient all-reduce."""
import ctypes as C

import numpy as np
import torch

from . import lib as L

ADAM_DT = np.dtype([("offset", "<i8"), ("numel", "<i8"), ("first_block", "<i4"), ("step", "<i4"), ("lr", "<f4"),
                    ("active", "<i4")])


class FlatAdam:
    """Owns flat param / grad / exp_avg / exp_avg_sq buffers; every nn.Parameter handed in is re-pointed to a view
    of the flat param buffer and gets a persistent .grad view of the flat grad buffer (autograd accumulates in
    place), so the RCCL all-reduce and the Adam kernel see one contiguous tensor each."""

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
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every tensor 16 B aligned
        self.total = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.desc = np.zeros(len(self.params), dtype=ADAM_DT)
        blocks, fb = [], 0
        for i, (p, o) in enumerate(zip(self.params, offs)):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            nb = (n + 1023) // 1024
            self.desc[i] = (o, n, fb, 0, 0.0, 1)
            blocks += [i] * nb
            fb += nb
        self.nblocks = fb
        self.block_tensor = torch.tensor(blocks, dtype=torch.int32, device=dev)
        self.desc_dev = torch.zeros(self.desc.nbytes, dtype=torch.uint8, device=dev)
        self.offsets = offs
        self.reduce_hook = None   # callable(flat_grad) -> grad_scale, installed by parallel.DataParallel
        # "p.grad is None -> skipped" (torch.optim.Adam after zero_grad(set_to_none=True)): a tensor is active in
        # a step iff autograd delivered a gradient for it since the last zero_grad().
        self.touched = np.zeros(len(self.params), dtype=np.int32)
        for i, p in enumerate(self.params):
            p.register_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(grad):
            self.touched[i] = 1
            return None
        return hook

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()
        self.touched[:] = 0
        for p, o in zip(self.params, self.offsets):  # restore views if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def step(self, clip=-1.0):
        grad_scale = 1.0
        if self.reduce_hook is not None:
            grad_scale = self.reduce_hook(self.grad)
        self.desc["active"] = self.touched
        for i in range(len(self.names)):
            self.desc["lr"][i] = self.param_groups[self.group_of[i]]["lr"]
        host = torch.from_numpy(self.desc.view(np.uint8).copy())
        self.desc_dev.copy_(host, non_blocking=False)
        L.call("adam_step", self.flat, self.grad, self.exp_avg, self.exp_avg_sq, C.c_void_p(self.desc_dev.data_ptr()),
               C.c_void_p(self.block_tensor.data_ptr()), self.nblocks, float(self.betas[0]), float(self.betas[1]),
               float(self.eps), float(grad_scale), float(clip))
        self.desc["step"] += self.desc["active"]

    def state_dict(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.desc["step"].copy()}
