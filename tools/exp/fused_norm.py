"""Experiment driver: fused SPADE/SEAN forward (dsee_spade_fused_fwd) vs the round-2 path (GEMM + output transform)."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import deepsee_oracle as O
from deepsee_amd import ops, networks as Nw
from tools._plan import use_plan


def nhwc(x):
    return ops.to_nhwc(x.cuda())


def rel(a, b):
    return float((a - b).norm() / b.norm())


def run(kind, C, R, N, fused, grad=True, H=None, max_fm=256):
    use_plan(fused_norm=fused)
    g = torch.Generator().manual_seed(11 + C + R)
    Lc, S = 19, 128
    H = H or max(R, 64)
    label = F.interpolate(torch.randint(0, Lc, (N, 1, H // 8, H // 8), generator=g).float(), size=(H, H), mode="nearest")
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1)
    x = torch.randn(N, C, R, R, generator=g)
    gy = torch.randn(N, C, R, R, generator=g)
    mod = Nw.SpadeNorm(kind, C, Lc, S, max_fm)
    st = {k: O.recipe_tensor("fz_" + kind, k, v.shape, 1.0) for k, v in mod.state_dict().items()}
    mod.load_state_dict(st)
    mod.cuda()
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    xs = nhwc(x).requires_grad_(grad)
    sty = style.cuda().requires_grad_(grad)
    if grad:
        h = mod(xs, labels, sty, True)
        h.backward(nhwc(gy))
        torch.cuda.synchronize()
        return h.detach(), xs.grad.detach(), [p.grad.detach().clone() for p in mod.parameters() if p.grad is not None]
    with torch.no_grad():
        h = mod(xs, labels, sty, True)
    torch.cuda.synchronize()
    return h.detach(), None, None


def timeit(kind, C, R, N, fused, grad, reps=5):
    use_plan(fused_norm=fused)
    g = torch.Generator().manual_seed(3)
    Lc, S, H = 19, 128, 256
    label = F.interpolate(torch.randint(0, Lc, (N, 1, 32, 32), generator=g).float(), size=(H, H), mode="nearest")
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1)
    mod = Nw.SpadeNorm(kind, C, Lc, S, 256).cuda()
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    xs = torch.randn(N, R, R, C, device="cuda")
    xs.dsee_layout = "nhwc"
    xs.requires_grad_(grad)
    sty = style.cuda().requires_grad_(grad)
    ts = []
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if grad:
            h = mod(xs, labels, sty, True)
        else:
            with torch.no_grad():
                h = mod(xs, labels, sty, True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        del h
    return min(ts[2:])


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "check"):
        for kind, C, R, N in [("sean", 64, 32, 2), ("spade", 64, 32, 2), ("sean", 128, 64, 3), ("spade", 64, 64, 2),
                              ("sean", 512, 64, 2), ("sean", 256, 128, 1)]:
            h1, dx1, gp1 = run(kind, C, R, N, True)
            h0, dx0, gp0 = run(kind, C, R, N, False)
            print("%-6s C=%d R=%d N=%d  h %.2e  dx %.2e  params %.2e  nan=%s" % (
                kind, C, R, N, rel(h1, h0), rel(dx1, dx0), max(rel(a, b) for a, b in zip(gp1, gp0)),
                bool(torch.isnan(h1).any())), flush=True)
    if what in ("all", "time"):
        for kind in ("sean", "spade"):
            for grad in (False, True):
                t1 = timeit(kind, 512, 256, 8, True, grad)
                t0 = timeit(kind, 512, 256, 8, False, grad)
                print("%-6s N=8 C=512 R=256 grad=%d: fused %.3f ms, round-2 path %.3f ms (whole norm forward, host-timed)" % (
                    kind, grad, t1, t0), flush=True)
        for R, N in ((128, 8), (64, 8)):
            t1 = timeit("sean", 512, R, N, True, False)
            t0 = timeit("sean", 512, R, N, False, False)
            print("sean N=%d C=512 R=%d nograd: fused %.3f ms, round-2 %.3f ms" % (N, R, t1, t0), flush=True)
