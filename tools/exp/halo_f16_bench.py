"""3 x 3 / stride 1 direct layers on split fp16 operands: halo kernel (flags 0) against the implicit-GEMM kernel (DSEE_CONV_NO_HALO)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import lib as L
def timeit(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for (n, r, ci, co) in [(8, 256, 4, 64), (8, 256, 64, 64), (8, 128, 64, 128), (8, 128, 128, 64), (8, 64, 128, 256), (16, 64, 64, 64)]:
    x = torch.randn(n, r, r, ci, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    geom = L.geom_fwd(n, r, r, ci, co, 3, 1, 1)
    wp = torch.empty(L.wrows(co), L.kpad(3, 3, ci), device="cuda")
    L.call("pack_weight_fwd", w, None, None, wp, co, ci, 3, 3, ci, geom.korder)
    ax, aw = torch.zeros(2048, device="cuda"), torch.zeros(2048, device="cuda")
    L.call("absmax", x, x.numel(), ax); L.call("absmax", wp, wp.numel(), aw)
    outs, line = [], []
    for flags in (0, 1):
        out = torch.empty(n, r, r, co, device="cuda")
        t = timeit(lambda: L.call("conv2d_fwd_f16x2_amax", C.byref(geom), x, wp, None, None, 0, out, 2, 0.2, ax, aw, None, flags))
        outs.append(out); line.append("%s %.3f ms (%.0f TF/s)" % ("halo" if flags == 0 else "igemm", t, 2.0 * n * r * r * ci * co * 9 / t / 1e9))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).relu().permute(0, 2, 3, 1)
    errs = [float((o.double() - ref).norm() / ref.norm()) for o in outs]
    print("N=%d %dx%d %d->%d: %s | rel err vs float64: halo %.1e, igemm %.1e" % (n, r, r, ci, co, ", ".join(line), errs[0], errs[1]))
