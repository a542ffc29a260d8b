"""Stand-alone timing of dsee_thin1x1_bwd at the to-RGB shape (M = 8 x 256^2, C = 512, K = 27)."""
import sys, ctypes as C, torch
sys.path.insert(0, ".")
from deepsee_amd import ops, lib as L
m, c, k, ldz = 8 * 256 * 256, 512, 27, 28
g = torch.Generator(device="cuda").manual_seed(1)
dz = torch.randn(m, ldz, device="cuda", generator=g); w = torch.randn(k, c, device="cuda", generator=g)
x = torch.randn(m, c, device="cuda", generator=g); dx = torch.empty_like(x); dw = torch.empty(k, c, device="cuda")
ws = ops.scratch(L.lib().dsee_thin1x1_bwd_workspace(c, k), "wgrad")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
t = timeit(lambda: L.call("thin1x1_bwd", dz, ldz, w, None, dx, None, C.c_long(m), c, k, None, 0, 0.2, None))
print("data gradient  : %.3f ms  %.2f TB/s" % (t, x.numel() * 4 / 1e9 / t))
am = ops.amax_slot()
t = timeit(lambda: L.call("thin1x1_bwd", dz, ldz, w, x, dx, None, C.c_long(m), c, k, None, 1, 0.2, am))
print("data gradient + LeakyReLU' + max: %.3f ms  %.2f TB/s" % (t, 2 * x.numel() * 4 / 1e9 / t))
t = timeit(lambda: L.call("thin1x1_bwd", dz, ldz, None, x, None, dw, C.c_long(m), c, k, ws, 0, 0.2, None))
print("weight gradient: %.3f ms  %.2f TB/s" % (t, x.numel() * 4 / 1e9 / t))
