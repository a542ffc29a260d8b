import ctypes as C, sys, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
def timeit(fn, it=6):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for (n, r, cin, cout) in [(8, 256, 512, 512), (8, 128, 512, 512), (8, 64, 512, 512)]:
    geom = L.geom_fwd(n, r, r, cin, cout, 3, 1, 1)
    x = torch.randn(n, r, r, cin, device="cuda"); wp = torch.randn(L.wrows(cout), L.kpad(3, 3, cin), device="cuda") * 0.02
    out = torch.empty(n, r, r, cout, device="cuda")
    ms = timeit(lambda: L.call("conv2d_fwd", C.byref(geom), x, wp, None, None, 0, out, 0, 0.2))
    fl = 2.0 * n * r * r * cin * 9 * cout
    print("fwd R=%d: %.3f ms %.1f TF/s (%.1f%%)" % (r, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100))
