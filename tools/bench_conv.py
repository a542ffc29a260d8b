"""Micro-benchmark of the conv kernels at the hot-path shapes (config 2: N=8, C=512)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L


def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def main():
    dev = "cuda"
    for (n, r, cin, cout) in [(8, 64, 512, 512), (8, 128, 512, 512), (8, 256, 512, 512), (8, 256, 256, 1024)]:
        geom = L.geom_fwd(n, r, r, cin, cout, 3, 1, 1)
        x = torch.randn(n, r, r, cin, device=dev)
        wp = torch.randn(L.wrows(cout), L.kpad(3, 3, cin), device=dev) * 0.02
        out = torch.empty(n, r, r, cout, device=dev)
        ms = timeit(lambda: L.call("conv2d_fwd", C.byref(geom), x, wp, None, None, 0, out, 0, 0.2))
        fl = 2.0 * n * r * r * cin * 9 * cout
        print("fwd   N=%d R=%d %d->%d: %.3f ms  %.1f TF/s (%.0f%% of 157.3)" % (n, r, cin, cout, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100))
        gy = torch.randn(n, r, r, cout, device=dev)
        wsb = L.lib().dsee_conv2d_wgrad_workspace(C.byref(geom))
        ws = torch.empty(wsb // 4, device=dev)
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        ms = timeit(lambda: L.call("conv2d_wgrad", C.byref(geom), x, gy, ws, C.c_size_t(wsb), dw, cout, 0, cin))
        print("wgrad N=%d R=%d %d->%d: %.3f ms  %.1f TF/s (%.0f%%)  ws=%.0f MB" % (n, r, cin, cout, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100, wsb / 1e6))


main()
