"""Stand-alone timing of the upsample + NoiseInjection pass (dsee_upsample_noise_rng_fwd[_stats]) and of sumpool at the top shape."""
import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import ops, lib as L
n, h, c = 8, 256, 512
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(n, h // 2, h // 2, c, device="cuda", generator=g)
nw = torch.randn(c, device="cuda", generator=g)
y = torch.empty(n, h, h, c, device="cuda")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
gb = (x.numel() + y.numel()) * 4 / 1e9
t = timeit(lambda: L.call("upsample_noise_rng_fwd", x, nw, y, n, h, h, c, 1, 7, 0))
print("up + rng noise        : %.3f ms  %.2f TB/s" % (t, gb / t))
import ctypes as C; rows = L.lib().dsee_stats_part_rows(C.c_long(n * h * h * c // 4))
sp = torch.empty(rows * 3 * c, device="cuda")
t = timeit(lambda: L.call("upsample_noise_rng_fwd_stats", x, nw, y, n, h, h, c, 1, 7, 0, sp))
print("up + rng noise + stats: %.3f ms  %.2f TB/s" % (t, gb / t))
t = timeit(lambda: L.call("upsample_noise_fwd", x, None, None, y, n, h, h, c, 1))
print("up only               : %.3f ms  %.2f TB/s" % (t, gb / t))
t = timeit(lambda: L.call("rng_fill", y, y.numel(), 7, 0, 1))
print("rng_fill normal       : %.3f ms  %.2f TB/s (write only)" % (t, y.numel() * 4 / 1e9 / t))
t = timeit(lambda: L.call("sumpool", y, x, n, h, h, c, 1))
print("sumpool               : %.3f ms  %.2f TB/s" % (t, gb / t))
