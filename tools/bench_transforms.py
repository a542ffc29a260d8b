"""Microbenchmark of the memory-bound Winograd-domain passes at the benchmark's dominant shape (N=8, 256^2, C=512):
the fused SPADE/SEAN output transform, the plain output transform, the input transform.  GB = bytes the kernel itself
moves (algorithmic for its formulation)."""
import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
def timeit(fn, it=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
n, h, c = 8, 256, 512
T = n * (h // 4) ** 2
x = torch.randn(n, h, h, c, device="cuda")
mean = torch.zeros(c, device="cuda"); inv = torch.ones(c, device="cuda")
M2 = torch.randn(36, T, 2 * c, device="cuda")
out = torch.empty_like(x); sc = torch.empty_like(x)
t = timeit(lambda: L.call("wino43_output_modulate", M2, None, x, mean, inv, out, sc, n, h, h, c, 2 * c, 1.0, 0.2, None))
gb = (M2.numel() + 3 * x.numel()) * 4 / 1e9
print("output_modulate (M 2C + x -> h, scale): %.3f ms  %.2f GB -> %.2f TB/s (%.0f %% of 8)" % (t, gb, gb / t, gb / t / 8 * 100))
t = timeit(lambda: L.call("wino43_output_modulate", M2, None, x, mean, inv, out, None, n, h, h, c, 2 * c, 1.0, 0.2, None))
gb = (M2.numel() + 2 * x.numel()) * 4 / 1e9
print("output_modulate, no scale output       : %.3f ms  %.2f GB -> %.2f TB/s (%.0f %% of 8)" % (t, gb, gb / t, gb / t / 8 * 100))
M1 = M2[:, :, :c].contiguous()
t = timeit(lambda: L.call("wino43_output", M1, None, None, 0, out, n, h, h, c, 0, 0.0, None, 0, 0, None, 0, 0, None))
gb = (M1.numel() + x.numel()) * 4 / 1e9
print("output (M -> y)                        : %.3f ms  %.2f GB -> %.2f TB/s (%.0f %% of 8)" % (t, gb, gb / t, gb / t / 8 * 100))
am = torch.zeros(64 * 32, device="cuda")
t = timeit(lambda: L.call("wino43_input", x, M1, n, h, h, c, am))
print("input (x -> V, max|V|)                 : %.3f ms  %.2f GB -> %.2f TB/s (%.0f %% of 8)" % (t, gb, gb / t, gb / t / 8 * 100))
t = timeit(lambda: out.copy_(x))
print("torch copy of x                        : %.3f ms  %.2f TB/s" % (t, 2 * x.numel() * 4 / 1e9 / t))
