"""Stand-alone timing of the norm backward's reduce pass (dsee_modulate_bwd_reduce_wino_f16x2 / _f16p) and apply pass at the
benchmark's top shape (N = 8, 256^2, C = 512)."""
import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import ops, lib as L
n, h, c = 8, 256, 512
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(n, h, h, c, device="cuda", generator=g)
dh = torch.randn(n, h, h, c, device="cuda", generator=g) * 0.01
out = torch.randn(n, h, h, c, device="cuda", generator=g)
scale = torch.rand(n, h, h, c, device="cuda", generator=g) + 0.5
mean, invstd = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
rows, t = 2 * c, n * (h // 4) ** 2
ws = ops.scratch(L.lib().dsee_modulate_bwd_wino_workspace(n, h, h, c), "norm")
ga = torch.zeros(2048, device="cuda"); ga[0] = 1.0
sums = ops.new(4, c)
# "mask" on the command line: the LeakyReLU branch from the fused forward's bit mask instead of `out`
MASK = torch.randint(-2**31, 2**31 - 1, (n * h * h * (c // 32),), dtype=torch.int32, device="cuda") if "mask" in sys.argv else None
if MASK is not None:
    out = None
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
dm2 = ops._i16(36 * t * rows * 2)
t2 = timeit(lambda: L.call("modulate_bwd_reduce_wino_f16x2", dh, out, x, scale, mean, invstd, dm2, rows, sums, n, h, h, c, 0.2, ws, ga, 1.0, MASK))
print("reduce two-term : %.3f ms  (%.2f TB/s on 4 x 1.07 GB read + 4.8 GB written)" % (t2, (4 * x.numel() * 4 + dm2.numel() * 2) / t2 / 1e9))
dm1 = ops._i16(36 * t * rows)
scale16 = scale.half()
t1 = timeit(lambda: L.call("modulate_bwd_reduce_wino_f16p", dh, out, x, scale16, mean, invstd, dm1, rows, sums, n, h, h, c, 0.2, ws, ga, 1.0, MASK))
print("reduce packed   : %.3f ms  (%.2f TB/s)" % (t1, (4 * x.numel() * 4 + dm1.numel() * 2) / t1 / 1e9))
dx = torch.empty_like(x); da = ops.amax_slot()
ta = timeit(lambda: L.call("modulate_bwd_apply_amax", dh, out, x, scale, mean, invstd, sums, None, dx, n, h * h, c, 1.0 / (n * h * h), 0.2, da, 0, MASK))
print("apply           : %.3f ms  (%.2f TB/s)" % (ta, 5 * x.numel() * 4 / ta / 1e9))
