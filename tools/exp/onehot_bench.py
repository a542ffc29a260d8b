"""Stand-alone timing of dsee_onehot_conv3x3_fwd (mlp_shared over one-hot labels, 19 -> 128 channels + the 32 one-hot channels) at 256^2."""
import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import ops, lib as L
n, h, co, l = 8, 256, 128, 19
g = torch.Generator(device="cuda").manual_seed(1)
lab = torch.randint(0, l, (n, h, h), device="cuda", generator=g, dtype=torch.uint8)
wt = torch.randn(9, l, co, device="cuda", generator=g); bias = torch.randn(co, device="cuda", generator=g)
out = torch.empty(n, h, h, 160, device="cuda")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
t = timeit(lambda: L.call("onehot_conv3x3_fwd", lab, wt, bias, out, n, h, h, 0, l, co, 160, 0, 1, 128, None, 0.0))
print("onehot conv fwd 256^2: %.3f ms  %.2f TB/s of output" % (t, out.numel() * 4 / 1e9 / t))
