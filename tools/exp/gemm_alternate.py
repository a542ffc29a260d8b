"""Does the Winograd-domain GEMM run slower when it alternates with a light (HBM-bound) kernel, as it does inside the step,
than back to back?  Times the GEMM launches alone (events around each) in three sequences."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deepsee_amd import lib as L
res = 256; t, k, n = 8 * (res // 4) ** 2, 512, 512; M = 36 * t
x = torch.nn.functional.leaky_relu(torch.randn(8, res, res, k, device="cuda"), 0.2)
am = torch.zeros(64 * 32, device="cuda"); L.call("absmax", x, x.numel(), am)
a2 = torch.empty(M * k * 2, dtype=torch.int16, device="cuda")
L.call("wino43_input_f16x2", x, a2, 8, res, res, k, am, 100.0)
w = torch.randn(n, k, 3, 3, device="cuda") * 0.02
amb = torch.zeros(64 * 32, device="cuda"); L.call("absmax", w, w.numel(), amb)
b2 = torch.empty(36 * n * k * 2, dtype=torch.int16, device="cuda")
L.call("wino43_weights", w, b2, n, k, 0, 2, amb)
c = torch.empty(M, n, device="cuda")
y = torch.empty_like(x)
entry = sys.argv[1] if len(sys.argv) > 1 else "gemm_f16x2_pre_w4"
gemm = lambda: L.call(entry, a2, b2, c, M, n, k, t, n, am, 100.0, amb)
light = lambda: L.call("wino43_input_f16x2", x, a2, 8, res, res, k, am, 100.0)      # 0.7 ms, HBM-bound
copy = lambda: y.copy_(x)
def seq(name, before, reps=30):
    for _ in range(5):
        for f in before: f()
        gemm()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        for f in before: f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gemm(); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print("%s %-40s GEMM alone: median %.3f ms, min %.3f, max %.3f" % (entry, name, ts[len(ts) // 2], ts[0], ts[-1]))
seq("back to back", [])
seq("after one input transform (0.7 ms)", [light])
seq("after transform + 3 copies (~3 ms light)", [light, copy, copy, copy])
seq("after 10 copies (~8 ms light)", [copy] * 10)
seq("back to back again", [])
