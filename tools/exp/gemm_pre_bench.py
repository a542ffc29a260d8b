"""Pre-split-A GEMM against the fp32-A form at the conv 512 -> 512 @256^2 bs=8 shape (M = 36 x 32768, N = K = 512)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deepsee_amd import lib as L, ops
t, k, n = 32768, 512, 512
M = 36 * t
a = torch.randn(M, k, device="cuda")
a2 = (torch.randn(M * k * 2, device="cuda") * 1000).half().view(torch.int16)     # (realistic bit patterns: power follows the data)
b2 = (torch.randn(36 * n * k * 2, device="cuda") * 1000).half().view(torch.int16)
c = torch.empty(M, n, device="cuda")
am = torch.zeros(64 * 32, device="cuda"); am[0] = 4.0
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t_f32 = timeit(lambda: L.call("gemm_f16x2_af32", a, b2, c, M, n, k, t, n, 2, am, am))
t_pre = timeit(lambda: L.call("gemm_f16x2_pre", a2, b2, c, M, n, k, t, n, am, 100.0, am))
fl = 2.0 * M * n * k
print("conv 512->512 @256^2: fp32 A %.3f ms (%.0f TF/s fp32-eq, %.2f of 839) | pre-split A %.3f ms (%.0f TF/s, %.2f)"
      % (t_f32, fl / t_f32 / 1e9, fl / t_f32 / 1e9 / 839, t_pre, fl / t_pre / 1e9, fl / t_pre / 1e9 / 839))
