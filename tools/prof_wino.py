"""One R=256 512->512 Winograd conv forward + weight gradient in both GEMM modes (run under rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsee_amd import ops
from tools._plan import use_plan
n, r, c = 8, 256, 512
x = torch.randn(n, r, r, c, device="cuda"); w = torch.randn(c, c, 3, 3, device="cuda") * 0.02; g = torch.randn(n, r, r, c, device="cuda")
for split in (True, False):
    use_plan(gemm_split=split)
    for _ in range(3):
        ops._wino_conv(x, w, n, r, r, c, c, False)
        ops._wino_wgrad(x, g, n, r, r, c, c, c, c)
torch.cuda.synchronize()
