"""Run dsee_gemm_bf16x3 on the dominant Winograd shape a few times: target for rocprofv3 --pmc."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsee_amd import lib as L
g, tg, n, k = 36, 32768, 512, 512
m = g * tg
a3 = torch.randn(m * k * 3 // 2, device="cuda").bfloat16().view(torch.int16); a3 = torch.cat([a3, a3])[: m * k * 3]
b3 = torch.randn(g * n * k * 3, device="cuda").bfloat16().view(torch.int16)
c = torch.empty(m, n, device="cuda")
for _ in range(3):
    L.call("gemm_bf16x3", a3, b3, c, C.c_long(m), n, k, C.c_long(tg), n, int(os.environ.get("TILE", "2")))
torch.cuda.synchronize()
