"""Target for rocprofv3 --pmc: the fp16x2 NT GEMM at the dominant shape (conv 512->512 @256^2 bs=8) and the TN
weight-gradient GEMM of the same layer (split-K 8), 3 launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsee_amd import lib as L
one = torch.zeros(64 * 32, device="cuda"); one[0] = 3.0
g, tg, n, k = 36, 32768, 512, 512
m = g * tg
a = torch.randn(m, k, device="cuda"); c = torch.empty(m, n, device="cuda")
b = torch.randn(g * n * k * 2, device="cuda").half().view(torch.int16)
for _ in range(3):
    L.call("gemm_f16x2_af32", a, b, c, m, n, k, tg, n, 2, one, one)
p = torch.randn(g * tg, 512, device="cuda"); q = torch.randn(g * tg, 512, device="cuda")
cw = torch.empty(g * 8, 512, 512, device="cuda")
for _ in range(3):
    L.call("gemm_f16x2_tn_f32", p, q, cw, g, tg, 512, 512, 512, 8, one, one)
torch.cuda.synchronize()
