"""Run only the 512->512 3x3 conv at N=8, R=256 (fwd kernel, then wgrad) a few times: target for rocprofv3 --pmc."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
n, r, cin, cout = 8, 256, 512, 512
geom = L.geom_fwd(n, r, r, cin, cout, 3, 1, 1)
x = torch.randn(n, r, r, cin, device="cuda"); wp = torch.randn(L.wrows(cout), L.kpad(3, 3, cin), device="cuda") * 0.02
out = torch.empty(n, r, r, cout, device="cuda"); gy = torch.randn(n, r, r, cout, device="cuda")
wsb = L.lib().dsee_conv2d_wgrad_workspace(C.byref(geom)); ws = torch.empty(wsb // 4, device="cuda"); dw = torch.empty(cout, cin, 3, 3, device="cuda")
for _ in range(3):
    L.call("conv2d_fwd", C.byref(geom), x, wp, None, None, 0, out, 0, 0.2)
    L.call("conv2d_wgrad", C.byref(geom), x, gy, ws, C.c_size_t(wsb), dw, cout, 0, cin)
torch.cuda.synchronize()
