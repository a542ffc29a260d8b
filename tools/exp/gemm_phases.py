"""Per-wave cycle totals of the ping-pong GEMM loop (measurement build tools/exp/libabl_32.so, DSEE_LIB must point at it)."""
import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
one = torch.zeros(64 * 32, device="cuda"); one[0] = 3.0
for name, g, tg, n, k in [("conv 512->512 @256^2", 36, 32768, 512, 512), ("gamma/beta fwd @256^2", 288, 4096, 1024, 160)]:
    m = g * tg
    a = torch.randn(m, k, device="cuda"); c = torch.empty(m, n, device="cuda")
    b = torch.randn(g * n * k * 2, device="cuda").half().view(torch.int16)
    L.call("gemm_f16x2_af32", a, b, c, m, n, k, tg, n, 2, one, one); torch.cuda.synchronize()
    t = c.reshape(-1)[:64 * 8].reshape(8, 8, 8).cpu()
    slabs = (m // 256) * (n // 256) * (k // 16) / 256.0
    print(name, "slab steps per block ~%.0f" % slabs)
    for w in (0, 4):
        v = t[0, w] / slabs
        print("  wave %d cycles per slab: frag reads %.0f | DMA issue %.0f | wait+convert %.0f | barrier1 %.0f | MFMA(+stores) %.0f | barrier2 %.0f | sum %.0f"
              % (w, v[0], v[1], v[2], v[3], v[4], v[5], float(v[:6].sum())))
