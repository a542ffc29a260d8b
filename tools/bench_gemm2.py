"""Microbenchmark of the fp16x2 (and, with BF16X3=1, bf16x3) Winograd-domain GEMMs at the benchmark's shapes (N = 8)."""
import os, sys, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
BF = os.environ.get("BF16X3", "0") == "1"
def timeit(fn, it=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
one = torch.zeros(64 * 32, device="cuda"); one[0] = 3.0
nt = [("conv 512->512 @256^2 fwd/dgrad", 36, 32768, 512, 512), ("gamma/beta fwd @256^2 (per-image)", 288, 4096, 1024, 160),
      ("gamma/beta adjoint dgrad @256^2", 36, 32768, 128, 1024), ("conv 512->512 @128^2", 36, 8192, 512, 512),
      ("gamma/beta fwd @128^2", 288, 1024, 1024, 160), ("conv 512->512 @64^2", 36, 2048, 512, 512)]
for name, g, tg, n, k in nt:
    m = g * tg
    a = torch.randn(m, k, device="cuda")
    c = torch.empty(m, n, device="cuda")
    terms = 3 if BF else 2
    b = torch.randn(g * n * k * terms, device="cuda").to(torch.bfloat16 if BF else torch.float16).view(torch.int16)
    for tile in (1, 2):
        if tile == 2 and (n % 256 or tg % 256): continue
        if BF:
            t = timeit(lambda: L.call("gemm_bf16x3_af32", a, b, c, m, n, k, tg, n, tile))
        else:
            t = timeit(lambda: L.call("gemm_f16x2_af32", a, b, c, m, n, k, tg, n, tile, one, one))
        fl, by = 2.0 * m * n * k, m * k * 4.0 + m * n * 4.0 + g * n * k * 2.0 * terms
        print("%-36s tile %s: %.3f ms  %4.0f TF/s fp32-eq  %.2f GB -> %.2f TB/s" % (name, "128" if tile == 1 else "256", t, fl / t / 1e9, by / 1e9, by / t / 1e9))
tn = [("wgrad conv 512x512 @256^2", 36, 32768, 512, 512, 1), ("wgrad conv 512x512 @256^2 split 8", 36, 32768, 512, 512, 8),
      ("wgrad conv 512x512 @256^2 split 16", 36, 32768, 512, 512, 16), ("wgrad conv 512x512 @256^2 split 32", 36, 32768, 512, 512, 32),
      ("wgrad conv 512x512 @128^2 split 8", 36, 8192, 512, 512, 8), ("wgrad conv 512x512 @64^2 split 4", 36, 2048, 512, 512, 4),
      ("wgrad conv 512x512 @64^2 split 2", 36, 2048, 512, 512, 2), ("wgrad table 1024x160 @256^2", 288, 4096, 1024, 160, 1),
      ("wgrad conv 512x512 @128^2", 36, 8192, 512, 512, 2), ("wgrad table 1024x160 @128^2", 288, 1024, 1024, 160, 1)]
for name, g, t_, rp, rq, sp in tn:
    p = torch.randn(g * t_, rp, device="cuda"); q = torch.randn(g * t_, rq, device="cuda")
    c = torch.empty(g * sp, rp, rq, device="cuda")
    if BF:
        t = timeit(lambda: L.call("gemm_bf16x3_tn_f32", p, q, c, g, t_, rp, rq, rq, sp))
    else:
        t = timeit(lambda: L.call("gemm_f16x2_tn_f32", p, q, c, g, t_, rp, rq, rq, sp, one, one))
    fl, by = 2.0 * g * t_ * rp * rq, g * t_ * (rp + rq) * 4.0
    print("%-36s: %.3f ms  %4.0f TF/s fp32-eq  %.2f GB -> %.2f TB/s" % (name, t, fl / t / 1e9, by / 1e9, by / t / 1e9))
