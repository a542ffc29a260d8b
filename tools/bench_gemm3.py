"""Microbenchmark of dsee_gemm_bf16x3 (fp32 GEMM on the bf16 matrix cores) on the Winograd-domain shapes."""
import sys, ctypes as C, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
def timeit(fn, it=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
shapes = [(36, 32768, 512, 512), (36, 16384, 1024, 160), (36, 16384, 128, 1024), (36, 8192, 512, 512), (36, 2048, 512, 512), (36, 512, 512, 512)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for (g, tg, n, k) in shapes:
    m = g * tg
    a3 = (torch.randn(m * k * 3 // 2, device="cuda")).bfloat16().view(torch.int16)   # plain bf16 values in the slots
    a3 = torch.cat([a3, a3])[: m * k * 3]
    b3 = torch.randn(g * n * k * 3, device="cuda").bfloat16().view(torch.int16)
    c = torch.empty(m, n, device="cuda")
    af = torch.randn(m, k, device="cuda")
    for tile in (1, 2):
        if tile == 2 and (n % 256 or tg % 256): continue
        t = timeit(lambda: L.call("gemm_bf16x3_af32", af, b3, c, C.c_long(m), n, k, C.c_long(tg), n, tile))
        print("   fp32-A variant tile=%s: %.3f ms  %.0f TF/s" % ("128x128" if tile == 1 else "256x256", t, 2.0 * m * n * k / t / 1e9))
    for tile in (1, 2):
        if tile == 2 and (n % 256 or tg % 256): continue
        t = timeit(lambda: L.call("gemm_bf16x3", a3, b3, c, C.c_long(m), n, k, C.c_long(tg), n, tile))
        fl = 2.0 * m * n * k
        print("G=%d Tg=%d N=%d K=%d tile=%s: %.3f ms  %.0f TF/s fp32-equivalent (%.0f%% of 419)  A+B+C bytes %.2f GB -> %.2f TB/s" % (
            g, tg, n, k, "128x128" if tile == 1 else "256x256", t, fl / t / 1e9, fl / t / 1e9 / 4.194, (m * k * 6 + m * n * 4) / 1e9, (m * k * 6 + m * n * 4) / t / 1e9))
