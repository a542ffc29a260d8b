"""Does the 256 MB Infinity Cache hold the Winograd intermediates of a conv chunk?  (VERDICT r3 #1a)
(1) device copy bandwidth vs buffer size (same buffers re-used: write-allocate + read-hit would exceed the HBM rate);
(2) the 512 -> 512 forward chain (input transform -> GEMM -> output transform) over 32 images of 128^2 (= the tile count of
    N = 8 at 256^2), in chunks of nb images with ONE re-used V / M buffer pair, per-kernel HIP-event times."""
import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import ops, lib as L


def ev():
    return torch.cuda.Event(enable_timing=True)


print("== copy bandwidth vs size (dst.copy_(src), both re-used)")
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a, b = torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
    for _ in range(3):
        b.copy_(a)
    s, e = ev(), ev()
    it = max(4, 4096 // mb)
    s.record()
    for _ in range(it):
        b.copy_(a)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / it
    print("  %5d MB: %.3f ms  %.2f TB/s (read + write)" % (mb, t, 2 * n * 4 / 1e9 / t))
    del a, b

print("== write-then-read of a scratch buffer between two kernels (fill_ then sum) vs size")
for mb in (32, 64, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device="cuda")
    o = torch.empty(n // 4, device="cuda")
    for _ in range(2):
        a.fill_(1.0)
        torch.add(a[: n // 4], a[n // 4: n // 2], out=o)
    s, e, e2 = ev(), ev(), ev()
    it = 8
    tf = tr = 0.0
    for _ in range(it):
        s.record()
        a.fill_(1.0)
        e.record()
        torch.add(a[: n // 4], a[n // 4: n // 2], out=o)
        e2.record()
        torch.cuda.synchronize()
        tf += s.elapsed_time(e)
        tr += e.elapsed_time(e2)
    print("  %5d MB: fill %.3f ms (%.2f TB/s)   read half back %.3f ms (%.2f TB/s read)" %
          (mb, tf / it, n * 4 / 1e9 / (tf / it), tr / it, n * 2 / 1e9 / (tr / it)))
    del a, o

print("== conv 512 -> 512 forward chain, 32 images of 128^2 (32768 tiles), chunks of nb images")
N, R, Cc = 32, 128, 512
x = torch.randn(N, R, R, Cc, device="cuda")
y = torch.empty_like(x)
w = torch.randn(Cc, Cc, 3, 3, device="cuda") * 0.02
ax = ops.amax_slot()
L.call("absmax", x, x.numel(), ax)
rows, kp = L.wrows(Cc), L.kpad(1, 1, Cc)
u, ua = ops._wino_u(w, Cc, Cc, False, rows, kp, 2)
tpi = (R // 4) ** 2
for nb in (1, 2, 3, 4, 8, 16, 32):
    if N % nb and nb != 3:
        continue
    t = nb * tpi
    v2 = ops._i16(36 * t * Cc * 2)
    m = ops.new(36, t, Cc)
    chunks = [(n0, nb) for n0 in range(0, N - nb + 1, nb)]

    def run(rec=None):
        for n0, k in chunks:
            xc, yc = x[n0:n0 + k], y[n0:n0 + k]
            e0, e1, e2, e3 = ev(), ev(), ev(), ev()
            e0.record()
            L.call("wino43_input_f16x2", xc, v2, k, R, R, Cc, ax, ops.FUSED_V_BOUND)
            e1.record()
            L.call("gemm_f16x2_pre", v2, u, m, 36 * t, Cc, Cc, t, rows, ax, ops.FUSED_V_BOUND, ua)
            e2.record()
            L.call("wino43_output", m, None, None, Cc, yc, k, R, R, Cc, 0, 0.2, None, 0, 0, None, 0, 0, None)
            e3.record()
            if rec is not None:
                rec.append((e0, e1, e2, e3))
    run()
    run()
    torch.cuda.synchronize()
    rec = []
    s, e = ev(), ev()
    s.record()
    run(rec)
    e.record()
    torch.cuda.synchronize()
    done = len(chunks) * nb
    sc = N / done        # scale to the full 32 images
    ti = sum(a.elapsed_time(b) for a, b, _, _ in rec) * sc
    tg = sum(b.elapsed_time(c) for _, b, c, _ in rec) * sc
    to = sum(c.elapsed_time(d) for _, _, c, d in rec) * sc
    print("  nb=%2d (T=%5d, V %4.0f MB + M %4.0f MB, %4d GEMM tiles): input %.3f  gemm %.3f  output %.3f  sum %.3f | wall %.3f ms" %
          (nb, t, 36 * t * Cc * 4 / 1e6, 36 * t * Cc * 4 / 1e6, (36 * t // 256) * 2, ti, tg, to, ti + tg + to,
           s.elapsed_time(e) * sc))
    del v2, m
