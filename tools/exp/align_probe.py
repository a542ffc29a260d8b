import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
def timeit(fn, it=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
n, h, c = 8, 256, 512
T = n * (h // 4) ** 2
numel = n * h * h * c
mean = torch.zeros(c, device="cuda"); inv = torch.ones(c, device="cuda")
pool = torch.empty(36 * T * 2 * c + 3 * numel + (64 << 20), device="cuda")
print("pool base %x" % pool.data_ptr())
for offs in [(0, 0, 0), (0, 1024, 2048), (0, 64 * 1024, 128 * 1024), (0, 1 << 20, 2 << 20), (0, 4096 + 256, 8192 + 512), (0, 33 * 4096, 66 * 4096), (0, (1 << 21) + 4096, (1 << 22) + 8192)]:
    p = 0
    M2 = pool[p:p + 36 * T * 2 * c].view(36, T, 2 * c); p += M2.numel()
    bufs = []
    for o in offs:
        p += o // 4
        bufs.append(pool[p:p + numel].view(n, h, h, c)); p += numel
    x, out, sc = bufs
    ts = [timeit(lambda: L.call("wino43_output_modulate", M2, None, x, mean, inv, out, sc, n, h, h, c, 2 * c, 1.0, 0.2, None)) for _ in range(3)]
    print(offs, " ".join("%.3f" % t for t in ts))
