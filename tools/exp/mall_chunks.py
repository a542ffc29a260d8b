"""Does running a Winograd layer in image chunks small enough for V and M to stay in the 256 MB Infinity Cache pay?  Forward of
one 512 -> 512 layer, N = 8, with the images-per-pass capped (ops._wino_chunk)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import ops
def timeit(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
orig = ops._wino_chunk
c = 512
for r in (64, 128, 256):
    x = torch.randn(8, r, r, c, device="cuda"); w = torch.randn(c, c, 3, 3, device="cuda") * 0.02
    ops.tag_amax(x, ops.amax_slot()); ops.L.call("absmax", x, x.numel(), x.dsee_amax)
    ref = None
    for cap in (8, 4, 2, 1):
        ops._wino_chunk = lambda n, h, w_, cmax, per_image=False, cap=cap: min(orig(n, h, w_, cmax, per_image) or 0, cap) or None
        t = timeit(lambda: ops._wino_conv(x, w, 8, r, r, c, c, False))
        y = ops._wino_conv(x, w, 8, r, r, c, c, False)
        ref = y if ref is None else ref
        v_mb = cap * (r // 4) ** 2 * c * 36 * 4 / 1e6
        print("R = %3d, %d image(s) per pass (V = M = %4.0f MB): %.3f ms, max |diff| vs whole batch %.1e" % (r, cap, v_mb, t, float((y - ref).abs().max())))
ops._wino_chunk = orig
