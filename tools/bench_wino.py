import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import ops, lib as L
from tools._plan import use_plan
def timeit(fn, it=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for (n, r, c) in [(8, 256, 512), (8, 128, 512), (8, 64, 512), (8, 32, 512)]:
    x = torch.randn(n, r, r, c, device="cuda"); w = torch.randn(c, c, 3, 3, device="cuda") * 0.02
    use_plan(gemm_split=False)
    tw = timeit(lambda: ops._wino_conv(x, w, n, r, r, c, c, False))
    yf = ops._wino_conv(x, w, n, r, r, c, c, False)
    use_plan()
    ts = timeit(lambda: ops._wino_conv(x, w, n, r, r, c, c, False))
    ys = ops._wino_conv(x, w, n, r, r, c, c, False)
    print("R=%d: winograd bf16x3 %.3f ms (%.0f TF/s algorithmic) | rel diff vs f32 mfma %.2e" % (r, ts, 2.0 * n * r * r * c * 9 * c / ts / 1e9, ((ys - yf).norm() / yf.norm()).item()))
    geom = L.geom_fwd(n, r, r, c, c, 3, 1, 1)
    wp = ops._pack_fwd(w, c, 1)
    td = timeit(lambda: ops.conv_raw(x, wp, geom))
    fl = 2.0 * n * r * r * c * 9 * c
    print("R=%d: winograd %.3f ms (%.0f TF/s algorithmic) | direct halo %.3f ms (%.0f TF/s)" % (r, tw, fl / tw / 1e9, td, fl / td / 1e9))
    ops.PROFILE = {}
    ops._wino_conv(x, w, n, r, r, c, c, False); torch.cuda.synchronize()
    for k, v in ops.PROFILE.items(): print("   ", k, sum(s.elapsed_time(e) for s, e, _ in v), "ms", sum(f for _, _, f in v) / sum(s.elapsed_time(e) for s, e, _ in v) / 1e9, "TF/s executed")
    ops.PROFILE = None

    g = torch.randn(n, r, r, c, device="cuda")
    use_plan()
    tws = timeit(lambda: ops._wino_wgrad(x, g, n, r, r, c, c, c, c))
    a3 = ops._wino_wgrad(x, g, n, r, r, c, c, c, c)
    use_plan(gemm_split=False)
    tw = timeit(lambda: ops._wino_wgrad(x, g, n, r, r, c, c, c, c))
    td = timeit(lambda: ops.wgrad_raw(x, g, geom, c, c, 3, 3))
    a, b = ops._wino_wgrad(x, g, n, r, r, c, c, c, c), ops.wgrad_raw(x, g, geom, c, c, 3, 3)
    print("R=%d wgrad: winograd bf16x3 %.3f ms (%.0f TF/s algorithmic) | rel diff vs direct %.2e" % (r, tws, fl / tws / 1e9, ((a3 - b).norm() / b.norm()).item()))
    print("R=%d wgrad: winograd %.3f ms (%.0f TF/s algorithmic) | direct %.3f ms (%.0f TF/s) | rel diff %.2e" % (r, tw, fl / tw / 1e9, td, fl / td / 1e9, ((a - b).norm() / b.norm()).item()))
