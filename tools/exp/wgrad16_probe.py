"""fp32-MFMA vs fp16x2 weight-gradient kernel on the step's non-Winograd geometries."""
import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import ops, lib as L
from tools._plan import use_plan
def timeit(fn, it=6):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
g = torch.Generator().manual_seed(0)
cases = [("160->128 k3 @256^2 N8", 8, 256, 160, 128, 3, 1, 1, 0), ("128->256 k3 ups @64 N8", 8, 64, 128, 256, 3, 1, 1, 1),
         ("160->128 k3 @128^2 N8", 8, 128, 160, 128, 3, 1, 1, 0), ("64->64 k3 @256^2 N8", 8, 256, 64, 64, 3, 1, 1, 0),
         ("128->256 k4 s1 @33 N16", 16, 33, 128, 256, 4, 1, 2, 0), ("32->64 k4 s2 @129 N16", 16, 129, 32, 64, 4, 2, 2, 0),
         ("64->128 k3 @128^2 N8", 8, 128, 64, 128, 3, 1, 1, 0)]
for name, n, h, ci, co, k, s, p, ups in cases:
    geom = L.geom_fwd(n, h, h, ci, co, k, s, p, ups)
    x = torch.randn(n, h, h, ci, generator=g).cuda(); dy = torch.randn(n, geom.Ho, geom.Wo, co, generator=g).cuda()
    res = {}
    for mode, thr in (("f32", 0.0), ("f16x2", 1.0)):
        use_plan(conv_f16x2_min_flop=thr)
        dw = ops.wgrad_raw(x, dy, geom, co, ci, k, k)
        res[mode] = (dw, timeit(lambda: ops.wgrad_raw(x, dy, geom, co, ci, k, k)))
    fl = ops._flops(geom)
    print("%-28s f32 %.3f ms (%5.1f TF/s) | f16x2 %.3f ms (%5.1f TF/s incl. 2 absmax) | rel diff %.1e" % (
        name, res["f32"][1], fl / res["f32"][1] / 1e9, res["f16x2"][1], fl / res["f16x2"][1] / 1e9,
        float((res["f16x2"][0] - res["f32"][0]).norm() / res["f32"][0].norm())))
