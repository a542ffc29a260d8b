"""fp32-MFMA vs fp16x2 direct convolution kernels on the step's non-Winograd geometries: time and error vs float64."""
import sys, torch
import torch.nn.functional as F
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
cases = [("64->64 k3 @256^2 N8", 8, 256, 64, 64, 3, 1, 1, 0), ("64->128 k3 @128^2 N8", 8, 128, 64, 128, 3, 1, 1, 0),
         ("128->256 k3 ups @64->128 N8", 8, 64, 128, 256, 3, 1, 1, 1), ("32->64 k4 s2 @129 N16", 16, 129, 32, 64, 4, 2, 2, 0),
         ("64->128 k4 s2 @65 N16", 16, 65, 64, 128, 4, 2, 2, 0), ("128->256 k4 s1 @33 N16", 16, 33, 128, 256, 4, 1, 2, 0),
         ("4->64 k3 @256^2 N8", 8, 256, 4, 64, 3, 1, 1, 0), ("256->128 k3 @128^2 N8", 8, 128, 256, 128, 3, 1, 1, 0)]
for name, n, h, ci, co, k, s, p, ups in cases:
    x = torch.randn(n, h, h, ci, generator=g).cuda(); w = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).cuda()
    geom = L.geom_fwd(n, h, h, ci, co, k, s, p, ups)
    wp = ops._pack_fwd(w, ci, geom.korder)
    res = {}
    for mode, thr in (("f32", 0.0), ("f16x2", 1.0)):
        use_plan(conv_f16x2_min_flop=thr)
        y = ops.conv_raw(x, wp, geom)
        res[mode] = (y, timeit(lambda: ops.conv_raw(x, wp, geom)))
    xr = x.permute(0, 3, 1, 2).double().cpu()
    if ups: xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    ref = F.conv2d(xr, w.double().cpu(), None, stride=s, padding=p).permute(0, 2, 3, 1) if n * h * h * ci * co * k * k < 3e11 else None
    fl = ops._flops(geom)
    msg = "%-30s f32 %.3f ms (%5.1f TF/s) | f16x2 %.3f ms (%5.1f TF/s incl. 2 absmax)" % (name, res["f32"][1], fl / res["f32"][1] / 1e9, res["f16x2"][1], fl / res["f16x2"][1] / 1e9)
    if ref is not None:
        e = [float((res[m][0][..., :co].double().cpu() - ref).norm() / ref.norm()) for m in ("f32", "f16x2")]
        msg += " | err vs f64: f32 %.1e f16x2 %.1e" % tuple(e)
    else:
        msg += " | f16x2 vs f32 %.1e" % float((res["f16x2"][0] - res["f32"][0]).norm() / res["f32"][0].norm())
    print(msg)
