"""mlp_shared's weight gradient in the SEAN table path: the generic MFMA weight gradient over the one-hot channels of `cat` (+ a
channel-dot for the bias) against dsee_onehot_conv3x3_wgrad (segmented sums by label, bias included)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import ops, lib as L
def timeit(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
n, nc, NH = 8, 19, 128
for r in (256, 128, 64, 32):
    lab = torch.randint(0, nc, (n, 256, 256), dtype=torch.uint8, device="cuda")
    shift = {256: 0, 128: 1, 64: 2, 32: 3}[r]
    ld = 160
    cat = torch.randn(n, r, r, ld, device="cuda")
    L.call("label_onehot", lab, cat, n, 256, 256, shift, ld, NH)
    dactv = torch.randn(n, r, r, NH, device="cuda") * (cat[..., :NH] > 0)
    gs = L.geom_fwd(n, r, r, ld, NH, 3, 1, 1, 0)
    def mfma():
        dw = ops.wgrad_raw(cat, dactv, gs, NH, nc, 3, 3, cin_first=NH)
        db = ops.channel_dot(dactv, None, NH)
        return dw, db
    dw2, db2 = ops.new(NH, nc, 3, 3), ops.new(NH)
    ws = ops.scratch(L.lib().dsee_onehot_conv3x3_wgrad_workspace(n, 256, 256, shift, nc), "ohw")
    def onehot():
        L.call("onehot_conv3x3_wgrad", lab, dactv, NH, cat, ld, n, 256, 256, shift, nc, dw2, db2, ws)
    oh = ops.new(n, r, r, 32)
    g32 = L.geom_fwd(n, r, r, 32, NH, 3, 1, 1, 0)
    def compact():
        L.call("label_onehot", lab, oh, n, 256, 256, shift, 32, 0)
        dw = ops.wgrad_raw(oh, dactv, g32, NH, nc, 3, 3)
        db = ops.channel_dot(dactv, None, NH)
        return dw, db
    t3 = timeit(compact)
    dw3, _ = compact()
    print("%3d^2: MFMA wgrad over a compact 32-channel one-hot tensor (+ label_onehot + channel_dot) %.3f ms" % (r, t3))
    t1, t2 = timeit(mfma), timeit(onehot)
    assert float((dw3 - mfma()[0]).abs().max()) < 1e-3 * float(dw3.abs().max())
    dw, db = mfma(); onehot(); torch.cuda.synchronize()
    print("%3d^2: MFMA wgrad + channel_dot %.3f ms | one-hot kernel %.3f ms | dw rel diff %.1e, db rel diff %.1e"
          % (r, t1, t2, float((dw - dw2).norm() / dw2.norm()), float((db - db2).norm() / db2.norm())))
