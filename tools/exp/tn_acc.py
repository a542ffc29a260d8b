import sys, torch
sys.path.insert(0, ".")
from deepsee_amd import lib as L
def amax(v):
    t = torch.zeros(64 * 32, device="cuda"); t[0] = v; return t
g = torch.Generator().manual_seed(3)
for (groups, t, rp, rq, splits) in [(2, 8192, 256, 256, 2), (2, 8192, 256, 256, 8), (1, 32768, 256, 256, 8), (1, 32768, 256, 256, 32)]:
    # gradient-like data: positive-mean so that sums grow (worst case for a long fp32 chain)
    for name, off in (("zero-mean", 0.0), ("mean 0.5", 0.5)):
        p = (torch.randn(groups * t, rp, generator=g) + off).cuda()
        q = (torch.randn(groups * t, rq, generator=g) + off).cuda()
        c = torch.empty(groups * splits, rp, rq, device="cuda")
        L.call("gemm_f16x2_tn_f32", p, q, c, groups, t, rp, rq, rq, splits, amax(float(p.abs().max())), amax(float(q.abs().max())))
        ts = t // splits
        ref = torch.einsum("ztp,ztq->zpq", p.view(groups * splits, ts, rp).double(), q.view(groups * splits, ts, rq).double())
        f32 = torch.einsum("ztp,ztq->zpq", p.view(groups * splits, ts, rp), q.view(groups * splits, ts, rq))
        e = ((c.double() - ref).norm() / ref.norm()).item(); e32 = ((f32.double() - ref).norm() / ref.norm()).item()
        # summed over splits (what the finalize kernel does)
        es = ((c.double().view(groups, splits, rp, rq).sum(1) - ref.view(groups, splits, rp, rq).sum(1)).norm() / ref.view(groups, splits, rp, rq).sum(1).norm()).item()
        print("K/split %5d %-9s: hip %.2e  (summed over splits %.2e) | torch fp32 einsum on GPU %.2e" % (ts, name, e, es, e32))
