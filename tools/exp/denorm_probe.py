"""Does the fp16 path keep subnormals? (1) v_cvt_f16_f32 of 2^-20; (2) fp16 MFMA (hipBLASLt matmul) on subnormal inputs;
(3) our fp16x2 GEMM on a row 2^-20 below the maximum: relative error of that row."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
x = torch.full((4,), 2.0 ** -20, device="cuda")
print("cvt f32->f16 of 2^-20:", x.half().float().tolist()[0] * 2 ** 20)
a = torch.full((128, 128), 2.0 ** -20, device="cuda", dtype=torch.half)
b = torch.full((128, 128), 1024.0, device="cuda", dtype=torch.half)
print("half matmul (subnormal a) expect %.4g got %.4g" % (128 * 2.0 ** -10, float((a @ b)[0, 0])))
from deepsee_amd import lib as L
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from test_gpu_conv import _amax, _pow2_scale, _split2_rows
g = torch.Generator().manual_seed(1)
n = k = 128
bm = torch.randn(n, k, generator=g)
am = torch.randn(128, k, generator=g)
for sh in (0, -4, -8, -12, -16, -20, -24):
    a2 = am.clone(); a2[5] *= 2.0 ** sh
    ref = a2.double() @ bm.double().t()
    c = torch.full((128, n), float("nan"), device="cuda")
    L.call("gemm_f16x2_af32", a2.cuda(), _split2_rows(bm, _pow2_scale(float(bm.abs().max())))[None].cuda(), c, 128, n, k, 128, n, 1,
           _amax(float(a2.abs().max())), _amax(float(bm.abs().max())))
    torch.cuda.synchronize()
    e = float((c[5].cpu().double() - ref[5]).norm() / ref[5].norm())
    print("A row 2^%d below max: rel err of that row %.2e" % (sh, e))
for sh in (0, -8, -16, -20):
    b2 = bm.clone(); b2[5] *= 2.0 ** sh
    ref = am.double() @ b2.double().t()
    c = torch.full((128, n), float("nan"), device="cuda")
    L.call("gemm_f16x2_af32", am.cuda(), _split2_rows(b2, _pow2_scale(float(b2.abs().max())))[None].cuda(), c, 128, n, k, 128, n, 1,
           _amax(float(am.abs().max())), _amax(float(b2.abs().max())))
    torch.cuda.synchronize()
    e = float((c[:, 5].cpu().double() - ref[:, 5]).norm() / ref[:, 5].norm())
    print("B row 2^%d below max (host split, subnormal lo kept): rel err of that column %.2e" % (sh, e))
