"""Weight-gradient TN GEMM at the conv 512x512 @256^2 bs=8 shape (36 groups, T = 32768 tiles, split-K 8): fp32 operands,
pre-split Q, both pre-split."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deepsee_amd import lib as L
g, t, rp, rq, splits = 36, 32768, 512, 512, 8
p = torch.randn(g * t, rp, device="cuda"); q = torch.randn(g * t, rq, device="cuda")
p2 = (torch.randn(g * t * rp * 2, device="cuda") * 1000).half().view(torch.int16)
q2 = (torch.randn(g * t * rq * 2, device="cuda") * 1000).half().view(torch.int16)
c = torch.empty(g * splits, rp, rq, device="cuda")
am = torch.zeros(64 * 32, device="cuda"); am[0] = 4.0
def timeit(f, reps=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl = 2.0 * g * t * rp * rq
for name, f in (("fp32 P, fp32 Q", lambda: L.call("gemm_f16x2_tn_f32", p, q, c, g, t, rp, rq, rq, splits, am, am)),
                ("fp32 P, pre-split Q", lambda: L.call("gemm_f16x2_tn_qpre", p, q2, c, g, t, rp, rq, rq, splits, am, am, 100.0)),
                ("both pre-split (ping-pong)", lambda: L.call("gemm_f16x2_tn_pqpre", p2, q2, c, g, t, rp, rq, rq, splits, am, 225.0, am, 100.0))):
    ms = timeit(f)
    print("TN 512x512 @256^2 %-28s %.3f ms  %.0f TF/s fp32-eq (%.2f of 839)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 839))
# the SEAN table weight gradient: 36 x 8 per-image groups, T/N = 4096 tiles, P = 1024 gamma/beta rows, Q = 160 columns
g, t, rp, rq, splits = 288, 4096, 1024, 160, 1
p = torch.randn(g * t, rp, device="cuda"); q = torch.randn(g * t, rq, device="cuda")
q2 = (torch.randn(g * t * rq * 2, device="cuda") * 1000).half().view(torch.int16)
c = torch.empty(g * splits, rp, rq, device="cuda")
fl = 2.0 * g * t * rp * rq
for name, f in (("fp32 P, fp32 Q", lambda: L.call("gemm_f16x2_tn_f32", p, q, c, g, t, rp, rq, rq, splits, am, am)),
                ("fp32 P, pre-split Q (ping-pong)", lambda: L.call("gemm_f16x2_tn_qpre", p, q2, c, g, t, rp, rq, rq, splits, am, am, 100.0))):
    ms = timeit(f)
    print("TN table 1024x160 @256^2 %-32s %.3f ms  %.0f TF/s fp32-eq (%.2f of 839)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 839))
