"""Pre-split NT GEMMs (dsee_gemm_f16x2_pre / dsee_gemm_f16p_pre) at the step's shapes: the 8-wave ping-pong kernel against the
one-wave-per-SIMD kernel (csrc/gemm_w4.hip).  DSEE_LIB selects a measurement build (tools/exp/build_abl.sh); with
`--stamps` the 8-wave kernel's cycle stamps (DSEE_GEMM_ABL & 32 build) are printed instead of timings."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deepsee_amd import lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--entries", default="gemm_f16x2_pre,gemm_f16x2_pre_w4")
ap.add_argument("--shapes", default="256,128,64")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--tag", default="")
args = ap.parse_args()

def timeit(f, reps):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

am = torch.zeros(64 * 32, device="cuda"); am[0] = 4.0
for res in [int(x) for x in args.shapes.split(",")]:
    t, k, n = 8 * (res // 4) ** 2, 512, 512
    M = 36 * t
    # (realistic bit patterns: the clock follows the data)
    a2 = (torch.randn(M * k * 2, device="cuda") * 1000).half().view(torch.int16)
    b2 = (torch.randn(36 * n * k * 2, device="cuda") * 1000).half().view(torch.int16)
    fl = 2.0 * M * n * k
    for e in args.entries.split(","):
        pk = "f16p" in e
        c = torch.empty(M, n, device="cuda", dtype=torch.float16 if pk else torch.float32)
        cs = torch.zeros(64 * 32, device="cuda")
        extra = (cs,) if pk else ()
        run = lambda: L.call(e, a2, b2, c, M, n, k, t, n, am, 100.0, am, *extra)
        if args.stamps and e.endswith("_w4"):
            run(); torch.cuda.synchronize()
            st = c.reshape(-1)[:8 * 4 * 8].reshape(8, 4, 8).cpu()
            slabs = (M // 256) * (n // 256) * (k // 16) / 256.0
            for w in range(4):
                v = st[0, w] / slabs
                print("%s 512->512 @%d^2 wave %d cycles per slab: body %.0f | vmcnt wait %.0f | barrier %.0f | tile stores (amortised) %.0f | sum %.0f"
                      % (args.tag, res, w, v[0], v[1], v[2], v[3], float(v[:4].sum())))
            continue
        if args.stamps:
            run(); torch.cuda.synchronize()
            st = c.reshape(-1)[:64 * 8].reshape(8, 8, 8).cpu()
            slabs = (M // 256) * (n // 256) * (k // 16) / 256.0
            for w in (0, 4):
                v = st[0, w] / slabs
                print("%s 512->512 @%d^2 wave %d cycles per slab: frag reads %.0f | DMA issue %.0f | wait %.0f | barrier1 %.0f | "
                      "MFMA(+stores) %.0f | barrier2 %.0f | sum %.0f" % (args.tag, res, w, v[0], v[1], v[2], v[3], v[4], v[5], float(v[:6].sum())))
            continue
        ms = timeit(run, args.reps)
        peak = 2516.6 if pk else 838.9
        print("%s %-22s 512->512 @%d^2 (M=%d): %.3f ms  %.0f TF/s (%.3f of %.0f)  A+C %.2f TB/s"
              % (args.tag, e, res, M, ms, fl / ms / 1e9, fl / ms / 1e9 / peak, peak,
                 (M * k * (2 if pk else 4) + M * n * (2 if pk else 4)) / ms / 1e9))
