"""Times dsee_spade_fused_fwd alone (HIP events) at the benchmark shape; DSEE_LIB selects an ablation build; --packed times the
16-bit storage mode's dsee_spade_fused_fwd_f16p (one-term operands, fp16 scale)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import ops, lib as L

def bench(n, h, c, ld, per_image, reps=20, scale=True, mask=False, packed=False):
    rows = 2 * c
    g = torch.Generator(device="cuda").manual_seed(1)
    cat = torch.rand(n, h, h, ld, device="cuda", generator=g)
    x = torch.randn(n, h, h, c, device="cuda", generator=g)
    mean, invstd = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    b2 = torch.zeros(rows, device="cuda")
    w2a = torch.randn(rows, 128, 3, 3, device="cuda", generator=g) * 0.05
    t = n * (h // 4) ** 2
    ac = ops.tensor_amax(cat)
    terms, sp = (1, 4) if packed else (2, 2)
    v2 = ops._i16(36 * t * ld * terms)
    L.call("wino43_input_f16p" if packed else "wino43_input_f16x2", cat, v2, n, h, h, ld, ac, 100.0)
    if per_image:
        tb = torch.randn(n, 9, rows, 32, device="cuda", generator=g) * 0.05
        ua = ops.weight_amax(w2a, tb)
        u = ops._i16(36 * n * rows * ld * terms)
        L.call("wino43_weights_table", w2a, tb, u, n, rows, 128, sp, ua)
    else:
        u, ua = ops._wino_u(w2a, rows, 128, False, rows, ld, sp)
    out, sc = torch.empty_like(x), torch.empty(x.shape, dtype=torch.float16 if packed else torch.float32, device="cuda")
    mk = torch.empty(n * h * h * (c // 32), dtype=torch.int32, device="cuda") if mask else None
    ts = []
    for i in range(reps + 2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        L.call("spade_fused_fwd_f16p" if packed else os.environ.get("FUSED_ENTRY", "spade_fused_fwd"), v2, u, ac, 100.0, ua, b2, x, mean, invstd, out, sc if scale else None, n, h, h, c, rows, ld,
               n if per_image else 1, 1.0, 0.2, None, None, mk)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts[2:])[reps // 2]

if __name__ == "__main__":
    tag = os.environ.get("DSEE_LIB", "shipped") + " " + os.environ.get("FUSED_ENTRY", "spade_fused_fwd")
    pk = "--packed" in sys.argv
    if pk:
        tag += " f16p"
    bench(8, 256, 512, 160, True, reps=5)       # (clocks / caches warm before the first reported number)
    for (n, h, c, ld, pi) in [(8, 256, 512, 160, True), (8, 256, 512, 128, False), (8, 128, 512, 160, True)]:
        print("%s: N=%d %dx%d C=%d K=%d per_image=%d: %.3f ms (scale + sign mask written), %.3f ms (scale written), %.3f ms (neither)" % (
            tag, n, h, h, c, ld, pi, bench(n, h, c, ld, pi, mask=True, packed=pk),
            bench(n, h, c, ld, pi, packed=pk), bench(n, h, c, ld, pi, scale=False, packed=pk)), flush=True)
