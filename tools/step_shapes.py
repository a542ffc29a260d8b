"""Per-geometry time of the non-Winograd MFMA conv launches in one G+D step (bs=8, 8x 32->256)."""
import os, sys, random, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsee_amd import ops
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt
from bench import synthetic_batch
opt = make_opt("independent_8x_256", batchSize=8, seed=0, hip_graphs=False)   # eager: the timers are events between launches
random.seed(1234)
tm = TrainerManager(opt)
batch = synthetic_batch(opt, 8, 1234, "cuda")
def step():
    tm.run_generator_one_step(batch); tm.run_discriminator_one_step(batch)
step(); step()
orig = ops._variant
def det(geom, modulate=False):
    return "%s N%d %dx%d C%d -> %dx%d C%d k%d mul%d ds%d ups%d" % (orig(geom, modulate), geom.N, geom.Hi, geom.Wi, geom.Cin, geom.Ho, geom.Wo, geom.Cout, geom.KH, geom.mul, geom.dshift, geom.ups)
ops._variant = det
_wr = ops.wgrad_raw
def wr(x, dout, geom, cout, cin, kh, kw, cin_first=0, **kw_):
    ops._wg = "wgrad N%d %dx%d C%d -> %dx%d C%d k%d mul%d" % (geom.N, geom.Hi, geom.Wi, geom.Cin, geom.Ho, geom.Wo, geom.Cout, geom.KH, geom.mul)
    return _wr(x, dout, geom, cout, cin, kh, kw, cin_first, **kw_)
ops.wgrad_raw = wr
_t = ops._timed
class T(_t):
    def __init__(self, name, flops, *rest):
        if name.startswith("conv_wgrad"): name = getattr(ops, "_wg", name)
        if name.startswith("winograd"): name = "%s %.3f TFLOP" % (name, flops / 1e12)
        super().__init__(name, flops, *rest)
ops._timed = T
ops.PROFILE = {}
step(); torch.cuda.synchronize()
rows = []
for k, v in ops.PROFILE.items():
    ms = sum(s.elapsed_time(e) for s, e, _ in v); fl = sum(f for _, _, f in v)
    rows.append((ms, len(v), fl / ms / 1e9, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total %.1f ms" % tot)
for ms, n, tf, k in rows[:45]: print("%7.3f ms  x%-3d %6.1f TF/s  %s" % (ms, n, tf, k))
