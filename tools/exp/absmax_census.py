"""Which tensors still get a stand-alone absmax pass in one G+D step (bs = 8, 8x 32->256)?  Groups by (elements, caller)."""
import os, sys, random, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import ops, lib as L
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt
from bench import synthetic_batch
opt = make_opt("independent_8x_256", batchSize=8, seed=0)
random.seed(1234)
tm = TrainerManager(opt)
batch = synthetic_batch(opt, 8, 1234, "cuda")
def step():
    tm.run_generator_one_step(batch); tm.run_discriminator_one_step(batch)
step()
seen = collections.Counter()
orig = L.call
def call(name, *a):
    if name == "absmax":
        st = traceback.extract_stack(limit=8)
        who = " < ".join("%s:%d" % (f.name, f.lineno) for f in reversed(st[:-1]) if "ops.py" in f.filename or "networks.py" in f.filename)[:110]
        seen[(int(a[1]), who)] += 1
    return orig(name, *a)
L.call = call
ops.L.call = call
step()
torch.cuda.synchronize()
tot = 0
for (n, who), c in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[1]):
    print("%3d x %12d elements (%7.1f MB)  %s" % (c, n, n * 4 / 1e6, who)); tot += c * n * 4
print("total", sum(seen.values()), "passes,", round(tot / 1e9, 2), "GB read")
