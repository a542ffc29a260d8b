"""C-ABI entry-point calls of one eager G + D step, by name (deepsee_amd.lib.CALLS)."""
import os, sys, random, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import lib as L
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt
from bench import synthetic_batch
import ast
plan = {}
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1); plan[k] = ast.literal_eval(v)
opt = make_opt("independent_8x_256", batchSize=8, seed=0, hip_graphs=False, kernel_plan=plan)
random.seed(1234)
tm = TrainerManager(opt)
batch = synthetic_batch(opt, 8, 1234, "cuda")
def step():
    tm.run_generator_one_step(batch); tm.run_discriminator_one_step(batch)
step(); step(); torch.cuda.synchronize()
names = collections.Counter()
orig = L.call
sizes = collections.Counter()
import traceback
def counted(name, *a, **k):
    names[name] += 1
    if name == "absmax":
        fr = [f for f in traceback.extract_stack() if "deepsee_amd" in f.filename][-3:]
        sizes[(int(a[1]), " < ".join("%s:%d" % (f.name, f.lineno) for f in reversed(fr)))] += 1
    return orig(name, *a, **k)
L.call = counted
import deepsee_amd.ops as ops
step(); torch.cuda.synchronize()
print("plan", plan, "entry-point calls per step:", sum(names.values()))
for n, c in names.most_common(25): print("%4d  %s" % (c, n))
for (n, where), c in sorted(sizes.items(), key=lambda kv: -kv[0][0]): print("absmax x%d  %10d floats  %s" % (c, n, where))
