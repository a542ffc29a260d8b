"""Where the ATen glue launches of one eager G + D step come from: torch.profiler with Python stacks, kernels not from the
dsee library grouped by (kernel family, innermost deepsee_amd frame)."""
import os, sys, random, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt
from bench import synthetic_batch
from torch.profiler import profile, ProfilerActivity
opt = make_opt("independent_8x_256", batchSize=8, seed=0, hip_graphs=False)
random.seed(1234)
tm = TrainerManager(opt)
batch = synthetic_batch(opt, 8, 1234, "cuda")
def step():
    tm.run_generator_one_step(batch); tm.run_discriminator_one_step(batch)
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = collections.Counter(); tus = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    nk = len(e.kernels) + sum(len(c.kernels) for c in e.cpu_children)
    dev = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    if dev <= 0:
        continue
    fr = [s for s in (e.stack or []) if "deepsee_amd" in s or "bench.py" in s]
    where = fr[0].split("deepsee_amd/")[-1] if fr else "(autograd engine / no python frame)"
    rows[(e.name, where)] += 1; tus[(e.name, where)] += dev
tot = sum(tus.values())
print("ATen ops with device time in one step: %d ops, %.3f ms" % (sum(rows.values()), tot / 1e3))
for k, us in tus.most_common(60):
    print("%4d x  %8.1f us  %-28s %s" % (rows[k], us, k[0], k[1]))
