"""GPU time per autograd Function (forward / backward) in one G+D step (bs=8, 8x 32->256)."""
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
recs = collections.defaultdict(list)
def wrap(cls, name):
    f = getattr(cls, name)
    def timed(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = f(*a, **k); e.record()
        tag = cls.__name__ + "." + name
        if cls.__name__ == "Conv2d":
            x = a[1] if name == "forward" else a[0].saved_tensors[0]
            w = a[2] if name == "forward" else a[0].saved_tensors[1]
            tag += " %dx%d %d->%d k%d" % (x.shape[1], x.shape[2], w.shape[1], w.shape[0], w.shape[2])
        elif cls.__name__ == "SeanNormTable":
            x = a[1] if name == "forward" else a[0].saved_tensors[0]
            tag += " %dx%d C%d" % (x.shape[1], x.shape[2], x.shape[3])
        recs[tag].append((s, e)); return r
    setattr(cls, name, staticmethod(timed))
for nm in dir(ops):
    c = getattr(ops, nm)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, "forward"); wrap(c, "backward")
step(); torch.cuda.synchronize()
rows = sorted(((sum(s.elapsed_time(e) for s, e in v), len(v), k) for k, v in recs.items()), reverse=True)
print("total %.1f ms" % sum(r[0] for r in rows))
for ms, n, k in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]: print("%7.2f ms  x%-3d %s" % (ms, n, k))
