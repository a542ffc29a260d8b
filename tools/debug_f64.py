import sys, random
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_gpu_model as T
from oracle import deepsee_oracle as O
base = dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8, no_ganFeat_loss=True)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 117
orc, tm, out = T.run_case(base, seed=seed)
r = out[0]
# f64 yardstick on the same tape
oopt = O.make_opt(**base)
states = O.recipe_state(oopt, gain=1.0)
batch = O.synthetic_batch(oopt, 2, seed=seed)
o64 = O.Oracle(oopt, states, O.ReplayCtl(orc.ctl.tape), dtype=torch.float64)
gl, fake = o64.run_generator_one_step({k: v.clone() for k, v in batch.items()})
g64 = {"%s.%s" % (net, k): p.grad.clone() for net in ("SR", "E") for k, p in o64.params(net) if p.grad is not None}
gmax = max(float(v.norm()) for v in g64.values())
def err(d):
    return sorted(((float((d[k].double() - v).norm()) / max(float(v.norm()), 1e-3 * gmax), k) for k, v in g64.items()), reverse=True)
eh, ec = err(r["hg"]), err(r["ggrads"])
print("seed", seed, "fake: hip-vs-f64 %.2e cpu32-vs-f64 %.2e" % (T.rel(r["hfake"], fake.detach()), T.rel(r["fake"], fake.detach())))
print("HIP   vs f64: median %.2e max %.2e %s" % (eh[len(eh)//2][0], eh[0][0], eh[0][1]))
print("CPU32 vs f64: median %.2e max %.2e %s" % (ec[len(ec)//2][0], ec[0][0], ec[0][1]))
