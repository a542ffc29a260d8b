import sys, random
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_gpu_model as T
from oracle import deepsee_oracle as O
base = dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8)
for extra in (dict(no_vgg_loss=True, no_ganFeat_loss=True), dict(no_vgg_loss=True), dict(no_ganFeat_loss=True), dict()):
    over = dict(base, **extra)
    orc, tm, out = T.run_case(over, seed=117)
    r = out[0]
    gmax = max(float(v.norm()) for v in r["ggrads"].values())
    errs = sorted(((float((r["hg"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax), k) for k, v in r["ggrads"].items()), reverse=True)
    print(extra, "fake rel %.2e" % T.rel(r["hfake"], r["fake"]), "median G err %.3e max %.3e %s" % (errs[len(errs)//2][0], errs[0][0], errs[0][1]))
