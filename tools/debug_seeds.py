import sys, random
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_gpu_model as T
base = dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8)
for seed in (7, 117, 1, 2, 3, 4):
    orc, tm, out = T.run_case(base, seed=seed)
    r = out[0]
    gmax = max(float(v.norm()) for v in r["ggrads"].values())
    errs = sorted(((float((r["hg"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax), k) for k, v in r["ggrads"].items()), reverse=True)
    eerr = [e for e in errs if e[1].startswith("E.")]
    print(seed, "full" if orc.last_encoded_style_is_full else "mini", "noisy" if orc.last_encoded_style_is_noisy else "clean",
          "fake rel %.2e" % T.rel(r["hfake"], r["fake"]), "median %.2e max %.2e %s | E max %.2e %s" % (errs[len(errs)//2][0], errs[0][0], errs[0][1], eerr[0][0], eerr[0][1]))
