import sys, random
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, torch.nn.functional as F
import test_gpu_model as T
from oracle import deepsee_oracle as O
from deepsee_amd import networks as N
base = dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8, no_ganFeat_loss=True)
for tap in range(5):
    ws = [0.0] * 5; ws[tap] = 1.0
    N.VGG_WEIGHTS[:] = ws
    def vgg_loss(self, x, y, ws=ws):
        fx, fy = self.vgg_features(x), self.vgg_features(y)
        loss = 0
        for w, a, b in zip(ws, fx, fy):
            loss = loss + w * F.l1_loss(a, b.detach())
        return loss
    O.Oracle.vgg_loss = vgg_loss
    orc, tm, out = T.run_case(base, seed=117)
    r = out[0]
    gmax = max(float(v.norm()) for v in r["ggrads"].values())
    errs = sorted(((float((r["hg"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax), k) for k, v in r["ggrads"].items()), reverse=True)
    print("tap", tap, "VGG loss", r["gl"]["VGG"], r["hgl"]["VGG"], "median G err %.3e max %.3e" % (errs[len(errs)//2][0], errs[0][0]))
