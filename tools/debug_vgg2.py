import sys, random
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, torch.nn.functional as F
from oracle import deepsee_oracle as O
from deepsee_amd import networks as N, ops, lib as L
torch.manual_seed(0)
opt = O.make_opt(ngf=8)
st = O.recipe_state(opt)["VGG"]
vgg = N.VGG19Taps().cuda()
for k, v in vgg.state_dict().items(): v.copy_(st[k])
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = torch.Generator().manual_seed(seed)
x = (torch.rand(2, 3, 32, 32, generator=g) * 2 - 1)
y = (torch.rand(2, 3, 32, 32, generator=g) * 2 - 1)
def cpu_run(dt):
    xs = x.to(dt).requires_grad_()
    o = O.Oracle(opt, {"VGG": st}, dtype=dt)
    fx = o.vgg_features(xs); fy = o.vgg_features(y.to(dt))
    outs = []
    for i in range(5):
        l = F.l1_loss(fx[i], fy[i].detach())
        gi, = torch.autograd.grad(l, xs, retain_graph=True)
        outs.append(gi)
    return outs
c32, c64 = cpu_run(torch.float32), cpu_run(torch.float64)
xd = ops.to_nhwc(x.cuda()).requires_grad_()
fx = vgg(xd)
with torch.no_grad(): fy = vgg(ops.to_nhwc(y.cuda()))
for i in range(5):
    l = ops.mean_loss(fx[i], fy[i], ops.MODE_L1, 1.0)
    gi, = torch.autograd.grad(l, xd, retain_graph=True)
    gh = ops.to_nchw(gi, 3).cpu()
    e = lambda a, b: float((a.double() - b).norm() / b.norm())
    print("tap %d: hip-vs-f64 %.2e   cpu32-vs-f64 %.2e   feat rel %.2e" % (i, e(gh, c64[i]), e(c32[i], c64[i]), 0))
# locate sign mismatches at tap 4
o = O.Oracle(opt, {"VGG": st})
cx, cy = o.vgg_features(x)[4], o.vgg_features(y)[4]
o64 = O.Oracle(opt, {"VGG": st}, dtype=torch.float64)
dx, dy = o64.vgg_features(x.double())[4], o64.vgg_features(y.double())[4]
hx, hy = ops.to_nchw(fx[4].detach(), 512).cpu(), ops.to_nchw(fy[4], 512).cpu()
sc, sh, sd = torch.sign(cx - cy), torch.sign(hx - hy), torch.sign(dx - dy)
print("mismatch hip-vs-f64:", int((sh.double() != sd).sum()), " cpu-vs-f64:", int((sc.double() != sd).sum()), "of", sd.numel())
idx = (sh.double() != sd).nonzero()
for i in idx[:5]:
    i = tuple(int(v) for v in i)
    print(i, "hip a,b", float(hx[i]), float(hy[i]), "cpu a,b", float(cx[i]), float(cy[i]), "f64", float(dx[i]), float(dy[i]))
print("frac zero a", float((dx == 0).double().mean()), "max feat", float(dx.max()))
