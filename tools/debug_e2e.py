import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_gpu_model as T
name = sys.argv[1] if len(sys.argv) > 1 else "indep_4to32_ngf8"
orc, tm, out = T.run_case(T.CASES[name], seed=101 + len(name))
r = out[0]
print("G losses", r["gl"], r["hgl"])
print("fake rel", T.rel(r["hfake"], r["fake"]))
print("D losses", r["dl"], r["hdl"])
gmax = max(float(v.norm()) for v in r["ggrads"].values())
errs = sorted(((float((r["hg"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax), k, float(v.norm())) for k, v in r["ggrads"].items()), reverse=True)
for e in errs[:12]: print("G  %.3e  %-50s norm %.3e" % e)
print("median G err %.3e" % errs[len(errs)//2][0])
dmax = max(float(v.norm()) for v in r["dgrads"].values())
errs = sorted(((float((r["hd"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * dmax), k, float(v.norm())) for k, v in r["dgrads"].items()), reverse=True)
for e in errs[:6]: print("D  %.3e  %-50s norm %.3e" % e)
