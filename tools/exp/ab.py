"""Same-box A/B runs of bench.py with module flags of deepsee_amd.ops flipped: python tools/exp/ab.py THIN_GEMM=False -- <bench args>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import ops
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
for kv in args[:cut]:
    k, v = kv.split("=")
    setattr(ops, k, eval(v))
sys.argv = ["bench.py"] + args[cut + 1:]
import bench
bench.main()
