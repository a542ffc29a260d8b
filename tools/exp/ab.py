"""Same-box A/B runs of bench.py with fields of the model's KernelPlan flipped (deepsee_amd/plan.py):
    python tools/exp/ab.py thin_gemm=False presplit_a=False -- <bench args>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
sys.argv = ["bench.py"] + args[cut + 1:] + (["--plan"] + args[:cut] if cut else [])
import bench
bench.main()
