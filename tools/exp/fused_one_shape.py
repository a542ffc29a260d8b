"""The fused SPADE kernel at ONE shape (N = 8, 256^2, C = 512, K = 160, per-image tables, scale + mask written), a few launches: the
workload of the FETCH_SIZE passes of tools/exp/r05_batch20.sh (DSEE_LIB selects a set-shape variant build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fused_kernel_bench as B
print("%s: %.3f ms" % (os.environ.get("DSEE_LIB", "shipped"), B.bench(8, 256, 512, 160, True, reps=6, mask=True)))
