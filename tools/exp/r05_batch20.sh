#!/bin/bash
# round 5, GPU batch 20: the co-running set's shape (A tile groups x 32/A row groups per XCD) against the traffic model of DESIGN 3.12:
# fetch = 8192 x 1.47 MB x (1/A + A/32) + 1.07 GB of x.  FETCH_SIZE (KB, x2 on gfx950) per launch + stand-alone time, same box.
R=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp; export TMPDIR=/tmp
for v in shipped seta2 seta8 seta16; do
  if [ $v = shipped ]; then unset DSEE_LIB; else export DSEE_LIB=$R/tools/exp/libfvar_$v.so; fi
  t=$(timeout 120 python $R/tools/exp/fused_one_shape.py 2>/dev/null | tail -1)
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/ps_$v -o p --output-format csv -- python $R/tools/exp/fused_one_shape.py > /dev/null 2>&1
  f=$(python - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for p in glob.glob("/tmp/ps_$v/*counter_collection.csv") for r in csv.DictReader(open(p)) if r["Counter_Name"]=="FETCH_SIZE" and "spade_fused_fwd_kernel<5" in r["Kernel_Name"]]
print("FETCH_SIZE x2: %.2f GB per launch (%d launches)" % (2*sum(v)*1024/len(v)/1e9, len(v)))
PY
)
  echo "$t | $f"
done
