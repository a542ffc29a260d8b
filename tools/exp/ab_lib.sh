#!/bin/bash
# usage (inside gpurun): bash tools/exp/ab_lib.sh [bench args]   -- alternates the working-tree library and tools/exp/libhead.so
for i in 1 2 3; do
  for lib in new head; do
    if [ $lib = head ]; then export DSEE_LIB=tools/exp/libhead.so; else unset DSEE_LIB; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
