#!/bin/bash
# usage (inside gpurun): bash tools/exp/ab_plan.sh FIELD=VALUE [bench args]  -- alternates the default plan and the plan with FIELD=VALUE
alt=$1; shift
for i in 1 2 3; do
  for side in default "$alt"; do
    if [ "$side" = default ]; then extra=""; else extra="--plan $side"; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run $extra "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$side', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
