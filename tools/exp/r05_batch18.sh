#!/bin/bash
# round 5, GPU batch 18: threshold above which the direct convolutions take split fp16x2 operands (plan.conv_f16x2_min_flop), same box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for v in 1e9 3e8 1e8 3e9 1e9; do
  python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline --plan conv_f16x2_min_flop=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],2), round(d['ms_per_step'],2))"
done > gpurun_out/r05_conv_threshold.txt 2>&1
cat gpurun_out/r05_conv_threshold.txt
