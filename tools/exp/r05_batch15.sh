#!/bin/bash
# round 5, GPU batch 15: the final tree once more as the driver runs it (suite, smoke, bench)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/r05_durations_final.txt
( time DSEE_TEST_DURATIONS=gpurun_out/r05_durations_final.txt timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r05_gpu_tests_final.log 2>&1
grep -v "^  File" gpurun_out/r05_gpu_tests_final.log | tail -8 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['spade_fused']['ms_per_step'])"
