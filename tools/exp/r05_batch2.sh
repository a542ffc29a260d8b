#!/bin/bash
# round 5, GPU batch 2 (after the container restart): full -m gpu suite with durations, branch-stream A/B, fused-kernel stamps
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 ) > gpurun_out/r05_gpu_tests.log 2>&1
tail -40 gpurun_out/r05_gpu_tests.log
{
DSEE_LIB=tools/exp/libfabl_32.so timeout 300 python tools/exp/fused_phases.py
timeout 300 python tools/exp/fused_kernel_bench.py
} > gpurun_out/r05_fused_phases.txt 2>&1
tail -20 gpurun_out/r05_fused_phases.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run > gpurun_out/r05_ab_branches_on.json 2> gpurun_out/r05_ab_branches_on.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run --plan branch_streams=False > gpurun_out/r05_ab_branches_off.json 2> gpurun_out/r05_ab_branches_off.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run --dtype fp16 > gpurun_out/r05_ab_branches_on_fp16.json 2>> gpurun_out/r05_ab_branches_on.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run --dtype fp16 --plan branch_streams=False > gpurun_out/r05_ab_branches_off_fp16.json 2>> gpurun_out/r05_ab_branches_off.err
tail -5 gpurun_out/r05_ab_branches_on.err gpurun_out/r05_ab_branches_off.err
for f in gpurun_out/r05_ab_branches_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],2), "img/s", round(d["ms_per_step"],2), "ms", "norm_forward", d["roofline"].get("norm_forward",{}).get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
