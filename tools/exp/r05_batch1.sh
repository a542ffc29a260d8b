#!/bin/bash
# round 5, GPU batch 1: fused-kernel stamps + ablations, branch-stream A/B, the tests the branches touch
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
DSEE_LIB=tools/exp/libfabl_32.so timeout 300 python tools/exp/fused_phases.py
timeout 300 python tools/exp/fused_kernel_bench.py
for m in 8 16 64; do DSEE_LIB=tools/exp/libfabl_$m.so timeout 300 python tools/exp/fused_kernel_bench.py; done
} > gpurun_out/r05_fused_phases.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run > gpurun_out/r05_ab_branches_on.json 2> gpurun_out/r05_ab_branches_on.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run --plan branch_streams=False > gpurun_out/r05_ab_branches_off.json 2> gpurun_out/r05_ab_branches_off.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run --dtype fp16 > gpurun_out/r05_ab_branches_on_fp16.json 2>> gpurun_out/r05_ab_branches_on.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run --dtype fp16 --plan branch_streams=False > gpurun_out/r05_ab_branches_off_fp16.json 2>> gpurun_out/r05_ab_branches_off.err
for f in gpurun_out/r05_ab_branches_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],2), "img/s", round(d["ms_per_step"],2), "ms")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
python -m pytest tests/test_gpu_model.py -m gpu -q -rA --durations=15 -k "partial_batches or hip_graphs or kernel_path or train_step_matches_oracle or dp_collect or half_mode_vs_oracle or training_loop" > gpurun_out/r05_subset2.log 2>&1
tail -30 gpurun_out/r05_subset2.log
