#!/bin/bash
# round 5, GPU batch 13: the coarse entry points' tests, smoke(), the 2-rank gloo bench on one device (dp path after this round's changes)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -s -k "coarse" 2>&1 | grep -E "^E|passed|failed|coarse resblock|Error" | head -20
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== 2 ranks, gloo, one device"
DSEE_DIST_BACKEND=gloo DSEE_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --no-f32-run 2>&1 | grep -v "Gloo\|amdgpu.ids" | tail -2 | cut -c1-1500
} > gpurun_out/r05_batch13.txt 2>&1
cat gpurun_out/r05_batch13.txt
