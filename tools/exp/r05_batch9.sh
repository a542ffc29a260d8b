#!/bin/bash
# round 5, GPU batch 9: bisect the graph-replay segfault; coarse entry point test; cycle stamps of the fused kernel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== coarse entry"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "coarse" 2>&1 | tail -3
echo "== fused kernel stamps"
DSEE_LIB=tools/exp/libfabl_32.so timeout 300 python tools/exp/fused_phases.py 2>&1 | grep -v amdgpu.ids
echo "== churn keep"
timeout 600 python tools/exp/graph_churn.py 40 loader keep 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|self.sr_model" | tail -12
echo "== churn close"
timeout 600 python tools/exp/graph_churn.py 40 loader close 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|self.sr_model" | tail -6
echo "== benchmark_path + training_loop"
( time timeout 1200 python -X faulthandler -m pytest tests/test_gpu_model.py -m gpu -q -x -k "benchmark_path or training_loop" 2>&1 | grep -v "^  File" | tail -8 ) 2>&1
} > gpurun_out/r05_batch9.txt 2>&1
cat gpurun_out/r05_batch9.txt
