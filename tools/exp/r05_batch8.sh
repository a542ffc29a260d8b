#!/bin/bash
# round 5, GPU batch 8: (A) the tail of test_gpu_model.py with the managers kept alive (batch 6's state growth), (C) the whole suite
# with the per-test release fixture
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== A: RCCL / half-mode / switches / loop tests in one process, managers kept"
( time DSEE_TEST_KEEP_MANAGERS=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_model.py -m gpu -q -x -k "world1 or two_gpu or half_mode or kernel_path or training_loop" 2>&1 | grep -v "^  File" | tail -8 ) 2>&1
} > gpurun_out/r05_segv2.txt 2>&1
cat gpurun_out/r05_segv2.txt
( time timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x --durations=30 ) > gpurun_out/r05_gpu_tests.log 2>&1
grep -v "^  File" gpurun_out/r05_gpu_tests.log | tail -60
