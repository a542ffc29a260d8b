#!/bin/bash
# round 5, GPU batch 10: native backtrace of the graph-replay segfault (LD_PRELOAD tools/exp/segv_bt.so) + which file triggers it
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== coarse entry"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "coarse" 2>&1 | grep -E "^E|assert|Error" | head -20
echo "== conv tests + training loop"
( time LD_PRELOAD=$PWD/tools/exp/segv_bt.so timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -q -x --durations=12 -k "test_gpu_conv or training_loop" 2>&1 | grep -v "^  File" | tail -60 ) 2>&1
echo "== test_gpu_model.py alone"
( time LD_PRELOAD=$PWD/tools/exp/segv_bt.so timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -x --durations=25 2>&1 | grep -v "^  File" | tail -90 ) 2>&1
} > gpurun_out/r05_batch10.txt 2>&1
tail -150 gpurun_out/r05_batch10.txt
