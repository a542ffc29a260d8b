#!/bin/bash
# round 5, GPU batch 7: the segfault of test_training_loop_with_device_loader_and_metrics inside graph.replay() (batch 6, full suite)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
T=tests/test_gpu_model.py
{
for i in 1 2 3; do
  echo "== alone $i"; timeout 300 python -X faulthandler -m pytest $T -m gpu -q -x -k "test_training_loop_with_device_loader_and_metrics" 2>&1 | grep -v "^  File" | tail -5
done
echo "== with the tests in front of it"
timeout 900 python -X faulthandler -m pytest $T -m gpu -q -x -k "test_half_mode_vs_oracle or test_kernel_path_switches or test_training_loop_with_device_loader_and_metrics" 2>&1 | grep -v "^  File" | tail -5
echo "== under rocgdb (sequence)"
timeout 1200 rocgdb -batch -ex "handle SIGSEGV stop print" -ex "handle SIG35 nostop noprint pass" -ex run -ex bt -ex "info threads" --args python -m pytest $T -m gpu -q -x -k "test_kernel_path_switches or test_training_loop_with_device_loader_and_metrics" 2>&1 | grep -v "^\[New Thread\|^\[Thread.*exited" | tail -80
echo "== rest of the suite"
( time timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --durations=15 -k "test_vgg_weights_option or test_gpu_ops" ) 2>&1 | tail -30
} > gpurun_out/r05_segv.txt 2>&1
tail -150 gpurun_out/r05_segv.txt
