#!/bin/bash
# round 5, GPU batch 6: the whole -m gpu suite (wall time, slowest tests), then headline bench + rocprof + PMC of the current tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 ) > gpurun_out/r05_gpu_tests.log 2>&1
tail -40 gpurun_out/r05_gpu_tests.log
bash tools/exp/refresh_evidence.sh 2>&1 | tail -5
