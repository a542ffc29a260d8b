#!/bin/bash
# round 5, GPU batch 17: the final tree's -m gpu suite as the driver runs it (wall time after the second trim)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/r05_durations_last.txt
( time DSEE_TEST_DURATIONS=gpurun_out/r05_durations_last.txt timeout 1200 python -m pytest tests -x -q -m gpu ) > gpurun_out/r05_gpu_tests_last.log 2>&1
grep -v "^  File" gpurun_out/r05_gpu_tests_last.log | tail -8 | cut -c1-200
sort -rn gpurun_out/r05_durations_last.txt | head -8
