#!/bin/bash
# round 5, GPU batch 12: the whole -m gpu suite as the driver runs it (per-test durations on the side), then the round's evidence
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/r05_durations.txt
( time DSEE_TEST_DURATIONS=gpurun_out/r05_durations.txt timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r05_gpu_tests.log 2>&1
grep -v "^  File" gpurun_out/r05_gpu_tests.log | tail -15
sort -rn gpurun_out/r05_durations.txt | head -25
bash tools/collect_evidence_r05.sh 2>&1 | tail -3
