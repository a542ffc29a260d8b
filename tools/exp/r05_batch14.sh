#!/bin/bash
# round 5, GPU batch 14: does keeping the captured hipGraph_t alive (torch.cuda.CUDAGraph(keep_graph=True)) avoid the replay segfault of
# graphs with parallel branches?  tests/test_gpu_model.py whole, plan.branch_streams = True, as in batches 10 / 11 (deterministic crash there)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/r05_durations_kg.txt
( time DSEE_PLAN=branch_streams=True DSEE_KEEP_GRAPH=1 DSEE_TEST_DURATIONS=gpurun_out/r05_durations_kg.txt timeout 1500 python -X faulthandler -m pytest tests/test_gpu_model.py -x -q -m gpu ) > gpurun_out/r05_keepgraph.log 2>&1
grep -v "^  File" gpurun_out/r05_keepgraph.log | tail -12 | cut -c1-300
