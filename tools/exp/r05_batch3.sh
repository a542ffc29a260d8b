#!/bin/bash
# round 5, GPU batch 3: LDS-DMA address-pattern probe, fused kernel with the coalescing-friendly chunk assignment, the failing
# smooth-loss case with / without branch streams, the tests batch 2 did not reach
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
PROBE_PATTERNS=1 timeout 300 tools/exp/ldsdma_probe > gpurun_out/r05_ldsdma_patterns.txt 2>&1
cat gpurun_out/r05_ldsdma_patterns.txt
timeout 300 python tools/exp/fused_kernel_bench.py > gpurun_out/r05_fused_bench_coalesced.txt 2>&1
cat gpurun_out/r05_fused_bench_coalesced.txt
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "spade_fused or fused_spade" 2>&1 | tail -3
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "test_full_size_smooth_loss_backward and guided" -s 2>&1 | grep -E "guided_32|passed|failed|Error" ) > gpurun_out/r05_guided_on.txt 2>&1
( DSEE_PLAN="branch_streams=False" timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "test_full_size_smooth_loss_backward and guided" -s 2>&1 | grep -E "guided_32|passed|failed|Error" ) > gpurun_out/r05_guided_off.txt 2>&1
head -5 gpurun_out/r05_guided_on.txt gpurun_out/r05_guided_off.txt
( time timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --durations=15 -k "not test_train_step_matches_oracle and not test_full_size_step_matches_oracle and not test_benchmark_path_matches_oracle and not test_smooth_loss_backward and not (test_full_size_smooth_loss_backward and not indep_16to512)" ) > gpurun_out/r05_gpu_tests_rest.log 2>&1
tail -30 gpurun_out/r05_gpu_tests_rest.log
