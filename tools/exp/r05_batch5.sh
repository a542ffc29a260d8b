#!/bin/bash
# round 5, GPU batch 5: schedule variants of the fused kernel (same box), its tests, the restored smooth-loss cases
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "spade_fused or fused_spade" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "norm or resblock" 2>&1 | tail -2
{
timeout 300 python tools/exp/fused_kernel_bench.py
for v in sched1 sched2 sched3 sched4 sched5; do DSEE_LIB=tools/exp/libfvar_$v.so timeout 300 python tools/exp/fused_kernel_bench.py; done
timeout 300 python tools/exp/fused_kernel_bench.py
timeout 300 python tools/exp/fused_kernel_bench.py --packed
for v in sched1 sched3 sched5; do DSEE_LIB=tools/exp/libfvar_$v.so timeout 300 python tools/exp/fused_kernel_bench.py --packed; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_fused_sched_variants.txt
cat gpurun_out/r05_fused_sched_variants.txt
( timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -s --durations=5 -k "test_full_size_smooth_loss_backward and not config1" 2>&1 | grep -E "fake deviation|passed|failed|Error|s call" ) > gpurun_out/r05_smooth_cases.txt 2>&1
cat gpurun_out/r05_smooth_cases.txt
