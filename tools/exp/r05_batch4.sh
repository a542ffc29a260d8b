#!/bin/bash
# round 5, GPU batch 4: Y update on the matrix pipe (v_mfma_f32_4x4x1), round-4 forms of the two smooth-loss cases (timing)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
tools/exp/mfma4x4_probe
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "spade_fused or fused_spade" 2>&1 | tail -3
timeout 300 python tools/exp/fused_kernel_bench.py > gpurun_out/r05_fused_bench_ymfma.txt 2>&1
cat gpurun_out/r05_fused_bench_ymfma.txt
timeout 300 python tools/exp/fused_kernel_bench.py --packed > gpurun_out/r05_fused_bench_ymfma_packed.txt 2>&1
cat gpurun_out/r05_fused_bench_ymfma_packed.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "norm or resblock" 2>&1 | tail -3
( timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -s --durations=5 -k "test_full_size_smooth_loss_backward and (guided_32to256_bs1-over4 or indep_16to512_bs1-over5)" 2>&1 | grep -E "guided_32|indep_16|passed|failed|Error|s call" ) > gpurun_out/r05_r04cases.txt 2>&1
cat gpurun_out/r05_r04cases.txt
