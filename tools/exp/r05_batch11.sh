#!/bin/bash
# round 5, GPU batch 11: the 16-wave form of the fused SPADE kernel (DSEE_FUSED_W16=1): its tests, stand-alone timings against the
# 8-wave form; then tests/test_gpu_model.py under rocgdb for the native backtrace of the graph-replay segfault
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests, W16"
DSEE_FUSED_W16=1 timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "spade_fused or fused_spade" 2>&1 | tail -3
DSEE_FUSED_W16=1 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "norm or resblock or coarse" 2>&1 | tail -3
echo "== coarse, 8 waves"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "coarse" 2>&1 | grep -E "^E|passed|failed" | head
echo "== stand-alone"
for w in 0 1 0 1; do DSEE_FUSED_W16=$w timeout 300 python tools/exp/fused_kernel_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/^/W16=$w /"; done
for w in 0 1; do DSEE_FUSED_W16=$w timeout 300 python tools/exp/fused_kernel_bench.py --packed 2>&1 | grep -v amdgpu.ids | sed "s/^/W16=$w /"; done
} > gpurun_out/r05_w16.txt 2>&1
cat gpurun_out/r05_w16.txt
( time timeout 1800 rocgdb -batch -ex "handle all nostop noprint pass" -ex "handle SIGSEGV stop print nopass" -ex run -ex bt -ex "info threads" -ex "thread apply all bt 12" --args python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:faulthandler 2>&1 | grep -v "^\[New Thread\|^\[Thread.*exited\|^\[Detaching\|^\[Attaching\|RuntimeWarning\|self.sr_model" | tail -250 ) > gpurun_out/r05_segv_gdb.txt 2>&1
grep -n "#[0-9]" gpurun_out/r05_segv_gdb.txt | head -60
tail -5 gpurun_out/r05_segv_gdb.txt
