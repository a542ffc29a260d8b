cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3p; mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_ops.py -q > $O/t0.log 2>&1; grep -E "passed|failed|^E " $O/t0.log | tail -6
python -m pytest tests/test_gpu_model.py -q -k "kernel_path or full_size or train_step or benchmark_config" > $O/t2.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/t2.log | tail -8
