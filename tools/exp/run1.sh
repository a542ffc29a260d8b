cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_model.py -x -q -k "kernel_path_switches" -s > gpurun_out/switches.log 2>&1
grep -n "=False\|=0.0" gpurun_out/switches.log | head -40
