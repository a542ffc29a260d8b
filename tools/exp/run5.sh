cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3l; mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_ops.py -q > $O/t1.log 2>&1; grep -E "passed|failed|^E " $O/t1.log | tail -5
python -m pytest tests/test_gpu_model.py -q -k "kernel_path or full_size_step or benchmark_config or full_size_smooth" > $O/t2.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/t2.log | tail -8
python tools/exp/graph_check.py 2>&1 | tail -1
for v in True False True False; do
python tools/exp/ab.py PRESPLIT_A=$v -- --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRESPLIT_A=$v', d['ms_per_step'], d.get('peak_hbm_gb'))"
done
