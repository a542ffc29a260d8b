cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -k "producers or resblock or norm" > $O/t1.log 2>&1; grep -E "passed|failed" $O/t1.log | tail -2
python -m pytest tests/test_gpu_model.py -q -k "hip_graphs or kernel_path" > $O/t2.log 2>&1; grep -E "passed|failed|^E  |^FAILED" $O/t2.log | tail -8
for v in True False True False; do
python tools/exp/ab.py PRODUCER_STATS=$v -- --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRODUCER_STATS=$v', d['ms_per_step'])"
done
