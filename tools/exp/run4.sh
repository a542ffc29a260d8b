cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_conv.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py -x -q -k "train_step or kernel_path or full_size_step or data_parallel" 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('launches_per_step'), d.get('host_enqueue_ms_per_step'))"
