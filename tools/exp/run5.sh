cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py -q -k "full_size_step or kernel_path or train_step" 2>&1 | grep -E "passed|failed|^E " | tail -3
python -m pytest tests/test_gpu_conv.py -q -k "winograd_conv_autograd" 2>&1 | grep -E "passed|failed|^E " | tail -3
python tools/exp/graph_check.py 2>&1 | tail -1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
