cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -q 2>&1 | grep -E "passed|failed|^E " | tail -3
python -m pytest tests/test_gpu_model.py -q -k "full_size_step or train_step" 2>&1 | grep -E "passed|failed|^E " | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > /tmp/b.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof/ev_results.db 60 | grep -E "onehot_conv_wgrad|total GPU" | cut -c1-180
python -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
