cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $O
python -m pytest tests -q -m gpu > $O/full.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/full.log | tail -6
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run > $O/bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
