cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_ops.py -q > $O/t1.log 2>&1; grep -E "passed|failed|^E " $O/t1.log | tail -4
python -m pytest tests/test_gpu_model.py -q -k "full_size_step or train_step" > $O/t2.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/t2.log | tail -4
python tools/exp/graph_check.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > $O/bench_prof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof/ev_results.db > $O/kernel_stats.md
grep -E "finalize|total GPU" $O/kernel_stats.md | cut -c1-190
python -c "
import json; d=json.loads(open('$O/bench_prof.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
