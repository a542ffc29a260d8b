cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -q -k "dout_transform or spade_fused or pre_split" 2>&1 | grep -E "passed|failed|^E " | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > $O/bench_prof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof/ev_results.db > $O/kernel_stats_nt.md
grep -E "f16x2_kernel|total GPU" $O/kernel_stats_nt.md | cut -c1-200
python -c "
import json
d=json.loads(open('$O/bench_prof.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
