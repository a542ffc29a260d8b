cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -q -k "both_operands or dout_transform or tn_pre or pre_split" -s > $O/t0.log 2>&1; grep -E "passed|failed|^E |vs f64" $O/t0.log | tail -12
python -m pytest tests/test_gpu_conv.py tests/test_gpu_ops.py -q > $O/t1.log 2>&1; grep -E "passed|failed|^E " $O/t1.log | tail -5
python -m pytest tests/test_gpu_model.py -q -k "kernel_path or full_size_step or train_step" > $O/t2.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/t2.log | tail -8
python tools/exp/graph_check.py 2>&1 | tail -1
for v in True False True False; do
python tools/exp/ab.py PRESPLIT_DM=$v -- --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRESPLIT_DM=$v', d['ms_per_step'], d.get('peak_hbm_gb'))"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > $O/bench_prof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof/ev_results.db > $O/kernel_stats.md
grep -E "dout|gemm3t|gemm3a" $O/kernel_stats.md | cut -c1-200
