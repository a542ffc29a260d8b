# end-of-round sanity on the MI355X box: smoke(), the 2-rank launcher path over gloo on one device, default bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
DSEE_DIST_BACKEND=gloo DSEE_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-f32-run --batch-per-gpu 4 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
echo "2-rank rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_2rank_gloo.json').read().strip().splitlines()[-1]); print('n_gpus', d['n_gpus'], d['value'], d['ms_per_step'], d['hip_graphs']['enabled'])"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['host_enqueue_ms_per_step'])"
