cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
DSEE_DIST_BACKEND=gloo DSEE_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-f32-run --batch-per-gpu 4 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
echo "rc=$?"; tail -c 600 $O/bench_2rank_gloo.json; tail -5 $O/bench_2rank_gloo.err
timeout 900 python bench.py --dtype fp16 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
echo "rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_fp16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['dtype'], d.get('host_enqueue_ms_per_step'))"
python -m pytest tests/test_gpu_model.py -x -q -k "kernel_path_switches" 2>&1 | tail -3
