cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -q -k "spade_fused or pre_split or small_channel" 2>&1 | grep -E "passed|failed|^E " | head
python tools/exp/fused_kernel_bench.py 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py -q -k "kernel_path or full_size_step" 2>&1 | grep -E "passed|failed|^E " | head
for v in True False True False; do
python tools/exp/ab.py PRESPLIT_A=$v -- --steps 10 --warmup 3 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRESPLIT_A=$v', d['ms_per_step'])"
done
