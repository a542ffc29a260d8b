#!/bin/bash
# Round evidence on the MI355X box (run through gpurun): bench lines, rocprofv3 kernel statistics, per-function step
# times, GEMM microbenchmarks + ablations, PMC traffic.  Everything lands in gpurun_out/evidence/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_fp32.json 2> $O/bench_fp32.err
python bench.py --steps 10 --warmup 3 --dtype fp16 > $O/bench_fp16.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --config guided_8x_256 --no-f32-run --no-cpu-baseline > $O/bench_guided_8x_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --config independent_32x_512 --no-f32-run --no-cpu-baseline > $O/bench_independent_32x_512.json 2>/dev/null
python bench.py --arith bf16x3 --steps 8 --warmup 2 --no-f32-run --no-cpu-baseline > $O/bench_bf16x3.json 2>/dev/null
python tools/bench_gemm2.py > $O/gemm_shapes_f16x2.txt 2>&1
BF16X3=1 python tools/bench_gemm2.py > $O/gemm_shapes_bf16x3.txt 2>&1
python tools/exp/gemm_pre_bench.py > $O/gemm_presplit.txt 2>&1
python tools/exp/fused_kernel_bench.py > $O/fused_kernel.txt 2>&1
python tools/exp/tn_bench.py > $O/gemm_tn_presplit.txt 2>&1
for v in True False True False; do python tools/exp/ab.py FUSED_NORM=$v -- --steps 8 --warmup 2 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSED_NORM=$v ms_per_step', d['ms_per_step'])"; done > $O/ab_fused_norm.txt
for v in True False True False; do python tools/exp/ab.py PRESPLIT_A=$v -- --steps 8 --warmup 2 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRESPLIT_A=$v ms_per_step', d['ms_per_step'])"; done > $O/ab_presplit.txt
python bench.py --steps 8 --warmup 2 --no-graphs --no-f32-run --no-cpu-baseline > $O/bench_no_graphs.json 2>/dev/null
python tools/step_functions.py > $O/step_functions.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/rocpd_summary.py /tmp/prof/ev_results.db > $O/kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-run --no-graphs > /dev/null 2>&1; done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/*counter_collection.csv /tmp/pmc_WRITE_SIZE/*counter_collection.csv > $O/pmc_traffic.json
ls -la $O
cd $R
for sw in PRESPLIT_DM PRESPLIT_GB; do for v in True False True False; do python tools/exp/ab.py $sw=$v -- --steps 8 --warmup 2 --no-cpu-baseline --no-f32-run 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=$v ms_per_step', d['ms_per_step'])"; done; done > $O/ab_presplit_dm.txt
cd $R && bash tools/exp/build_fused_abl.sh 1 2 4 8 16 64 > /dev/null 2>&1
for m in 1 2 4 8 16 64; do DSEE_LIB=tools/exp/libfabl_$m.so python tools/exp/fused_kernel_bench.py 2>&1 | grep -v amdgpu.ids; done > $O/fused_ablation.txt
cd $R && python -m pytest tests -q -m gpu -s > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -2
