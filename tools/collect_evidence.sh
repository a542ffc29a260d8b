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
for m in 1 2 4 8 16 14; do echo "ablation mask $m"; DSEE_LIB=tools/exp/libabl_$m.so python tools/bench_gemm2.py 2>&1 | grep -E "^conv 512->512 @256.*tile 256|^gamma/beta fwd @256.*tile 256"; done > $O/gemm_ablation.txt
python tools/step_functions.py > $O/step_functions.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/rocpd_summary.py /tmp/prof/ev_results.db > $O/kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-run > /dev/null 2>&1; done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/*counter_collection.csv /tmp/pmc_WRITE_SIZE/*counter_collection.csv > $O/pmc_traffic.json
ls -la $O
cd $R && DSEE_LIB=tools/exp/libabl_32.so python tools/exp/gemm_phases.py > $O/gemm_phases.txt 2>&1
