#!/bin/bash
# Round-6 evidence on the MI355X box (run through gpurun): bench lines (fp32 headline as the driver runs it, the 16-bit storage
# mode, the other BASELINE configs), rocprofv3 kernel statistics for both precisions, per-function / per-geometry step times,
# PMC traffic, per-kernel shader clocks.  Everything lands in gpurun_out/evidence/; the summaries to be judged are copied to
# profiles/r06_* by hand.  A tool that exits non-zero (or writes a traceback instead of its table) makes the script exit non-zero
# after the remaining steps ran: its output file gets a ".FAILED" suffix instead of being mistaken for a profile.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
cd $R
FAILED=0
run() {   # run <output file> <command ...>: stdout -> file, non-zero exit or a Python traceback in it -> .FAILED
  local out=$1; shift
  "$@" > "$out" 2> "$out.err"
  local rc=$?
  if [ $rc -ne 0 ] || grep -q "^Traceback (most recent call last)" "$out" "$out.err"; then
    echo "FAILED (rc $rc): $*" | tee -a $O/FAILED.txt
    mv "$out" "$out.FAILED"; FAILED=1
  else
    rm -f "$out.err"
  fi
}
run $O/bench_fp32.json python bench.py --steps 20 --warmup 5
run $O/bench_fp16.json python bench.py --steps 20 --warmup 5 --dtype fp16
run $O/bench_fp16_two_term_norms.json python bench.py --steps 10 --warmup 3 --dtype fp16 --plan half_norms=False
run $O/bench_guided_8x_256.json python bench.py --steps 5 --warmup 2 --config guided_8x_256 --no-f32-run --no-cpu-baseline
run $O/bench_independent_32x_512.json python bench.py --steps 5 --warmup 2 --config independent_32x_512 --no-f32-run --no-cpu-baseline
run $O/bench_independent_32x_512_fp16.json python bench.py --steps 5 --warmup 2 --config independent_32x_512 --dtype fp16
run $O/bench_no_graphs.json python bench.py --steps 8 --warmup 2 --no-graphs --no-f32-run --no-cpu-baseline
# same-box A/B of this round's plan switches (each against bench_fp32_again.json)
run $O/bench_fp32_again.json python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline
run $O/bench_ab_gemm_w8.json python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline --plan gemm_w4=False
run $O/bench_ab_dgrad_gather.json python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline --plan dgrad_s2_parity=False
run $O/bench_ab_absmax_passes.json python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline --plan conv_amax_out=False
run $O/bench_ab_conv_igemm_no_halo.json python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline --plan conv_halo_f16=False
run $O/bench_ab_onehot_wgrad_valu.json python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline --plan onehot_wgrad_mfma=False
run $O/bench_ab_fused_w4.json python bench.py --steps 20 --warmup 5 --no-f32-run --no-cpu-baseline --plan fused_w4=True
run $O/step_shapes.txt python tools/step_shapes.py
run $O/step_functions.txt python tools/step_functions.py 60
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/prof16 /tmp/pmc_* /tmp/pmc16_* /tmp/pmcclk
run $O/bench_under_rocprof.json rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run
run $O/kernel_stats.md python $R/tools/rocpd_summary.py /tmp/prof/ev_results.db
run $O/kernel_stats_all.md python $R/tools/rocpd_summary.py /tmp/prof/ev_results.db 400
run $O/bench_fp16_under_rocprof.json rocprofv3 --kernel-trace --stats -d /tmp/prof16 -o ev -- python $R/bench.py --dtype fp16 --steps 5 --warmup 2
run $O/kernel_stats_fp16.md python $R/tools/rocpd_summary.py /tmp/prof16/ev_results.db
for c in FETCH_SIZE WRITE_SIZE; do run /tmp/pmc_$c.log rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-run --no-graphs; done
run $O/pmc_traffic.json python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/*counter_collection.csv /tmp/pmc_WRITE_SIZE/*counter_collection.csv
for c in FETCH_SIZE WRITE_SIZE; do run /tmp/pmc16_$c.log rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc16_$c -o p --output-format csv -- python $R/bench.py --dtype fp16 --steps 2 --warmup 1 --no-graphs; done
run $O/pmc_traffic_fp16.json python $R/tools/pmc_traffic.py /tmp/pmc16_FETCH_SIZE/*counter_collection.csv /tmp/pmc16_WRITE_SIZE/*counter_collection.csv
run /tmp/pmcclk.log rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/pmcclk -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-run --no-graphs
run $O/kernel_clocks_in_step.md python $R/tools/pmc_clock.py /tmp/pmcclk/*counter_collection.csv
cd $R
ls -la $O
exit $FAILED
