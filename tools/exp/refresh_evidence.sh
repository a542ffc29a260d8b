# the headline bench line, kernel statistics and PMC traffic of the final code (subset of tools/collect_evidence.sh)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_fp32.json 2> $O/bench_fp32.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/rocpd_summary.py /tmp/prof/ev_results.db > $O/kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-run --no-graphs > /dev/null 2>&1; done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/*counter_collection.csv /tmp/pmc_WRITE_SIZE/*counter_collection.csv > $O/pmc_traffic.json
python -c "
import json; d=json.loads(open('$O/bench_fp32.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['mfma']['frac'], r['mfma_kernels_ms_per_step'], r['norm_forward']['ms_per_call'], d['spade_fused']['ms_per_step'], d['f32_mfma_only']['value'], d['bf16x3_exact']['value'], d['cpu_baseline']['value'], d['host_enqueue_ms_per_step'])"
tail -1 $O/kernel_stats.md
