cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3f; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -5 $O/gputests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ev -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-run > $O/bench_prof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof/ev_results.db > $O/kernel_stats.md
head -3 $O/kernel_stats.md
