#!/bin/bash
# round 5, GPU batch 16: coarse tests + smoke on the final library; SQ / LDS counters of the fused SPADE kernel alone
# (tools/exp/fused_kernel_bench.py under rocprofv3 --pmc, one counter set per pass)
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "coarse" 2>&1 | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pf_$i -o p --output-format csv -- python $R/tools/exp/fused_kernel_bench.py > /dev/null 2>&1
done
PMC_KERNEL=spade_fused python $R/tools/pmc_summary.py /tmp/pf_*/*counter_collection.csv > $R/gpurun_out/r05_pmc_fused.md 2>&1
cat $R/gpurun_out/r05_pmc_fused.md
