#!/bin/bash
# PMC passes over the two fp16x2 GEMM kernels (run through gpurun) -> gpurun_out/evidence/pmc_gemm.md
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pg_$i -o p --output-format csv -- python $R/tools/pmc_gemm_f16x2.py > /dev/null 2>&1
done
python $R/tools/pmc_summary.py /tmp/pg_*/*counter_collection.csv > $O/pmc_gemm.md
rocprofv3 --kernel-trace --stats -d /tmp/pg_t -o t --output-format csv -- python $R/tools/pmc_gemm_f16x2.py > /dev/null 2>&1
grep -h gemm3 /tmp/pg_t/*kernel_stats.csv >> $O/pmc_gemm.md
cat $O/pmc_gemm.md
