#!/bin/bash
# SQ counters of the split-fp16 GEMM microbenchmark (one rocprofv3 --pmc pass; kernel-trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pm -o a -- python $R/tools/microbench.py --h3only > /tmp/pm.log 2>&1
tail -6 /tmp/pm.log
python $R/tools/pmc_summary.py $(find /tmp/pm -name "*counter_collection.csv" | head -1) $O/mb_pmc_sq.csv
