#!/bin/bash
# per-kernel statistics of the formula decode loop (rocprofv3 --kernel-trace --stats); writes gpurun_out/formula_stats.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RD_DECODE_GRAPH=0 timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o f -- python $R/tools/bench_formula.py ${1:-8} > /tmp/pf.log 2>&1
tail -8 /tmp/pf.log
f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
cp $f $O/formula_stats.csv
head -25 $f | cut -c1-200
