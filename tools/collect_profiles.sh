#!/bin/bash
# Round profile collection on the GPU box: bench line (+cpu baseline), rocprofv3 kernel stats of the same command, SQ and TCC PMC passes.
# Usage: bash tools/collect_profiles.sh <round-tag>      (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python bench.py --dump-profile $O/${TAG}_bench_per_kernel.csv 2>/dev/null | tail -1 > $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes > /tmp/rp.log 2>&1
cp /tmp/rp_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv
# same workload with the recognition batches on ONE stream: per-kernel durations without co-running kernels (these are
# the durations bench.py's roofline pass measures with HIP events, so the two must agree)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rq_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes --rec-streams 1 > /tmp/rq.log 2>&1
cp /tmp/rq_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats_1stream.csv
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-passes --rec-streams 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p1_$TAG -o a -- $B > /tmp/p1.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/p1_$TAG -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_sq.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_${c}_$TAG -o a -- $B > /tmp/p.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/p_${c}_$TAG -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_$c.csv > /dev/null
done
python $R/tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv $O/${TAG}_pmc_traffic.json "collection ${TAG}, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --steps 1 --rec-streams 1" > /dev/null
timeout 200 python $R/bench.py --only backbone 2>/dev/null | tail -1 > $O/${TAG}_bench_backbone.json
timeout 200 python $R/bench.py --rec-mode strict --no-cpu-baseline --no-extra-passes 2>/dev/null | tail -1 > $O/${TAG}_bench_strict.json
cat $O/${TAG}_bench.json | cut -c1-900
head -8 $O/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-140
