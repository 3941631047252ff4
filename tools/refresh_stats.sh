#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command at the hand-over tree (8 rec streams / 1 rec stream), as tools/collect_profiles.sh does
set -u
TAG=${1:-rX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ONE="--vary-pages 1 --resident-pages"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes $ONE > /tmp/rp.log 2>&1
cp /tmp/rp_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rq_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes --rec-streams 1 --no-prefetch $ONE > /tmp/rq.log 2>&1
cp /tmp/rq_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats_1stream.csv
head -12 $O/${TAG}_rocprofv3_kernel_stats_1stream.csv | cut -c1-150
# the same 8-stream workload with ONE hardware queue: every stream's kernels are dispatched in order, so a kernel's duration is its duration
# alone on the chip - the figure bench.py's roofline pass measures with per-op HIP events (same launches: 154 of the dominant kernel per step)
GPU_MAX_HW_QUEUES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes --no-prefetch $ONE > /tmp/rs.log 2>&1
cp /tmp/rs_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats_serial.csv
head -4 $O/${TAG}_rocprofv3_kernel_stats_serial.csv | cut -c1-150
grep -o '"ms_per_step": [0-9.]*' /tmp/rs.log | head -1
