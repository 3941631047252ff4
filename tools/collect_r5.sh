#!/bin/bash
# Round-5 evidence in one GPU call: the driver-shaped test run, SQ / FETCH / WRITE counter passes at the bench's own launch mix (first: the
# bench line reads the traffic summary they produce), the bench line (+ per-kernel table), rocprofv3 kernel stats (8 streams and
# serialised), the trace-gap analysis, backbone-only line.  Usage: bash tools/collect_r5.sh [tag]
set -u
TAG=${1:-r5}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/${TAG}_gputest.log; cat $O/${TAG}_gputest.log
cd /tmp && export TMPDIR=/tmp
ONE="--vary-pages 1 --resident-pages"
# counter passes: the bench's OWN configuration (8 rec streams, prefetch, same launch mix), exactly one step per pass, no per-op
# profiling pass behind it (RD_BENCH_STOP_AFTER_TIMED=1) - dispatches = launches per step
export RD_BENCH_STOP_AFTER_TIMED=1
B="python $R/bench.py --steps 1 --warmup 0 --setup-steps 0 --setup-passes 0 --no-cpu-baseline --no-extra-passes $ONE"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p1_$TAG -o a -- $B > /tmp/p1.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/p1_$TAG -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_sq.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_${c}_$TAG -o a -- $B > /tmp/p.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/p_${c}_$TAG -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_$c.csv > /dev/null
done
python $R/tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv $O/${TAG}_pmc_traffic.json "collection ${TAG}, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --steps 1 --warmup 0 --setup-steps 0 (8 rec streams, the bench's launch mix)" 1 > /dev/null
cp $O/${TAG}_pmc_traffic.json $R/profiles/pmc_traffic.json          # what bench.py reads (committed from gpurun_out afterwards)
unset RD_BENCH_STOP_AFTER_TIMED
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --dump-profile $O/${TAG}_bench_per_kernel.csv 2>/dev/null | tail -1 > $O/${TAG}_bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes $ONE > /tmp/rp.log 2>&1
cp /tmp/rp_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv
GPU_MAX_HW_QUEUES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes --no-prefetch $ONE > /tmp/rs.log 2>&1
cp /tmp/rs_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats_serial.csv
head -6 $O/${TAG}_rocprofv3_kernel_stats_serial.csv | cut -c1-170
(RD_BENCH_STOP_AFTER_TIMED=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$TAG -o t -- python $R/bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-extra-passes > /tmp/tg.log 2>&1
 python $R/tools/trace_gaps.py $(find /tmp/tg_$TAG -name "*kernel_trace.csv" | head -1) 330 > $O/${TAG}_trace_gaps.txt 2>&1)
head -12 $O/${TAG}_trace_gaps.txt
timeout 200 python $R/bench.py --only backbone 2>/dev/null | tail -1 > $O/${TAG}_bench_backbone.json
python - <<PY
import json
r = json.load(open("$O/${TAG}_bench.json"))
print(r["value"], r["ms_per_step"], r["config"]["step_wall_ms"], r["config"].get("gather_page_dets_v2_ms"))
print({k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_us", "traffic", "traffic_launches_per_step", "launches_per_step", "traffic_bytes_per_step", "algorithmic_bytes_per_step")})
print(json.dumps(r.get("s2_dropin"))[:300]); print(json.dumps(r["backbone"]["roofline"])[:300]); print(json.dumps(r["backbone"].get("fp32_mode")))
print(json.dumps(r["cpu_baseline"])[:500])
PY
