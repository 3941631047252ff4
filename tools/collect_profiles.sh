#!/bin/bash
# Round profile collection on the GPU box: bench line (+cpu baseline), rocprofv3 kernel stats of the same command, SQ and TCC PMC passes.
# Usage: bash tools/collect_profiles.sh <round-tag>      (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 --dump-profile $O/${TAG}_bench_per_kernel.csv 2>/dev/null | tail -1 > $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
# (the profiled passes run the round-1..3 form of the workload - one resident page set - so that per-kernel figures stay comparable
#  across rounds and a pass does not spend 20 s synthesising eight page sets; the kernels and their launch sizes are the same)
ONE="--vary-pages 1 --resident-pages"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes $ONE > /tmp/rp.log 2>&1
cp /tmp/rp_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv
# same workload with the recognition batches on ONE stream: per-kernel durations without co-running kernels (these are
# the durations bench.py's roofline pass measures with HIP events, so the two must agree)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rq_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-extra-passes --rec-streams 1 --no-prefetch $ONE > /tmp/rq.log 2>&1
cp /tmp/rq_$TAG/s_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats_1stream.csv
# the counter passes run the bench's OWN configuration (8 rec streams, prefetch, the same launch mix - VERDICT r4 weak #4); --setup-steps 0:
# exactly one step per pass, so that dispatches = launches per step
# RD_BENCH_STOP_AFTER_TIMED=1: no per-op profiling pass behind the timed step (it would launch every kernel a second time)
export RD_BENCH_STOP_AFTER_TIMED=1
B="python $R/bench.py --steps 1 --warmup 0 --setup-steps 0 --setup-passes 0 --no-cpu-baseline --no-extra-passes $ONE"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p1_$TAG -o a -- $B > /tmp/p1.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/p1_$TAG -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_sq.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_${c}_$TAG -o a -- $B > /tmp/p.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/p_${c}_$TAG -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_$c.csv > /dev/null
done
python $R/tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv $O/${TAG}_pmc_traffic.json "collection ${TAG}, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --steps 1 --warmup 0 --setup-steps 0 (8 rec streams, the bench's launch mix)" 1 > /dev/null
unset RD_BENCH_STOP_AFTER_TIMED
timeout 200 python $R/bench.py --only backbone 2>/dev/null | tail -1 > $O/${TAG}_bench_backbone.json
timeout 200 python $R/bench.py --rec-mode throughput --no-cpu-baseline --no-extra-passes $ONE 2>/dev/null | tail -1 > $O/${TAG}_bench_throughput_mode.json
timeout 200 python $R/bench.py --no-cpu-baseline --no-extra-passes $ONE 2>/dev/null | tail -1 > $O/${TAG}_bench_resident_one_set.json
cat $O/${TAG}_bench.json | cut -c1-900
head -8 $O/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-140
# round 4: the LDS-DMA-staged depthwise 3x3 against the register kernel, the step's GPU-busy analysis, the scheduling A/B
{
  timeout 100 python $R/tools/mb_dw.py 2>/dev/null | grep dw3x3
  RD_DW_LDS=0 timeout 100 python $R/tools/mb_dw.py 2>/dev/null | grep dw3x3
} > $O/${TAG}_microbench_dw.txt
(cd /tmp && RD_BENCH_STOP_AFTER_TIMED=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$TAG -o t -- python $R/bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-extra-passes > /tmp/tg.log 2>&1
 python $R/tools/trace_gaps.py $(find /tmp/tg_$TAG -name "*kernel_trace.csv" | head -1) 330 > $O/${TAG}_trace_gaps.txt 2>&1)
[ "${SKIP_AB:-0}" = "1" ] || {
  echo "# python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes <flags>: pages/s, ms/step, per-step wall (median / min / max / first), t_wait_maps"
  for f in "default" "--no-prefetch" "Q4" "Q4 --no-prefetch"; do
    q=16; a="$f"; case "$f" in default) a="";; Q4*) q=4; a="${f#Q4}";; esac
    GPU_MAX_HW_QUEUES=$q timeout 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes $a 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('GPU_MAX_HW_QUEUES=$q $a:', r['value'], r['ms_per_step'], r['config']['step_wall_ms'], r['config']['host_stage_ms']['t_wait_maps_ms'])"
  done
} > $O/${TAG}_ab_prefetch_queues.txt
cat $O/${TAG}_microbench_dw.txt; [ -f $O/${TAG}_ab_prefetch_queues.txt ] && cat $O/${TAG}_ab_prefetch_queues.txt; head -20 $O/${TAG}_trace_gaps.txt
if [ "${LIGHT:-0}" = "1" ]; then python $R/tools/isa_mix.py > $O/${TAG}_isa_mix.txt 2>/dev/null || true; exit 0; fi
# microbenchmarks behind DESIGN.md s3c: the ws mixer (round-2 form 200 vs prefetching form 400) and its ablations, each in its own process
{
  echo "# tools/mb_ws.py 200 400  (lc_mixer_ws_kernel<192>: round-2 form vs prefetching form, isolated, 20 iterations, error vs fp64)"
  timeout 200 python $R/tools/mb_ws.py 200 400 2>/dev/null | grep mixer
  echo "# tools/mb_ws_abl.py <bits>  (ablations, results are garbage, timing only; 32+ = prefetching form)"
  for a in 0 1 4 8 12 13 16 17 25 29 32 36 40 44 48 60 96 104 120; do timeout 100 python $R/tools/mb_ws_abl.py $a 2>/dev/null | grep "ws M="; done
} > $O/${TAG}_microbench_mixer_ws.txt
# formula head: decode tokens/s at B = 8 / 32 and the per-kernel statistics of the B = 8 loop
{
  echo "# tools/bench_formula.py 8 32"
  timeout 200 python $R/tools/bench_formula.py 8 32 2>/dev/null | grep -E "encoder|decoder"
  echo "# RD_SKINNY2=0 (round-2 skinny GEMM)"
  RD_SKINNY2=0 timeout 200 python $R/tools/bench_formula.py 8 32 2>/dev/null | grep -E "decoder"
  echo "# rocprofv3 --kernel-trace --stats, B = 8, graphs off (tools/prof_formula.sh): decode-loop kernels"
  bash $R/tools/prof_formula.sh 8 2>/dev/null | grep -E "skinny|dec_|layernorm"
} > $O/${TAG}_formula_decode.txt
cat $O/${TAG}_formula_decode.txt | cut -c1-160
# phase stamps of the 8-wavefront DMA GEMM in its round-3 form (DMA pieces in a clump behind the barrier): DESIGN.md s3d
timeout 100 python $R/tools/mb_gemm_trace.py 131072 768 384 0 2> $O/${TAG}_gemm_trace.txt > /dev/null
# static instruction mix / issue budget of the hot loops (hipcc -S only: also runs without a GPU)
python $R/tools/isa_mix.py > $O/${TAG}_isa_mix.txt 2>/dev/null || true
