R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
export RD_BENCH_STOP_AFTER_TIMED=1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_clk -o a -- python $R/bench.py --steps 1 --warmup 0 --setup-steps 0 --setup-passes 0 --no-cpu-baseline --no-extra-passes --vary-pages 1 --resident-pages > /tmp/pclk.log 2>&1
python $R/tools/pmc_clock.py $(find /tmp/p_clk -name "*counter_collection.csv" | head -1) $(find /tmp/p_clk -name "*kernel_trace.csv" | head -1) $O/r6d_pmc_clock.csv
