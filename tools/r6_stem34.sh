#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_stem34.py -x -q 2>&1 | tail -5 > $O/r6_stem34_tests.txt
timeout 200 python tools/mb_stem34.py > $O/r6_stem34_mb.txt 2>&1
if [ "$1" = "full" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_launch_invariance.py -x -q -k "not formula" 2>&1 | tail -30 > $O/r6_stem34_parity.txt; fi
cat $O/r6_stem34_tests.txt $O/r6_stem34_mb.txt; [ "$1" = "full" ] && cat $O/r6_stem34_parity.txt
