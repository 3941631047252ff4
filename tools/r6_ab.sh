#!/bin/bash
# Round-6 A/B: routing thresholds of the narrow pointwise layers on the backbone-only line + the small-input mixer test.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_parity.py -q -x -k "small_inputs" -s 2>&1 | grep -E "mixer small|passed|failed|Error" > $O/r6_ab_small.txt
run() { echo "== $*" >> $O/r6_ab.txt; env "$@" timeout 200 python bench.py --only backbone 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], json.dumps(r.get('roofline',{}))[:200])" >> $O/r6_ab.txt 2>&1; }
: > $O/r6_ab.txt
run RD_X=0
run RD_H3_1X1_MIN_N=64
run RD_H3_1X1_MIN_N=64 RD_H3_1X1_MIN_K=64
run RD_H3_1X1_MIN_K=64
run RD_X=0
cat $O/r6_ab.txt $O/r6_ab_small.txt
