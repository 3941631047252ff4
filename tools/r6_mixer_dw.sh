#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_mixer_dw.py -x -q 2>&1 | tail -5 > $O/r6_mixer_dw_tests.txt
cat $O/r6_mixer_dw_tests.txt
for e in RD_X=0 RD_MIXER_DW=0; do
  env $e python bench.py --steps 6 --warmup 3 --no-cpu-baseline --dump-profile $O/r6_mixer_dw_$e.csv 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$e', r['value'], r['ms_per_step'])"
  head -8 $O/r6_mixer_dw_$e.csv
done
