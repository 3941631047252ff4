#!/bin/bash
# A/B: CUs the persistent matrix kernels size their grids for (RD_PERSISTENT_CUS), whole-step pages/s, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
: > $O/r6_cus.txt
for rep in 1 2; do for n in 256 240 224 208 192; do
  echo -n "RD_PERSISTENT_CUS=$n  " >> $O/r6_cus.txt
  RD_PERSISTENT_CUS=$n python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-passes 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['config'].get('step_wall_ms'))" >> $O/r6_cus.txt
done; done
cat $O/r6_cus.txt
