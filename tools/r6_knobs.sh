cd $GRAFT_REPO_ROOT
for rep in 1 2; do for e in RD_X=0 RD_GEMM1_WGS=1; do
  echo -n "$e  "; env $e python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-passes 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['config'].get('step_wall_ms'))"
done; done
