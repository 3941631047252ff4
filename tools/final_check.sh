set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r4_gputest_final.log
cat $O/r4_gputest_final.log
timeout 600 python bench.py --steps 20 --warmup 5 --dump-profile $O/r4_bench_per_kernel.csv 2>/dev/null | tail -1 > $O/r4_bench.json
python -c "
import json; r=json.load(open('$O/r4_bench.json')); print(r['value'], r['ms_per_step'], r['config']['step_wall_ms'], r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['traffic'], r['formula'] if 'formula' in r else None)"
{
  echo "# tools/bench_formula.py 8 32"
  timeout 200 python $R/tools/bench_formula.py 8 32 2>/dev/null | grep -E "encoder|decoder"
  echo "# RD_DEC_FUSED=0 (round-3 form: separate q|k|v and q projections, 52 launches per token)"
  RD_DEC_FUSED=0 timeout 200 python $R/tools/bench_formula.py 8 32 2>/dev/null | grep -E "decoder"
  echo "# rocprofv3 --kernel-trace --stats, B = 8, graphs off (tools/prof_formula.sh): decode-loop kernels"
  bash $R/tools/prof_formula.sh 8 2>/dev/null | grep -E "skinny|dec_|layernorm"
} > $O/r4_formula_decode.txt
cat $O/r4_formula_decode.txt | cut -c1-170
