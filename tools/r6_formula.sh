#!/bin/bash
# Round-6 formula decode: parity tests, A/B of the new launches (GEMV / attention / embedding-in-select) against the round-5 ones, per-kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -q -x -k "formula" 2>&1 | tail -3 > $O/r6_formula_tests.txt
{
echo "# new (default)"; python tools/bench_formula.py 8 32 2>&1 | grep decoder
echo "# RD_DEC_GEMV=0"; RD_DEC_GEMV=0 python tools/bench_formula.py 8 32 2>&1 | grep decoder
echo "# RD_DEC_ATTN2=0"; RD_DEC_ATTN2=0 python tools/bench_formula.py 8 32 2>&1 | grep decoder
echo "# RD_DEC_EMBED_IN_SELECT=0"; RD_DEC_EMBED_IN_SELECT=0 python tools/bench_formula.py 8 32 2>&1 | grep decoder
echo "# all off (round 5)"; RD_DEC_GEMV=0 RD_DEC_ATTN2=0 RD_DEC_EMBED_IN_SELECT=0 python tools/bench_formula.py 8 32 2>&1 | grep decoder
echo "# RD_DEC_GEMV_WGS=256"; RD_DEC_GEMV_WGS=256 python tools/bench_formula.py 8 2>&1 | grep decoder
echo "# RD_DEC_GEMV_WGS=1024"; RD_DEC_GEMV_WGS=1024 python tools/bench_formula.py 8 2>&1 | grep decoder
} > $O/r6_formula_ab.txt 2>&1
bash tools/prof_formula.sh 8 > /dev/null 2>&1
grep -E "dec_|gemm|gemv|layernorm" $O/formula_stats.csv | cut -c1-230 | head -14 > $O/r6_formula_stats.txt
cat $O/r6_formula_tests.txt $O/r6_formula_ab.txt $O/r6_formula_stats.txt
