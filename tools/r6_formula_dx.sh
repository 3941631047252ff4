cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -q -x -k "formula" 2>&1 | tail -2
for rep in 1 2; do
echo -n "default       "; python tools/bench_formula.py 8 2>&1 | grep decoder
echo -n "RD_DEC_GEMV_DX=0 "; RD_DEC_GEMV_DX=0 python tools/bench_formula.py 8 2>&1 | grep decoder
done
