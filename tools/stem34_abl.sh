python tools/mb_stem34.py 2>/dev/null
echo WGS=1; RD_STEM34_WGS=1 python tools/mb_stem34.py 2>/dev/null
echo WGS=3; RD_STEM34_WGS=3 python tools/mb_stem34.py 2>/dev/null
for i in 1 2; do
echo "bench default"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
echo "bench RD_STEM34=0"; RD_STEM34=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
done
