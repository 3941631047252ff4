for cfg in "64 8" "64 12" "48 8" "48 12" "32 12" "32 16" "80 8"; do
  set -- $cfg
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rec-batch $1 --rec-streams $2 2>&1 | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('rec_batch', $1, 'streams', $2, d['value'], d['ms_per_step'], d['roofline']['step_kernel_ms'], d['config']['host_stage_ms'])"
done
