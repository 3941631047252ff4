#!/bin/bash
# usage: tools/sweep_workers.sh  -> pages/s for several (workers, rec-streams) settings
for cfg in "1 8" "2 8" "2 16" "3 12" "4 16"; do
  set -- $cfg
  out=$(timeout 200 python bench.py --no-cpu-baseline --workers $1 --rec-streams $2 2>&1 | tail -1)
  echo "workers=$1 rec_streams=$2 $(echo "$out" | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["host_stage_ms"])
except Exception as e: print("ERR", e)')"
done
